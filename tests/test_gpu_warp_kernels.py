"""The HIP ray-cast kernels through the C ABI vs frames produced by EXECUTING the reference's own kernel source and sensor
classes under oracle/wp_emul.py (tests/golden/warp_kernels_*.npz; generator oracle/gen_golden_warp_kernels.py; what is
emulated: that module's header).  No CPU oracle is involved here: GPU output vs the reference's recorded output, bit for bit
-- raw distances / point clouds / normals, segmentation and face ids, and the frame after WarpSensor's range limits and
normalisation (both as the separate post-processing launch and fused into the ray-cast's store epilogue).
Rows a23, a24, a25, f1 of SURVEY section 8."""
import ctypes as C

import numpy as np
import pytest
import torch
from warp_golden_util import bits, cases, cfg_of, kind_of, limits_of, load, mode_of, seg_mask

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODE = {"range": 0, "depth": 1, "pointcloud": 2, "pointcloud_world": 3, "normal": 4, "normal_world": 5}


def _scene(g, box_objects=False):
    """the fixture's world-frame triangles as the scene (identity asset pose), BVH built by the library.  The fixture's boxes are
    closed 12-triangle meshes in a face order of their own -- NOT trimesh.creation.box's: with AGX_BVH_BOX_OBJECTS the builder must
    recognise that and keep their triangle subtrees (`box_objects=True`: the adversarial case)."""
    from test_gpu_raycast import Scene

    tw = g["tri_world"]
    n, nt = tw.shape[0], tw.shape[1]
    ident = np.zeros((n, 1, 13), np.float32)
    ident[:, :, 6] = 1.0
    sc = dict(tri_local=tw, tri_asset=np.zeros(nt, np.int32), asset_state=ident, tri_seg=g["tri_seg"], half=np.ones((n, 1, 3), np.float32))
    S = Scene(sc)
    # True: AGX_BVH_BOX_OBJECTS | AGX_BVH_OBJECT_TREE (box scenes as SceneManager builds them); "triangle_level": object nodes in the
    # triangle-level tree (scenes with non-box primitives); False: triangle subtrees only
    S.ppo = (S.ppo & 0xFFFF) | {True: 0x30000000, "triangle_level": 0x20000000, False: 0}[box_objects]
    S.build()
    S.tri_world.copy_(torch.from_numpy(tw).to(DEV))  # exactly the fixture's bits (the identity transform may flip a -0)
    S.L.check(S.lib.agx_bvh_build(S.n, S.nt, S.ppo, S.L.dptr(S.tri_world), None, S.L.dptr(S.nodes), S.L.dptr(S.work), S.stream))
    torch.cuda.synchronize()
    return S


def _post(S, g, tag, raw):
    lim = limits_of(g, tag)
    if lim is None:
        return raw
    c = cfg_of(g, tag)
    px = torch.from_numpy(raw.copy()).to(DEV)
    p = S.L.dptr
    if c["return_pointcloud"]:
        S.L.check(S.lib.agx_sensor_postprocess_points(px.numel() // 3, p(px), None, None, 0.0, 0.0, 0.0, 0.0, 0.0, lim[0], lim[1], lim[2], lim[3],
                                                      int(not c["world_frame"]), int(lim[4]), S.stream))
    else:
        S.L.check(S.lib.agx_sensor_postprocess(px.numel(), p(px), None, None, 0.0, 0.0, 0.0, 0.0, 0.0, lim[0], lim[1], lim[2], lim[3], int(lim[4]), S.stream))
    torch.cuda.synchronize()
    return px.cpu().numpy()


def _check(S, g, tag, px, seg, fused=None):
    raw = g[tag + "_raw"]
    assert px.shape == raw.shape
    assert np.array_equal(bits(px), bits(raw)), (tag, int((bits(px) != bits(raw)).sum()), float(np.abs(px - raw).max()))
    if tag + "_seg" in g.files:
        m, ref = seg_mask(g, tag), g[tag + "_seg"]
        assert np.array_equal(seg[m], ref[m]) if m is not None else np.array_equal(seg, ref), tag
    assert np.array_equal(bits(_post(S, g, tag, px)), bits(g[tag + "_final"])), tag
    if fused is not None:  # limits + normalisation in the ray-cast's own epilogue (scalar images)
        assert np.array_equal(bits(fused), bits(g[tag + "_final"])), tag


@pytest.fixture(scope="module")
def scenes():
    return {}


def _get(scenes, kind):
    if kind not in scenes:
        g = load(kind)
        scenes[kind] = (g, _scene(g))
    return scenes[kind]


@pytest.mark.parametrize("tag", cases("camera"))
def test_camera_kernels_vs_reference_source(scenes, tag):
    g, S = _get(scenes, "camera")
    c = cfg_of(g, tag)
    mode = MODE[mode_of(g, tag, "camera")]
    args = (int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], int(g[tag + "_cxy"][0]), int(g[tag + "_cxy"][1]), mode,
            g[tag + "_sensor_position"], g[tag + "_sensor_orientation"])
    px, seg = S.camera(*args)
    fused = S.camera(*args, limits=limits_of(g, tag))[0] if mode <= 1 else None
    _check(S, g, tag, px, seg, fused)


@pytest.mark.parametrize("tag", cases("lidar"))
def test_lidar_kernels_vs_reference_source(scenes, tag):
    g, S = _get(scenes, "lidar")
    c = cfg_of(g, tag)
    mode = MODE[mode_of(g, tag, "lidar")]
    args = (g[tag + "_ray_vectors"], c["max_range"], mode, g[tag + "_sensor_position"], g[tag + "_sensor_orientation"])
    px, seg = S.lidar(*args)
    fused = S.lidar(*args, limits=limits_of(g, tag))[0] if mode == 0 else None
    _check(S, g, tag, px, seg, fused)


@pytest.mark.parametrize("tag", cases("stereo"))
def test_stereo_kernels_vs_reference_source(scenes, tag):
    g, S = _get(scenes, "stereo")
    c = cfg_of(g, tag)
    mode = MODE[mode_of(g, tag, "stereo")]
    args = (int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], c["baseline"], int(g[tag + "_cxy"][0]), int(g[tag + "_cxy"][1]), mode,
            g[tag + "_sensor_position"], g[tag + "_sensor_orientation"])
    px, seg = S.stereo(*args)
    fused = S.stereo(*args, limits=limits_of(g, tag))[0] if mode <= 1 else None
    _check(S, g, tag, px, seg, fused)


def test_box_object_flag_on_boxes_of_another_face_order_keeps_their_triangle_subtrees():
    """AGX_BVH_BOX_OBJECTS on a scene whose 12-triangle objects ARE boxes, but not in trimesh's face order (the fixture's own
    BOX_F): every vertex is a corner, yet the triangles are not where the ray-cast's face table expects them -- the builder must
    not end the tree there.  No object node in the tree, and the frames stay the reference-executed golden's, bit for bit."""
    g = load("camera")
    S = _scene(g, box_objects=True)
    assert object_node_count(S) == 0
    c = cfg_of(g, "depth_seg")
    px, seg = S.camera(int(c["width"]), int(c["height"]), g["depth_seg_kinv"], c["max_range"], int(g["depth_seg_cxy"][0]), int(g["depth_seg_cxy"][1]), 1,
                       g["depth_seg_sensor_position"], g["depth_seg_sensor_orientation"])
    assert np.array_equal(bits(px), bits(g["depth_seg_raw"])) and np.array_equal(seg, g["depth_seg_seg"])


def object_node_count(S):
    """object nodes REACHED from the roots of the built trees (child reference >= 0 with AGX_BVH_OBJECT_REF, bit 30, set;
    include/aerial_gym_hip.h).  A walk, not a scan: an object node's own record holds floats in the child slots."""
    NI = S.nodes.cpu().numpy().view(np.int32)
    count = 0
    for e in range(NI.shape[0]):
        stack = [0]
        while stack:
            i = stack.pop()
            for slot in (3, 7):
                c = int(NI[e, i, slot])
                if c < 0:
                    continue
                if c & 0x40000000:
                    count += 1
                else:
                    stack.append(c)
    return count


def box_scene_frame(S, g, tag, limits=None):
    """one case of tests/golden/warp_kernels_boxes.npz through the C ABI (camera / LiDAR / stereo by the case's `_kind`)"""
    kind = kind_of(g, tag)
    c = cfg_of(g, tag)
    mode = MODE[mode_of(g, tag, kind)]
    pose = (g[tag + "_sensor_position"], g[tag + "_sensor_orientation"])
    kw = {} if limits is None else {"limits": limits}
    if kind == "camera":
        return S.camera(int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], int(g[tag + "_cxy"][0]), int(g[tag + "_cxy"][1]), mode, *pose, **kw), mode <= 1
    if kind == "lidar":
        return S.lidar(g[tag + "_ray_vectors"], c["max_range"], mode, *pose, **kw), mode == 0
    return S.stereo(int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], c["baseline"], int(g[tag + "_cxy"][0]), int(g[tag + "_cxy"][1]), mode,
                    *pose, **kw), mode <= 1


@pytest.mark.parametrize("box_objects", [True, "triangle_level", False])
@pytest.mark.parametrize("tag", cases("boxes"))
def test_object_node_traversal_vs_reference_source(scenes, tag, box_objects):
    """the traversal that ships by default (AGX_BVH_BOX_OBJECTS: the tree ends at a recognised box, its faces are
    chosen by box_face_candidates) against frames the reference's own kernels produced over boxes in trimesh.creation.box's vertex and
    face order -- what its loader gives Warp (assets/warp_asset.py:19-24).  (a) every box of the fixture IS an object node in the
    tree; (b) distances, point clouds, normals, segmentation and face ids bit for bit; (c) the same with triangle subtrees."""
    key = ("boxes", box_objects)
    if key not in scenes:
        g = load("boxes")
        scenes[key] = (g, _scene(g, box_objects=box_objects))
    g, S = scenes[key]
    n_boxes = g["tri_world"].shape[0] * g["tri_world"].shape[1] // 12
    assert object_node_count(S) == (n_boxes if box_objects else 0)  # (both tree forms end at every box)
    (px, seg), scalar = box_scene_frame(S, g, tag)
    fused = box_scene_frame(S, g, tag, limits=limits_of(g, tag))[0][0] if scalar and limits_of(g, tag) is not None else None
    _check(S, g, tag, px, seg, fused)
