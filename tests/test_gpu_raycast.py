"""GPU parity tests of the scene / ray-cast path through the C ABI: bit-exact depth, range,
segmentation and point clouds vs the brute-force CPU oracle on identical inputs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
from scene_util import random_box_scene, random_robot_states

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dt is None else t.to(dt)


class Scene:
    def __init__(self, sc):
        from aerial_gym_simulator_amd import _lib

        self.L, self.lib = _lib, _lib.load()
        self.n, self.nt = sc["tri_local"].shape[0], sc["tri_local"].shape[1]
        self.na = sc["asset_state"].shape[1]
        self.tri_local, self.tri_asset = T(sc["tri_local"]), T(sc["tri_asset"])
        self.asset_state, self.tri_seg, self.half = T(sc["asset_state"]), T(sc["tri_seg"]), T(sc["half"])
        self.tri_world = torch.zeros_like(self.tri_local)
        self.nodes = torch.zeros(self.n, self.nt - 1, 16, device=DEV)
        self.work = torch.zeros(self.n + 2, dtype=torch.int32, device=DEV)
        self.ppo = 12 if self.nt % 12 == 0 else 0  # scene_util scenes are box soups
        # AGX_BVH_BOX_OBJECTS (the product's default): the tree ends at every object the builder recognises as a trimesh box;
        # AGX_TEST_BOX_OBJECTS=0 runs this file on triangle subtrees only.  Either way every frame is compared bit for bit.
        if self.ppo and os.environ.get("AGX_TEST_BOX_OBJECTS", "1") != "0":
            self.ppo |= 0x30000000  # AGX_BVH_BOX_OBJECTS | AGX_BVH_OBJECT_TREE: what SceneManager passes for box scenes
        self.stream = _lib.current_stream(DEV)

    def build(self, mask=None):
        p, L = self.L.dptr, self.L
        self._mask_t = T(mask) if mask is not None else None  # keep alive until the kernels ran
        mk = p(self._mask_t) if mask is not None else None
        L.check(self.lib.agx_scene_transform(self.n, self.nt, self.na, p(self.tri_local), p(self.tri_asset), p(self.asset_state),
                                             mk, p(self.tri_world), self.stream))
        L.check(self.lib.agx_bvh_build(self.n, self.nt, self.ppo, p(self.tri_world), mk, p(self.nodes), p(self.work), self.stream))
        torch.cuda.synchronize()

    def _lim(self, limits):
        """limits = (min_range, max_range, far_oor, near_oor, normalize) -> AgxRangeLimits*, or NULL (raw distances)"""
        if limits is None:
            return None
        self._limits_struct = self.L.AgxRangeLimits(*[float(x) for x in limits[:4]], int(limits[4]))
        return C.byref(self._limits_struct)

    def camera(self, W, H, kinv, far, cx, cy, mode, pos, quat, seg=True, limits=None):
        p, L = self.L.dptr, self.L
        S = pos.shape[1]
        shape = (self.n, S, H, W) if mode <= 1 else (self.n, S, H, W, 3)
        px = torch.zeros(shape, device=DEV)
        sg = torch.zeros((self.n, S, H, W), dtype=torch.int32, device=DEV) if seg else None
        kin = (C.c_float * 4)(*[float(x) for x in kinv])
        tp, tq = T(pos), T(quat)
        L.check(self.lib.agx_raycast_camera(self.n, S, W, H, kin, float(far), cx, cy, mode, p(tp), p(tq), p(self.tri_world),
                                            p(self.tri_seg), p(self.nodes), self.nt, p(px), p(sg) if seg else None,
                                            self._lim(limits), self.stream))
        torch.cuda.synchronize()
        return px.cpu().numpy(), (sg.cpu().numpy() if seg else None)

    def stereo(self, W, H, kinv, far, baseline, cx, cy, mode, pos, quat, limits=None):
        p, L = self.L.dptr, self.L
        S = pos.shape[1]
        shape = (self.n, S, H, W) if mode <= 1 else (self.n, S, H, W, 3)
        px = torch.zeros(shape, device=DEV)
        sg = torch.zeros((self.n, S, H, W), dtype=torch.int32, device=DEV)
        kin = (C.c_float * 4)(*[float(x) for x in kinv])
        tp, tq = T(pos), T(quat)
        L.check(self.lib.agx_raycast_stereo_camera(self.n, S, W, H, kin, float(far), float(baseline), cx, cy, mode, p(tp), p(tq),
                                                   p(self.tri_world), p(self.tri_seg), p(self.nodes), self.nt, p(px), p(sg),
                                                   self._lim(limits), self.stream))
        torch.cuda.synchronize()
        return px.cpu().numpy(), sg.cpu().numpy()

    def lidar(self, rv, far, mode, pos, quat, limits=None):
        p, L = self.L.dptr, self.L
        S, H, W = pos.shape[1], rv.shape[0], rv.shape[1]
        shape = (self.n, S, H, W) if mode == 0 else (self.n, S, H, W, 3)
        px = torch.zeros(shape, device=DEV)
        sg = torch.zeros((self.n, S, H, W), dtype=torch.int32, device=DEV)
        trv, tp, tq = T(rv), T(pos), T(quat)
        L.check(self.lib.agx_raycast_lidar(self.n, S, W, H, p(trv), float(far), mode, p(tp), p(tq), p(self.tri_world),
                                           p(self.tri_seg), p(self.nodes), self.nt, p(px), p(sg), self._lim(limits), self.stream))
        torch.cuda.synchronize()
        return px.cpu().numpy(), sg.cpu().numpy()


def _poses(orc, n, sc, seed, lidar=False, S=1):
    st = random_robot_states(n, seed, *sc["bounds"])
    rng = np.random.default_rng(seed + 100)
    lp = rng.uniform([0.07, -0.06, 0.01], [0.12, 0.03, 0.04], (n, S, 3)).astype(np.float32)
    e = np.deg2rad(rng.uniform(-5, 5, (n * S, 3))).astype(np.float32)
    lq = orc.quat_from_euler(e).reshape(n, S, 4)
    frame_deg = [0.0, 0.0, 0.0] if lidar else [-90.0, 0.0, -90.0]
    frame = orc.quat_from_euler(np.deg2rad(np.array([frame_deg], np.float32)))[0]
    pos, quat = orc.sensor_pose(st, lp, lq, frame)
    return st, lp, lq, frame, pos, quat


def test_scene_transform_and_pose_bit_exact(orc):
    from gpu_harness import DynHarness
    from conftest import golden_params, load_golden

    sc = random_box_scene(5, 30, seed=1)
    S = Scene(sc)
    mask = np.array([1, 0, 1, 1, 0], np.uint8)
    S.build(mask)
    ref = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    got = S.tri_world.cpu().numpy()
    assert np.array_equal(got[mask.astype(bool)], ref[mask.astype(bool)])
    assert np.all(got[~mask.astype(bool)] == 0)  # masked-out envs untouched
    n = 5
    st, lp, lq, frame, pos, quat = _poses(orc, n, sc, 3, S=2)
    H = DynHarness(golden_params(load_golden("step_quad_position")), n)
    H.set(state=st)
    from aerial_gym_simulator_amd import _lib

    gp, gq = torch.zeros(n, 2, 3, device=DEV), torch.zeros(n, 2, 4, device=DEV)
    fq = (C.c_float * 4)(*[float(x) for x in frame])
    tlp, tlq = T(lp), T(lq)
    _lib.check(H.lib.agx_sensor_pose(H.B, n, 2, _lib.dptr(tlp), _lib.dptr(tlq), fq, _lib.dptr(gp), _lib.dptr(gq), H.stream()))
    torch.cuda.synchronize()
    assert np.array_equal(gp.cpu().numpy(), pos) and np.array_equal(gq.cpu().numpy(), quat)


@pytest.mark.parametrize("mode", ["depth", "range", "pointcloud", "pointcloud_world"])
def test_camera_bit_exact_config3_scene(orc, mode):
    """BASELINE config 3 scene: 100 random boxes + 6 walls (1272 triangles), 64 x 48, hfov 87."""
    n = 6
    sc = random_box_scene(n, 100, seed=7)
    S = Scene(sc)
    S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    _, _, _, _, pos, quat = _poses(orc, n, sc, 11)
    kinv, cx, cy = orc.camera_kinv(64, 48, 87.0)
    ref_px, ref_seg = orc.raycast_camera(64, 48, kinv, 10.0, cx, cy, mode, pos, quat, tris, sc["tri_seg"])
    got_px, got_seg = S.camera(64, 48, kinv, 10.0, cx, cy, orc.MODE[mode], pos, quat)
    assert np.array_equal(got_seg, ref_seg)          # segmentation ids: bit-exact
    assert np.array_equal(got_px, ref_px)            # depth / range / points: bit-exact
    hit = ref_seg != -2
    assert 0.3 < hit.mean() <= 1.0


def test_camera_odd_size_multi_sensor(orc):
    n = 3
    sc = random_box_scene(n, 20, seed=9)
    S = Scene(sc)
    S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    _, _, _, _, pos, quat = _poses(orc, n, sc, 5, S=2)
    kinv, cx, cy = orc.camera_kinv(37, 21, 70.0)  # not multiples of the 8x8 tile
    ref_px, ref_seg = orc.raycast_camera(37, 21, kinv, 7.0, cx, cy, "depth", pos, quat, tris, sc["tri_seg"])
    got_px, got_seg = S.camera(37, 21, kinv, 7.0, cx, cy, 1, pos, quat)
    assert np.array_equal(got_seg, ref_seg) and np.array_equal(got_px, ref_px)


@pytest.mark.parametrize("mode", ["range", "pointcloud"])
def test_lidar_bit_exact_config4(orc, mode):
    """BASELINE config 4 sensor: 32 x 512 LiDAR, hfov +-180, vfov +-45."""
    n = 2
    sc = random_box_scene(n, 100, seed=21)
    S = Scene(sc)
    S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    _, _, _, _, pos, quat = _poses(orc, n, sc, 2, lidar=True)
    rv = orc.lidar_ray_table(32, 512, -180, 180, -45, 45)
    ref_px, ref_seg = orc.raycast_lidar(rv, 10.0, mode, pos, quat, tris, sc["tri_seg"])
    got_px, got_seg = S.lidar(rv, 10.0, orc.MODE[mode], pos, quat)
    assert np.array_equal(got_seg, ref_seg) and np.array_equal(got_px, ref_px)


@pytest.mark.parametrize("mode", ["normal", "normal_world"])
def test_normal_face_id_sensors_bit_exact(orc, mode):
    """warp_normal_faceID_{cam,lidar}: geometric normals + face indices."""
    n = 3
    sc = random_box_scene(n, 100, seed=31)
    S = Scene(sc)
    S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    _, _, _, _, pos, quat = _poses(orc, n, sc, 4)
    kinv, cx, cy = orc.camera_kinv(64, 48, 87.0)
    ref_px, ref_face = orc.raycast_camera(64, 48, kinv, 10.0, cx, cy, mode, pos, quat, tris, sc["tri_seg"])
    got_px, got_face = S.camera(64, 48, kinv, 10.0, cx, cy, orc.MODE[mode], pos, quat)
    assert np.array_equal(got_face, ref_face) and np.array_equal(got_px, ref_px)
    assert ref_face.max() > 72 and (ref_face >= 0).mean() > 0.3  # face indices, boxes are hit
    hit = ref_face >= 0
    if mode == "normal_world":  # the camera-frame variant projects on rd_p, rd_p x e_z, rd_p x e_y: not orthonormal
        assert np.allclose(np.linalg.norm(ref_px[hit], axis=-1), 1.0, atol=1e-5)  # once the camera is tilted (reference quirk, kept)
    _, _, _, _, lpos, lquat = _poses(orc, n, sc, 6, lidar=True)
    rv = orc.lidar_ray_table(16, 128, -180, 180, -45, 45)
    ref_px, ref_face = orc.raycast_lidar(rv, 10.0, mode, lpos, lquat, tris, sc["tri_seg"])
    got_px, got_face = S.lidar(rv, 10.0, orc.MODE[mode], lpos, lquat)
    assert np.array_equal(got_face, ref_face) and np.array_equal(got_px, ref_px)


@pytest.mark.parametrize("mode", ["depth", "range", "pointcloud", "pointcloud_world"])
def test_stereo_camera_bit_exact(orc, mode):
    """warp_stereo_camera_kernels.py -- occlusion-checked pixels (-1 where the stereo
    partner cannot see the point), second ray is an any-hit packet traversal."""
    n = 4
    sc = random_box_scene(n, 100, seed=13)
    S = Scene(sc)
    S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    _, _, _, _, pos, quat = _poses(orc, n, sc, 17)
    kinv, cx, cy = orc.camera_kinv(80, 45, 87.0)
    for baseline in (0.095, 0.6):
        ref_px, ref_seg = orc.raycast_stereo_camera(80, 45, kinv, 10.0, baseline, cx, cy, mode, pos, quat, tris, sc["tri_seg"])
        got_px, got_seg = S.stereo(80, 45, kinv, 10.0, baseline, cx, cy, orc.MODE[mode], pos, quat)
        assert np.array_equal(got_seg, ref_seg) and np.array_equal(got_px, ref_px)
    if mode == "depth":
        invalid = ref_px == -1.0
        assert 0.002 < invalid.mean() < 0.5  # occlusion shadows exist at a 0.6 m baseline
        mono, mono_seg = S.camera(80, 45, kinv, 10.0, cx, cy, 1, pos, quat)
        assert np.array_equal(got_px[~invalid], mono[~invalid]) and np.array_equal(got_seg[~invalid], mono_seg[~invalid])


def test_rebuild_after_reset_is_idempotent_and_masked(orc):
    n = 4
    sc = random_box_scene(n, 50, seed=2)
    S = Scene(sc)
    S.build()
    nodes0 = S.nodes.clone()
    S.build()
    I = lambda t: t.view(torch.int32)  # noqa: E731  (child slots hold int bits: compare bit patterns)
    assert torch.equal(I(S.nodes), I(nodes0))  # deterministic build
    # move the obstacles of envs 1 and 3 only
    sc2 = random_box_scene(n, 50, seed=99)
    S.asset_state[1] = T(sc2["asset_state"][1])
    S.asset_state[3] = T(sc2["asset_state"][3])
    S.build(np.array([0, 1, 0, 1], np.uint8))
    assert torch.equal(I(S.nodes[0]), I(nodes0[0])) and torch.equal(I(S.nodes[2]), I(nodes0[2]))
    st_mix = sc["asset_state"].copy()
    st_mix[[1, 3]] = sc2["asset_state"][[1, 3]]
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], st_mix)
    _, _, _, _, pos, quat = _poses(orc, n, sc, 8)
    kinv, cx, cy = orc.camera_kinv(64, 48, 87.0)
    ref = orc.raycast_camera(64, 48, kinv, 10.0, cx, cy, "depth", pos, quat, tris, sc["tri_seg"])
    got = S.camera(64, 48, kinv, 10.0, cx, cy, 1, pos, quat)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


@pytest.mark.parametrize("n,boxes,mask", [(5, 50, [0, 1, 0, 1, 1]), (3, 106, [1, 1, 1]), (6, 20, [0, 0, 0, 0, 0, 0]), (4, 50, None)])
def test_scene_refresh_in_one_call_equals_the_three_launches(n, boxes, mask):
    """agx_scene_refresh (what AssetManager calls behind a reset: one persistent launch over the dirty envs) against
    agx_scene_transform + agx_bvh_build + agx_boxes_from_assets: world-frame triangles, tree and collision boxes bit for bit,
    clean envs untouched."""
    sc = random_box_scene(n, boxes, seed=5)
    sc2 = random_box_scene(n, boxes, seed=77)
    I = lambda t: t.view(torch.int32)  # noqa: E731
    out = []
    for one_call in (False, True):
        S = Scene(sc)
        p, L = S.L.dptr, S.L
        bx = torch.full((S.na, 11, n), -7.0, device=DEV)
        S.build()  # every env, from the first poses
        L.check(S.lib.agx_boxes_from_assets(n, S.na, p(S.asset_state), p(S.half), None, p(bx), S.stream))
        S.asset_state.copy_(T(sc2["asset_state"]))  # every obstacle moves; only the flagged envs may follow
        mt = T(np.array(mask, np.uint8)) if mask is not None else None
        mk = p(mt) if mt is not None else None
        if one_call:
            L.check(S.lib.agx_scene_refresh(n, S.nt, S.na, p(S.tri_local), p(S.tri_asset), p(S.asset_state), p(S.half), S.ppo, mk,
                                            p(S.tri_world), p(bx), p(S.nodes), p(S.work), S.stream))
        else:
            L.check(S.lib.agx_scene_transform(n, S.nt, S.na, p(S.tri_local), p(S.tri_asset), p(S.asset_state), mk, p(S.tri_world), S.stream))
            L.check(S.lib.agx_bvh_build(n, S.nt, S.ppo, p(S.tri_world), mk, p(S.nodes), p(S.work), S.stream))
            L.check(S.lib.agx_boxes_from_assets(n, S.na, p(S.asset_state), p(S.half), mk, p(bx), S.stream))
        torch.cuda.synchronize()
        out.append((S.tri_world.clone(), S.nodes.clone(), bx.clone()))
    for a, b in zip(*out):
        assert torch.equal(I(a), I(b))
    if mask is not None and any(mask) and not all(mask):  # the refresh really was masked
        S0 = Scene(sc)
        S0.build()
        clean = [e for e, m in enumerate(mask) if not m]
        assert torch.equal(I(out[1][0][clean]), I(S0.tri_world[clean])) and torch.equal(I(out[1][1][clean]), I(S0.nodes[clean]))
        dirty = [e for e, m in enumerate(mask) if m]
        assert not torch.equal(I(out[1][0][dirty]), I(S0.tri_world[dirty]))


@pytest.mark.parametrize("normalize", [True, False])
def test_range_limits_fused_into_the_raycast_equal_the_separate_pass(orc, normalize):
    """AgxRangeLimits: the ray-cast kernel stores the limited / normalised pixel == raw ray-cast + agx_sensor_postprocess
    (and == the restated warp kernel + WarpSensor.apply_range_limits / normalize_observation), camera depth, stereo, LiDAR."""
    from aerial_gym_simulator_amd import _lib

    n = 3
    sc = random_box_scene(n, 60, seed=33)
    S = Scene(sc)
    S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    lim = (1.5, 6.0, 6.0, -1.0, normalize)  # both limits bite in this scene

    def separate(raw):
        t = T(raw)
        _lib.check(S.lib.agx_sensor_postprocess(t.numel(), _lib.dptr(t), None, None, 0.0, 0.0, 0.0, 0.0, 0.0, lim[0], lim[1],
                                                lim[2], lim[3], int(normalize), S.stream))
        torch.cuda.synchronize()
        return t.cpu().numpy()

    _, _, _, _, pos, quat = _poses(orc, n, sc, 4)
    kinv, cx, cy = orc.camera_kinv(53, 37, 87.0)
    raw, seg0 = S.camera(53, 37, kinv, 10.0, cx, cy, 1, pos, quat)
    fused, seg1 = S.camera(53, 37, kinv, 10.0, cx, cy, 1, pos, quat, limits=lim)
    ref_raw, _ = orc.raycast_camera(53, 37, kinv, 10.0, cx, cy, "depth", pos, quat, tris, sc["tri_seg"])
    ref = orc.sensor_postprocess(ref_raw.copy(), lim[0], lim[1], lim[2], lim[3], normalize)
    assert np.array_equal(seg0, seg1) and np.array_equal(fused, separate(raw)) and np.array_equal(fused, ref)
    assert (raw > lim[1]).any() and (raw < lim[0]).any() and ((raw >= lim[0]) & (raw <= lim[1])).any()

    raw, _ = S.stereo(53, 37, kinv, 10.0, 0.095, cx, cy, 0, pos, quat)
    fused, _ = S.stereo(53, 37, kinv, 10.0, 0.095, cx, cy, 0, pos, quat, limits=lim)
    assert np.array_equal(fused, separate(raw))

    _, _, _, _, pos, quat = _poses(orc, n, sc, 6, lidar=True)
    rv = orc.lidar_ray_table(16, 130, -180, 180, -45, 45)
    raw, _ = S.lidar(rv, 10.0, 0, pos, quat)
    fused, _ = S.lidar(rv, 10.0, 0, pos, quat, limits=lim)
    assert np.array_equal(fused, separate(raw))
    # point clouds are limited on the point's norm by agx_sensor_postprocess_points: not fusable, refused
    px = torch.zeros(n, 1, 16, 130, 3, device=DEV)
    tp, tq, trv = T(pos), T(quat), T(rv)
    assert S.lib.agx_raycast_lidar(n, 1, 130, 16, _lib.dptr(trv), 10.0, 2, _lib.dptr(tp), _lib.dptr(tq), _lib.dptr(S.tri_world),
                                   _lib.dptr(S.tri_seg), _lib.dptr(S.nodes), S.nt, _lib.dptr(px), None, S._lim(lim), S.stream) == -1


def test_postprocess_and_image_min(orc):
    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(0)
    px = rng.uniform(0, 12, (4, 1, 48, 64)).astype(np.float32)
    px[rng.random(px.shape) < 0.1] = 1000.0
    z, u = rng.normal(size=px.shape).astype(np.float32), rng.random(px.shape).astype(np.float32)
    for noise in (False, True):
        ref = orc.sensor_postprocess(px.copy(), 0.2, 10.0, 10.0, -10.0, True, z_normal=z if noise else None,
                                     u_dropout=u if noise else None, std_a=1e-3, std_b=2e-3, std_c=1e-3, mean_offset=-0.05,
                                     dropout_prob=0.05)
        t, tz, tu = T(px), T(z), T(u)
        _lib.check(lib.agx_sensor_postprocess(t.numel(), _lib.dptr(t), _lib.dptr(tz) if noise else None,
                                              _lib.dptr(tu) if noise else None, 1e-3, 2e-3, 1e-3, -0.05, 0.05, 0.2, 10.0, 10.0,
                                              -10.0, 1, _lib.current_stream(DEV)))
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), ref)
    img = T(ref)
    out = torch.zeros(4, device=DEV)
    _lib.check(lib.agx_image_min(4, 48 * 64, _lib.dptr(img), _lib.dptr(out), _lib.current_stream(DEV)))
    v = 10.0 * ref.reshape(4, -1)
    v[v < 0] = 10.0
    assert np.array_equal(out.cpu().numpy(), v.min(axis=1))


def test_boxes_from_assets_and_collision(orc):
    from conftest import golden_params, load_golden
    from gpu_harness import DynHarness, to_aos

    from aerial_gym_simulator_amd import _lib

    n = 64
    sc = random_box_scene(n, 100, seed=13)
    K = sc["asset_state"].shape[1]
    lib = _lib.load()
    boxes = torch.zeros(K * 11, n, device=DEV)
    tas, thalf = T(sc["asset_state"]), T(sc["half"])
    _lib.check(lib.agx_boxes_from_assets(n, K, _lib.dptr(tas), _lib.dptr(thalf), None,
                                         _lib.dptr(boxes), _lib.current_stream(DEV)))
    torch.cuda.synchronize()
    ref_boxes = np.concatenate([sc["asset_state"][..., :7], sc["half"]], axis=-1)
    got_boxes = to_aos(boxes).reshape(n, K, 11)
    assert np.array_equal(got_boxes[..., :10], ref_boxes)
    assert np.all(got_boxes[..., 10] >= np.linalg.norm(sc["half"].astype(np.float64), axis=-1))  # conservative bound
    g = load_golden("step_quad_position")
    pd = golden_params(g)
    H = DynHarness(pd, n)
    st = random_robot_states(n, 4, *sc["bounds"])
    H.set(state=st, thrust=np.full((n, 4), 0.6, np.float32), kT=np.full((n, 4), 1.2e-5, np.float32),
          tau_inc=np.full((n, 4), 0.04, np.float32), tau_dec=np.full((n, 4), 0.04, np.float32))
    H.set_gains(*(np.tile(g[k][:1], (n, 1)) for k in ("Kp", "Kv", "KR", "Kw")))
    H.boxes = boxes
    H.rebind()
    act = np.zeros((n, 4), np.float32)
    act[:, :3] = st[:, :3]
    H.substeps(act, 1)
    crashes = np.zeros(n, np.uint8)
    orc.collide_sphere_boxes(pd["collision_radius"], H.get("state"), ref_boxes, crashes)
    assert np.array_equal(H.crashes.cpu().numpy(), crashes.astype(bool))


@pytest.mark.parametrize("device_rng", [False, True])
def test_reset_assets_vs_oracle(orc, device_rng):
    """agx_reset_assets (asset_manager.py:51-71 + env_manager.py:283-295): host-tensor draws and the
    device generator both reproduce the oracle's restatement bit for bit."""
    from aerial_gym_simulator_amd import _lib
    from aerial_gym_simulator_amd._lib import AgxEnvBuffers, AgxResetArgs

    lib = _lib.load()
    n, K, nk, n_obs = 24, 44, 9, 20
    rng = np.random.default_rng(17)
    seed = 0xDEADBEEF12345
    mask = (rng.random(n) < 0.6).astype(np.uint8)
    episodes = rng.integers(0, 9, n).astype(np.int32)
    lo = np.zeros((n, K, 13), np.float32)
    hi = np.zeros((n, K, 13), np.float32)
    lo[..., :3], hi[..., :3] = 0.1, 0.9
    lo[..., 3:6], hi[..., 3:6] = -np.pi, np.pi
    bounds_cfg = ([-2.0, -4.0, -3.0], [-1.0, -2.5, -2.0], [9.0, 2.5, 2.0], [10.0, 4.0, 3.0])
    if device_rng:
        ub = orc.rng_fill(seed, episodes, orc.RNG_BOUNDS, 6)
        sel = orc.rng_fill(seed, episodes, orc.RNG_ASSET_SEL, 1)[:, 0] < 0.15
        u = np.stack([orc.rng_fill(seed, episodes, orc.RNG_ASSETS + a, 6) for a in range(K)], axis=1)
        u = np.concatenate([u, np.zeros((n, K, 7), np.float32)], axis=2)
        u1 = u2 = u
    else:
        ub = rng.random((n, 6)).astype(np.float32)
        sel = rng.random(n) < 0.3
        u1, u2 = rng.random((n, K, 13)).astype(np.float32), rng.random((n, K, 13)).astype(np.float32)
        u = np.where(sel[:, None, None], u2, u1)
    bmin = (np.array(bounds_cfg[1], np.float32) - np.array(bounds_cfg[0], np.float32)) * ub[:, :3] + np.array(bounds_cfg[0], np.float32)
    bmax = (np.array(bounds_cfg[3], np.float32) - np.array(bounds_cfg[2], np.float32)) * ub[:, 3:] + np.array(bounds_cfg[2], np.float32)
    ref = np.zeros((n, K, 13), np.float32)
    ref[..., 6] = 1
    init = ref.copy()
    orc.reset_assets(mask, u, sel.astype(np.uint8), lo, hi, bmin.astype(np.float32), bmax.astype(np.float32), n_obs, nk, ref)
    B, R = AgxEnvBuffers(), AgxResetArgs()
    t_mask, t_flag, t_ep = T(mask), torch.tensor([1, 0], dtype=torch.int32, device=DEV), T(episodes)
    B.reset_mask, B.reset_flag, B.flag_parity, B.episode_count = _lib.dptr(t_mask), _lib.dptr(t_flag), 0, _lib.dptr(t_ep)
    for i in range(3):
        R.lower_bound_min[i], R.lower_bound_max[i] = bounds_cfg[0][i], bounds_cfg[1][i]
        R.upper_bound_min[i], R.upper_bound_max[i] = bounds_cfg[2][i], bounds_cfg[3][i]
    R.seed = seed
    keep = []
    if not device_rng:
        t_ulo, t_uhi, t_us = T(ub[:, :3]), T(ub[:, 3:]), torch.zeros(n, 13, device=DEV)
        keep += [t_ulo, t_uhi, t_us]
        R.u_bounds_lo, R.u_bounds_hi, R.u_state = _lib.dptr(t_ulo), _lib.dptr(t_uhi), _lib.dptr(t_us)
    t_u1, t_u2, t_sel = T(u1), T(u2), T(sel.astype(np.float32))
    t_lo, t_hi, t_st = T(lo), T(hi), T(init)
    _lib.check(lib.agx_reset_assets(B, n, K, R, None if device_rng else _lib.dptr(t_u1), None if device_rng else _lib.dptr(t_u2),
                                    None if device_rng else _lib.dptr(t_sel), _lib.dptr(t_lo), _lib.dptr(t_hi), n_obs, nk,
                                    _lib.dptr(t_st), _lib.current_stream(DEV)))
    torch.cuda.synchronize()
    got = t_st.cpu().numpy()
    assert np.array_equal(got[..., :3], ref[..., :3])           # positions: IEEE + - * only -> bit-exact
    assert np.abs(got[..., 3:7] - ref[..., 3:7]).max() < 3e-7  # quaternions: device sincos vs libm
    assert np.array_equal(got[..., 7:], ref[..., 7:])
    m = mask.astype(bool)
    assert np.array_equal(got[~m], init[~m])
    parked = got[m][..., 0] == -1000.0
    assert parked[:, :9].sum() < parked[:, 20:].sum()  # keep-in-env assets stay, the tail is parked
    assert np.all(parked[:, n_obs:])


@pytest.mark.parametrize("limits", [True, False])
def test_postprocess_points_bit_exact(orc, limits):
    """point-cloud branch of apply_noise / apply_range_limits / normalize_observation"""
    from aerial_gym_simulator_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(1)
    px = rng.uniform(-8, 8, (3, 1, 20, 30, 3)).astype(np.float32)
    px[rng.random(px.shape[:-1]) < 0.1] = 1000.0   # missed rays
    px[rng.random(px.shape[:-1]) < 0.1] *= 0.01    # too close
    z, u = rng.normal(size=px.shape).astype(np.float32), rng.random(px.shape).astype(np.float32)
    for noise in (False, True):
        ref = orc.sensor_postprocess_points(px.copy(), 0.2, 10.0, 10.0, -10.0, limits, True, z_normal=z if noise else None,
                                            u_dropout=u if noise else None, std_a=1e-3, std_b=2e-3, std_c=1e-3,
                                            mean_offset=-0.05, dropout_prob=0.05)
        t, tz, tu = T(px), T(z), T(u)
        _lib.check(lib.agx_sensor_postprocess_points(t.numel() // 3, _lib.dptr(t), _lib.dptr(tz) if noise else None,
                                                     _lib.dptr(tu) if noise else None, 1e-3, 2e-3, 1e-3, -0.05, 0.05, 0.2, 10.0,
                                                     10.0, -10.0, int(limits), 1, _lib.current_stream(DEV)))
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), ref)
    if limits:
        assert (ref == 1.0).any() and (ref == -1.0).any()


@pytest.mark.parametrize("robot", ["base_quadrotor_with_stereo_camera", "base_quadrotor_with_faceid_normal_camera"])
def test_sensor_front_end_stereo_and_normal_robots(orc, robot):
    """The reference's robot names for the f1 sensors build through SimBuilder and their images equal
    the oracle's on the env's own scene and sensor pose (incl. range limits / normalisation)."""
    import random

    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    random.seed(11)       # asset shuffle (asset_loader.py:181 uses python `random`)
    torch.manual_seed(11)  # sensor mount jitter, robot spawn
    n = 3
    env = SimBuilder().build_env("base_sim", "env_with_random_boxes", robot, "lee_velocity_control", DEV, num_envs=n,
                                 args={"rng_seed": 5})
    env.reset()
    a = torch.zeros(n, 4, device=DEV)
    for _ in range(3):
        env.step(actions=a)
        env.post_reward_calculation_step()
    g = env.get_obs()
    sen, sc = env.robot_manager.warp_sensor, env.scene
    cfg = sen.cfg
    tris, tri_seg = sc.tri_world.cpu().numpy(), sc.tri_seg.cpu().numpy()
    pos, quat = sen.sensor_position.cpu().numpy(), sen.sensor_orientation.cpu().numpy()
    kinv, cx, cy = orc.camera_kinv(cfg.width, cfg.height, cfg.horizontal_fov_deg)
    px, seg = g["depth_range_pixels"].cpu().numpy(), g["segmentation_pixels"].cpu().numpy()
    if "stereo" in robot:
        ref, ref_seg = orc.raycast_stereo_camera(cfg.width, cfg.height, kinv, cfg.max_range, cfg.baseline, cx, cy, "depth", pos, quat,
                                                 tris, tri_seg)
        ref = orc.sensor_postprocess(ref, cfg.min_range, cfg.max_range, cfg.far_out_of_range_value, cfg.near_out_of_range_value,
                                     cfg.normalize_range)
        assert px.shape == (n, 1, 270, 480)
        assert (ref == -1.0).mean() > 0.001  # occluded / too close pixels exist (near_out_of_range / max_range = -1)
    else:
        ref, ref_seg = orc.raycast_camera(cfg.width, cfg.height, kinv, cfg.max_range, cx, cy, "normal_world", pos, quat, tris, tri_seg)
        assert px.shape == (n, 1, 270, 480, 3) and ref_seg.max() >= 72
    if not (np.array_equal(seg, ref_seg) and np.array_equal(px, ref)):  # diagnostics for the report
        bad = np.argwhere((seg != ref_seg) | (px != ref).reshape(seg.shape + (-1,)).any(-1))
        e, s_, y, x = bad[0]
        import os

        if os.path.isdir("gpurun_out") or os.environ.get("AGX_DUMP_FAILURES"):  # keep the failing case for analysis
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/fail_case.npz", tris=tris, tri_seg=tri_seg, pos=pos, quat=quat, nodes=sc.bvh_nodes.cpu().numpy(),
                     bad=bad, px=px, seg=seg, ref=ref, ref_seg=ref_seg, kinv=np.array(kinv), cxy=np.array([cx, cy]),
                     wh=np.array([cfg.width, cfg.height]), baseline=getattr(cfg, "baseline", 0.0))
        raise AssertionError(f"{len(bad)} pixels differ, first at env {e} pixel ({y},{x}): got seg {seg[e, s_, y, x]} px {px[e, s_, y, x]}, "
                             f"ref seg {ref_seg[e, s_, y, x]} px {ref[e, s_, y, x]}; sensor pos {pos[e, s_]} robot crashed "
                             f"{g['crashes'].cpu().numpy()}")


def test_axis_parallel_rays(orc):
    """Direction components that are exactly 0 (rcp = inf): the slab test must stay conservative.
    LiDAR rays along +-x, +-y, +-z from points whose coordinates have mixed signs w.r.t. the boxes."""
    n = 6
    sc = random_box_scene(n, 60, seed=3)
    S = Scene(sc)
    S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    rv = np.array([[[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]], [[0, 0, 1], [0, 0, -1], [0.6, 0.8, 0], [0, -0.6, 0.8]]], np.float32)
    rng = np.random.default_rng(0)
    lo, hi = sc["bounds"]
    for trial in range(4):
        pos = rng.uniform(lo + 0.2, hi - 0.2, (n, 1, 3)).astype(np.float32)
        quat = np.tile(np.float32([0, 0, 0, 1]), (n, 1, 1))
        for mode in ("range", "normal_world"):
            ref_px, ref_seg = orc.raycast_lidar(rv, 10.0, mode, pos, quat, tris, sc["tri_seg"])
            got_px, got_seg = S.lidar(rv, 10.0, orc.MODE[mode], pos, quat)
            assert np.array_equal(got_seg, ref_seg) and np.array_equal(got_px, ref_px)
    assert (ref_seg >= 0).mean() > 0.9  # the walls enclose the env: every axis ray hits something


def test_stereo_occlusion_ray_with_zero_direction_component(orc):
    """Captured from a crashed robot 1.6 mm in front of a box face: the occlusion ray towards the stereo partner
    has d_z == 0 exactly; a non-conservative slab test culled the occluder (tests/golden/stereo_axis_parallel_case.npz)."""
    from conftest import load_golden

    g = load_golden("stereo_axis_parallel_case")
    T_ = g["tris"].shape[1]
    sc = dict(tri_local=g["tris"], tri_asset=np.repeat(np.arange(T_ // 12, dtype=np.int32), 12), tri_seg=g["tri_seg"],
              asset_state=np.tile(np.float32([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]), (1, T_ // 12, 1)), half=np.ones((1, T_ // 12, 3), np.float32))
    S = Scene(sc)
    S.build()  # identity asset poses: tri_world == the captured world-frame soup
    assert np.array_equal(S.tri_world.cpu().numpy(), g["tris"])
    W, H = [int(v) for v in g["wh"]]
    cx, cy = [int(v) for v in g["cxy"]]
    ref_px, ref_seg = orc.raycast_stereo_camera(W, H, g["kinv"], 10.0, float(g["baseline"]), cx, cy, "depth", g["pos"], g["quat"], g["tris"], g["tri_seg"])
    got_px, got_seg = S.stereo(W, H, g["kinv"], 10.0, float(g["baseline"]), cx, cy, 1, g["pos"], g["quat"])
    for y, x in g["pixels"]:
        assert ref_seg[0, 0, y, x] == -2 and ref_px[0, 0, y, x] == -1.0  # occluded
    assert np.array_equal(got_seg, ref_seg) and np.array_equal(got_px, ref_px)


def test_full_size_frame_properties(orc):
    """BASELINE config-3 size (8192 envs x 64 x 48 rays, 1272 triangles per env): size-independent properties --
    the frame is deterministic, equals the concatenation of its two 4096-env halves (envs never interact),
    a masked rebuild of half the envs leaves images untouched, and a random subset of envs matches the oracle."""
    n = 8192
    base = random_box_scene(64, 100, seed=5)  # 64 distinct scenes tiled over the batch
    rep = n // 64
    sc = {k: (np.tile(v, (rep,) + (1,) * (v.ndim - 1)) if k in ("tri_local", "tri_seg", "asset_state", "half") else v) for k, v in base.items()}
    S = Scene(sc)
    S.build()
    st = random_robot_states(n, 77, *base["bounds"])
    lp = np.tile(np.float32([0.1, 0.0, 0.03]), (n, 1, 1))
    lq = np.tile(np.float32([0, 0, 0, 1]), (n, 1, 1))
    frame = orc.quat_from_euler(np.deg2rad(np.array([[-90.0, 0.0, -90.0]], np.float32)))[0]
    pos, quat = orc.sensor_pose(st, lp, lq, frame)
    kinv, cx, cy = orc.camera_kinv(64, 48, 87.0)
    px, seg = S.camera(64, 48, kinv, 10.0, cx, cy, 1, pos, quat)
    px2, seg2 = S.camera(64, 48, kinv, 10.0, cx, cy, 1, pos, quat)
    assert np.array_equal(px, px2) and np.array_equal(seg, seg2)  # deterministic
    half = n // 2
    for lo in (0, half):
        sub = {k: (v[lo:lo + half] if k in ("tri_local", "tri_seg", "asset_state", "half") else v) for k, v in sc.items()}
        Sh = Scene(sub)
        Sh.build()
        hp, hs = Sh.camera(64, 48, kinv, 10.0, cx, cy, 1, pos[lo:lo + half], quat[lo:lo + half])
        assert np.array_equal(hp, px[lo:lo + half]) and np.array_equal(hs, seg[lo:lo + half])
    mask = (np.arange(n) % 2).astype(np.uint8)
    S.build(mask)  # rebuild the odd envs from unchanged poses: same trees, same images
    px3, seg3 = S.camera(64, 48, kinv, 10.0, cx, cy, 1, pos, quat)
    assert np.array_equal(px3, px) and np.array_equal(seg3, seg)
    pick = np.random.default_rng(3).choice(n, 24, replace=False)
    tris = orc.scene_transform(sc["tri_local"][pick], sc["tri_asset"], sc["asset_state"][pick])
    rp, rs = orc.raycast_camera(64, 48, kinv, 10.0, cx, cy, "depth", pos[pick], quat[pick], tris, sc["tri_seg"][pick], use_bvh=True)
    assert np.array_equal(px[pick], rp) and np.array_equal(seg[pick], rs)
    assert (seg >= 0).mean() > 0.9


def test_bvh_object_level_order_equals_the_full_key_sort(orc):
    """The builder orders the keys by sorting the OBJECTS and ranking each triangle inside its object; AGX_BVH_FULL_SORT takes
    the 2048-key sort instead.  Same keys, same order, same tree -- bit for bit -- on box scenes with and without walls; and
    with two objects at the SAME place (equal Morton codes: their triangles interleave by face class) the builder notices
    and falls back, again the same tree."""
    FULL = 0x40000000
    for n, k, walls, twin in ((2, 7, False, False), (3, 100, True, False), (2, 7, False, True), (2, 40, True, True)):
        sc = random_box_scene(n, k, seed=11, walls=walls)
        if twin:
            sc["asset_state"][:, 1, 0:7] = sc["asset_state"][:, 0, 0:7]  # the second box sits on the first
        S = Scene(sc)
        assert S.ppo & 0xFFFF == 12
        S.ppo &= ~0x10000000  # the TRIANGLE-level build, object nodes kept (what scenes with non-box primitives use)
        S.build()
        a = S.nodes.clone()
        S.nodes.zero_()
        S.ppo |= FULL
        S.build()
        assert torch.equal(a.view(torch.int32), S.nodes.view(torch.int32)), (n, k, walls, twin)


def _walk_tree(nodes, NI, tris_e, e, nt):
    """-> (times each triangle is reached, object nodes, internal nodes visited); checks every child box on the way"""
    OBJ = 0x40000000
    seen = np.zeros(nt, int)
    stack, visited, objects = [0], 0, 0
    while stack:
        i = stack.pop()
        visited += 1
        assert 0 <= i < nt - 1
        for cslot, sslot, lo, hi in ((3, 11, slice(0, 3), slice(4, 7)), (7, 15, slice(8, 11), slice(12, 15))):
            c, s2 = int(NI[e, i, cslot]), int(NI[e, i, sslot])
            blo, bhi = nodes[e, i, lo], nodes[e, i, hi]
            if c < 0:
                for f in [~c] + ([s2] if s2 >= 0 else []):
                    seen[f] += 1
                    assert (tris_e[f] >= blo - 1e-6).all() and (tris_e[f] <= bhi + 1e-6).all()
            elif c & OBJ:
                f0 = int(NI[e, c & ~OBJ, 15])
                v = tris_e[f0:f0 + 12].reshape(-1, 3)
                assert (v >= blo - 1e-6).all() and (v <= bhi + 1e-6).all()
                seen[f0:f0 + 12] += 1
                objects += 1
            else:
                stack.append(c)
    return seen, objects, visited


def test_object_level_build_with_parked_and_non_box_objects(orc):
    """The object-level build (round 6: the tree over the K objects, csrc/agx_scene.hip bvh_build_objects_env) on a scene that is
    not all boxes in view: half of the obstacles parked at -1000 m (the curriculum's doing: float32 no longer resolves a box there),
    one obstacle deformed (a moved vertex: not a box).  Those keep every triangle reachable through a five-node subtree, the
    rest are object nodes; K - 1 internal nodes over them.  And the frames are the triangle-level tree's, bit for bit."""
    n, k = 3, 20
    sc = random_box_scene(n, k, seed=21, walls=False)
    sc["asset_state"][:, 10:, 0:3] = -1000.0
    sc["tri_local"] = sc["tri_local"].copy()
    sc["tri_local"][:, 12 * 3, 0:3] *= 1.3  # vertex 0 of triangle 0 of object 3
    frames = []
    for box_objects in (True, False):
        S = Scene(sc)
        S.ppo = 12 | (0x30000000 if box_objects else 0)
        S.build()
        if box_objects:
            nodes = S.nodes.cpu().numpy()
            tris = S.tri_world.cpu().numpy().reshape(n, -1, 3, 3)
            for e in range(n):
                seen, objects, visited = _walk_tree(nodes, nodes.view(np.int32), tris[e], e, S.nt)
                assert seen.min() == 1 and seen.max() == 1
                assert objects == 9 and visited == (k - 1) + 5 * (k - 9)
        _, _, _, _, pos, quat = _poses(orc, n, sc, 5)
        kinv, cx, cy = orc.camera_kinv(64, 48, 87.0)
        frames.append(S.camera(64, 48, kinv, 10.0, cx, cy, 1, pos, quat))
        frames.append(S.camera(64, 48, kinv, 3000.0, cx, cy, 0, pos, quat))  # a far plane beyond the parked obstacles
    assert np.array_equal(frames[0][0].view(np.uint32), frames[2][0].view(np.uint32)) and np.array_equal(frames[0][1], frames[2][1])
    assert np.array_equal(frames[1][0].view(np.uint32), frames[3][0].view(np.uint32)) and np.array_equal(frames[1][1], frames[3][1])
    assert (frames[0][1] >= 0).any()


def test_bvh_structure_covers_every_triangle_once(orc):
    """Independent of any ray: the device-built tree reaches every triangle exactly once (one- and two-triangle
    leaves, OBJECT NODES under AGX_BVH_BOX_OBJECTS), every child box contains its triangles, and folded nodes are
    unreachable.  An object node's record is the frame of its box: every vertex of its 12 triangles is a corner of it."""
    OBJ = 0x40000000
    for n, k, walls, tree in ((1, 1, False, 0x30000000), (2, 7, False, 0x30000000), (3, 100, True, 0x30000000), (2, 7, False, 0x20000000),
                              (3, 100, True, 0x20000000)):  # the tree over the objects / over the triangles with object nodes
        sc = random_box_scene(n, k, seed=4, walls=walls)
        S = Scene(sc)
        if S.ppo & 0x20000000:
            S.ppo = (S.ppo & 0xFFFF) | tree
        S.build()
        box_objects = bool(S.ppo & 0x20000000)
        nodes = S.nodes.cpu().numpy()
        NI = nodes.view(np.int32)
        tris = S.tri_world.cpu().numpy().reshape(n, -1, 3, 3)
        nt = S.nt
        for e in range(n):
            seen = np.zeros(nt, int)
            stack, visited, objects = [0], 0, 0
            while stack:
                i = stack.pop()
                visited += 1
                assert 0 <= i < nt - 1
                for cslot, sslot, lo, hi in ((3, 11, slice(0, 3), slice(4, 7)), (7, 15, slice(8, 11), slice(12, 15))):
                    c, s2 = NI[e, i, cslot], NI[e, i, sslot]
                    blo, bhi = nodes[e, i, lo], nodes[e, i, hi]
                    if c < 0:
                        members = [~c] + ([s2] if s2 >= 0 else [])
                        for f in members:
                            assert 0 <= f < nt
                            seen[f] += 1
                            assert (tris[e, f] >= blo - 1e-6).all() and (tris[e, f] <= bhi + 1e-6).all()  # grown by 1e-3
                    elif c & OBJ:
                        assert box_objects and s2 == -1
                        rec, reci = nodes[e, c & ~OBJ], NI[e, c & ~OBJ]
                        f0 = int(reci[15])
                        assert f0 % 12 == 0 and 0 <= f0 < nt
                        axes, h, cen = rec[[0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(3, 3).astype(np.float64), rec[[3, 7, 11]].astype(np.float64), rec[12:15].astype(np.float64)
                        assert np.allclose(axes @ axes.T, np.eye(3), atol=1e-5)
                        v = tris[e, f0:f0 + 12].reshape(-1, 3).astype(np.float64)
                        assert (v >= blo - 1e-6).all() and (v <= bhi + 1e-6).all()
                        loc = (v - cen) @ axes.T
                        assert np.allclose(np.abs(loc), h, atol=1e-5 * (1 + h.sum()))  # every vertex is a corner of the recorded box
                        seen[f0:f0 + 12] += 1
                        objects += 1
                    else:
                        assert s2 == -1
                        stack.append(int(c))
            assert seen.min() == 1 and seen.max() == 1
            assert visited <= nt - 1 and (nt < 24 or visited < 0.7 * (nt - 1))  # box faces fold into two-triangle leaves
            if box_objects and nt >= 24:
                assert objects == nt // 12 and visited == nt // 12 - 1  # the tree ends at the objects: K - 1 internal nodes
