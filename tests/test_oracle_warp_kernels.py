"""The C oracle's ray-cast restatement (oracle/oracle_raycast.c) vs frames produced by EXECUTING the reference's own kernel
source and sensor classes (warp_camera_kernels.py:13-282, warp_lidar_kernels.py:13-194, warp_stereo_camera_kernels.py:13-299,
warp_cam.py / warp_lidar.py / warp_stereo_cam.py / warp_normal_faceID_*.py / warp_sensor.py) under the `warp` emulation of
oracle/wp_emul.py.  Bit for bit: distances, point clouds, normals, segmentation ids, face ids -- the emulated built-ins and the
oracle are the same sequence of binary32 operations; where that could not hold a tolerance would be stated here (none is)."""
import os
import subprocess
import sys

import numpy as np
import pytest
from warp_golden_util import bits, cases, cfg_of, kind_of, limits_of, load, mode_of, seg_mask


def _postprocess(orc, g, tag, raw):
    lim = limits_of(g, tag)
    if lim is None:
        return raw
    c = cfg_of(g, tag)
    px = raw.copy()
    if c["return_pointcloud"]:
        orc.sensor_postprocess_points(px, lim[0], lim[1], lim[2], lim[3], limits=not c["world_frame"], normalize=bool(lim[4]))
    else:
        orc.sensor_postprocess(px, lim[0], lim[1], lim[2], lim[3], bool(lim[4]))
    return px


def _check(orc, g, tag, px, seg):
    raw = g[tag + "_raw"]
    assert px.shape == raw.shape
    assert np.array_equal(bits(px), bits(raw)), (tag, int((bits(px) != bits(raw)).sum()), float(np.abs(px - raw).max()))
    if tag + "_seg" in g.files:
        m = seg_mask(g, tag)
        ref = g[tag + "_seg"]
        assert np.array_equal(seg[m], ref[m]) if m is not None else np.array_equal(seg, ref), tag
    final = _postprocess(orc, g, tag, px)
    assert np.array_equal(bits(final), bits(g[tag + "_final"])), tag


@pytest.mark.parametrize("tag", cases("camera"))
def test_camera_kernels_vs_reference_source(orc, tag):
    g = load("camera")
    c = cfg_of(g, tag)
    px, seg = orc.raycast_camera(int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], int(g[tag + "_cxy"][0]), int(g[tag + "_cxy"][1]),
                                 mode_of(g, tag, "camera"), g[tag + "_sensor_position"], g[tag + "_sensor_orientation"], g["tri_world"], g["tri_seg"])
    _check(orc, g, tag, px, seg)


@pytest.mark.parametrize("tag", cases("lidar"))
def test_lidar_kernels_vs_reference_source(orc, tag):
    g = load("lidar")
    c = cfg_of(g, tag)
    px, seg = orc.raycast_lidar(g[tag + "_ray_vectors"], c["max_range"], mode_of(g, tag, "lidar"), g[tag + "_sensor_position"],
                                g[tag + "_sensor_orientation"], g["tri_world"], g["tri_seg"])
    _check(orc, g, tag, px, seg)


@pytest.mark.parametrize("tag", cases("stereo"))
def test_stereo_kernels_vs_reference_source(orc, tag):
    g = load("stereo")
    c = cfg_of(g, tag)
    px, seg = orc.raycast_stereo_camera(int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], c["baseline"], int(g[tag + "_cxy"][0]),
                                        int(g[tag + "_cxy"][1]), mode_of(g, tag, "stereo"), g[tag + "_sensor_position"], g[tag + "_sensor_orientation"],
                                        g["tri_world"], g["tri_seg"])
    _check(orc, g, tag, px, seg)


def _oracle_frame(orc, g, tag, kind, **kw):
    c = cfg_of(g, tag)
    mode = mode_of(g, tag, kind)
    pose = (g[tag + "_sensor_position"], g[tag + "_sensor_orientation"], g["tri_world"], g["tri_seg"])
    if kind == "camera":
        return orc.raycast_camera(int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], int(g[tag + "_cxy"][0]), int(g[tag + "_cxy"][1]), mode, *pose, **kw)
    if kind == "lidar":
        return orc.raycast_lidar(g[tag + "_ray_vectors"], c["max_range"], mode, *pose, **kw)
    return orc.raycast_stereo_camera(int(c["width"]), int(c["height"]), g[tag + "_kinv"], c["max_range"], c["baseline"], int(g[tag + "_cxy"][0]),
                                     int(g[tag + "_cxy"][1]), mode, *pose, **kw)


@pytest.mark.parametrize("tag", cases("boxes"))
def test_box_scene_kernels_vs_reference_source(orc, tag):
    """obstacles in trimesh.creation.box's vertex / face order -- what the reference's loader hands Warp
    (assets/warp_asset.py:19-24) and what the product's BVH builder turns into object nodes.  The oracle (brute force, and through
    its own BVH) vs the reference's kernels executed over that scene: shared face planes, an origin ON a face (t = +-0), an origin
    inside a box, rays through corners / edge midpoints, rays parallel to a face 0.5 mm away."""
    g = load("boxes")
    kind = kind_of(g, tag)
    px, seg = _oracle_frame(orc, g, tag, kind)
    _check(orc, g, tag, px, seg)
    px2, seg2 = _oracle_frame(orc, g, tag, kind, use_bvh=True)
    assert np.array_equal(bits(px2), bits(px)) and np.array_equal(seg2, seg)


def test_box_fixture_holds_what_it_was_designed_for():
    from scene_util import BOX_FACES

    g = load("boxes")
    assert np.array_equal(g["faces"][:12], BOX_FACES)  # trimesh.creation.box's face list, the one the builder recognises
    raw, seg = g["cam_depth_seg_zero_mount_raw"], g["cam_depth_seg_zero_mount_seg"]
    base = 100 + 12 * np.arange(7)
    # env 1: the two boxes whose front faces are coplanar both show, the seam between them runs down the principal column
    assert {int(base[1]), int(base[1]) + 1} <= set(seg[1].ravel().tolist()) and np.allclose(raw[1][seg[1] == base[1]], 2.5, atol=1e-6)
    # env 2: origin in the plane of a face: hits at t = +0 and t = -0, and rays for which the zero was rejected
    r2 = raw[2][seg[2] == base[2]]
    assert (bits(r2) == 0).any() and (bits(r2) == 0x80000000).any() and (r2 > 1.0).any()
    # env 3: origin inside a box: every ray hits it from the inside, below min_range
    assert (seg[3] == base[3]).all() and (raw[3] < 0.5).all() and (g["cam_depth_seg_zero_mount_final"][3] == -1.0).any()
    # env 6: the row that runs 0.5 mm above the top face passes it and reaches the wall; the row below lands on it
    assert seg[6, 0, 6, 8] == base[6] + 2 and seg[6, 0, 7, 8] == base[6]
    # face ids: every face pair of trimesh's box appears
    fid = g["cam_normal_world_zero_mount_seg"]
    assert len(set((fid[fid >= 0] % 12).tolist())) >= 10
    names = {str(g[t + "_kernel"]) for t in cases("boxes")}
    assert len(names) >= 9, sorted(names)


def test_fixtures_cover_every_kernel_and_every_pixel_class():
    """all 14 kernels of the three reference modules ran; the frames hold hits, misses, hits below min_range (sensor inside an
    obstacle), hits beyond max_range is impossible by construction of the query -- but far-plane misses behind which geometry
    lies are there (env 2), and the stereo frames hold occluded (-1) pixels"""
    names = set()
    for kind in ("camera", "lidar", "stereo"):
        g = load(kind)
        names |= {str(g[t + "_kernel"]) for t in cases(kind)}
    assert len(names) == 14, sorted(names)
    g = load("camera")
    raw, seg = g["range_seg_raw"], g["range_seg_seg"]
    assert (raw == 1000.0).any() and (seg == -2).any() and (raw < 0.2).any() and ((raw > 0.2) & (raw < 10.0)).any()
    assert (seg > 70000).any() and ((seg >= 100) & (seg < 130)).any()
    assert (g["range_seg_final"] == -1.0).any() and (g["range_seg_final"] == 1.0).any()  # near / far out of range after normalisation
    s = load("stereo")
    assert (s["range_wide_baseline_raw"] == -1.0).any() and (s["range_wide_baseline_raw"] == 1000.0).any()
    l = load("lidar")
    assert l["points_seg_seg_undefined"].any() and not l["range_seg_seg_undefined"].any()
    assert (l["normal_world_seg"] == -1).any() and (l["normal_world_seg"] >= 0).any()  # face ids


def test_host_intrinsics_match_the_reference_under_warp_arithmetic(orc):
    """K_inv as warp_cam.py:43-61 gets it -- wp.mat44 rounds K to float32, wp.inverse inverts THAT -- vs the product's host code
    and the oracle's camera_kinv"""
    from aerial_gym_simulator_amd.sensors.hip_sensor import pinhole_kinv

    for kind in ("camera", "stereo"):
        g = load(kind)
        for tag in cases(kind):
            c = cfg_of(g, tag)
            k, cx, cy = pinhole_kinv(int(c["width"]), int(c["height"]), c["hfov_deg"])
            assert np.array_equal(bits(np.array(k, np.float32)), bits(g[tag + "_kinv"])), (tag, list(k), g[tag + "_kinv"])
            assert (cx, cy) == tuple(int(v) for v in g[tag + "_cxy"])
            k2, cx2, cy2 = orc.camera_kinv(int(c["width"]), int(c["height"]), c["hfov_deg"])
            assert np.array_equal(bits(k2), bits(g[tag + "_kinv"])) and (cx2, cy2) == (cx, cy)


def test_warp_kernel_goldens_are_reproducible_from_the_reference(tmp_path):
    """provenance: the committed generator, run against /root/reference (build container only), rewrites the four fixtures
    bit for bit"""
    from conftest import ROOT

    if not os.path.isdir("/root/reference/aerial_gym"):
        pytest.skip("the reference tree is not on this machine")
    code = ("import sys; sys.path.insert(0, %r)\nimport gen_golden_warp_kernels as g\ng.OUT = %r\ng.main()\ng.main_boxes()\n" % (os.path.join(ROOT, "oracle"), str(tmp_path)))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, timeout=600)
    for kind in ("camera", "lidar", "stereo", "boxes"):
        new, old = np.load(os.path.join(str(tmp_path), "warp_kernels_%s.npz" % kind)), load(kind)
        assert sorted(new.files) == sorted(old.files)
        for k in new.files:
            assert (str(new[k]) == str(old[k])) if new[k].dtype.kind in "US" else np.array_equal(new[k], old[k], equal_nan=True), (kind, k)


def test_emulator_query_equals_the_c_oracle_query_ray_by_ray(orc):
    """The fixtures' closest hits come from oracle/wp_emul.py's own brute-force loop (numpy float32, correctly rounded fmaf emulation),
    written from Warp's intersect.h independently of oracle_raycast.c.  Ray by ray the two restatements agree -- hit / miss, t to the
    bit, face index -- on random rays, on rays aimed EXACTLY at vertices and edge midpoints (edge functions that are exactly 0: the
    double-precision fallback; shared edges: the tie rule), on axis-parallel rays, and on rays that start inside a box."""
    import wp_emul  # noqa: PLC0415  (oracle/ is on sys.path: conftest)

    g = load("camera")
    rng = np.random.default_rng(5)
    n_checked = n_hits = n_zero_edge = 0
    for env in range(3):
        tris = np.ascontiguousarray(g["tri_world"][env])
        verts = tris.reshape(-1, 3)
        rays = []
        for _ in range(250):  # random origin / direction
            rays.append((rng.uniform(-4, 4, 3), rng.normal(size=3), rng.choice([3.0, 10.0, 50.0])))
        for _ in range(150):  # through a vertex, or the midpoint of a triangle edge, exactly representable targets
            o = rng.uniform(-4, 4, 3).astype(np.float32)
            f = rng.integers(0, tris.shape[0])
            a, b = tris[f, 0:3], tris[f, 3:6]
            tgt = a if rng.random() < 0.5 else (a + b) * np.float32(0.5)
            rays.append((o, (tgt - o), 50.0))
        for ax in range(3):  # axis-parallel rays (two direction components exactly 0)
            for _ in range(30):
                d = np.zeros(3); d[ax] = rng.choice([-1.0, 1.0])
                rays.append((rng.uniform(-4, 4, 3), d, 20.0))
        for _ in range(90):  # axis-parallel THROUGH a vertex, exactly: two of its sheared coordinates are 0, edge functions vanish
            v = verts[rng.integers(0, verts.shape[0])]
            ax = int(rng.integers(0, 3))
            d = np.zeros(3, np.float32); d[ax] = rng.choice([-1.0, 1.0])
            o = v.copy(); o[ax] = v[ax] - np.float32(3.0) * d[ax]  # (the two other coordinates equal the vertex's bit for bit)
            rays.append((o, d, 20.0))
        for k in range(30):  # from inside an obstacle
            c = g["asset_pose"][env][k % g["asset_pose"].shape[1], 0:3]
            rays.append((c + rng.normal(scale=0.01, size=3), rng.normal(size=3), 10.0))
        for o, d, max_t in rays:
            o = np.asarray(o, np.float32)
            d = np.asarray(d, np.float32)
            d = (d / np.sqrt((d * d).sum(dtype=np.float32))).astype(np.float32)
            h1, t1, f1 = wp_emul._query_brute_force(o, d, np.float32(max_t), tris)
            h2, t2, f2 = orc.mesh_query_ray(o, d, np.float32(max_t), tris)
            assert h1 == h2, (env, o, d)
            if h1:
                assert np.float32(t1).view(np.uint32) == np.float32(t2).view(np.uint32) and f1 == f2, (env, o, d, t1, t2, f1, f2)
                n_hits += 1
            n_checked += 1
    assert n_checked > 1700 and n_hits > 500
