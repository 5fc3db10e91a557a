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
from warp_golden_util import bits, cases, cfg_of, limits_of, load, mode_of, seg_mask


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
    """provenance: the committed generator, run against /root/reference (build container only), rewrites the three fixtures
    bit for bit"""
    from conftest import ROOT

    if not os.path.isdir("/root/reference/aerial_gym"):
        pytest.skip("the reference tree is not on this machine")
    code = ("import sys; sys.path.insert(0, %r)\nimport gen_golden_warp_kernels as g\ng.OUT = %r\ng.main()\n" % (os.path.join(ROOT, "oracle"), str(tmp_path)))
    subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, timeout=600)
    for kind in ("camera", "lidar", "stereo"):
        new, old = np.load(os.path.join(str(tmp_path), "warp_kernels_%s.npz" % kind)), load(kind)
        assert sorted(new.files) == sorted(old.files)
        for k in new.files:
            assert (str(new[k]) == str(old[k])) if new[k].dtype.kind in "US" else np.array_equal(new[k], old[k], equal_nan=True), (kind, k)
