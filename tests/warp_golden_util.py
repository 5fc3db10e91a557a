"""Shared by the CPU and GPU tests of tests/golden/warp_kernels_{camera,lidar,stereo,boxes}.npz -- frames the REFERENCE'S OWN kernel
bodies and sensor classes produced under oracle/wp_emul.py (generator: oracle/gen_golden_warp_kernels.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(kind):
    return np.load(os.path.join(GOLDEN, "warp_kernels_%s.npz" % kind))


def cases(kind):
    g = load(kind)
    return [k[: -len("_kernel")] for k in g.files if k.endswith("_kernel")]


def kind_of(g, tag, default=None):
    """warp_kernels_boxes.npz holds camera, LiDAR and stereo cases in one file: `<tag>_kind` says which"""
    return str(g[tag + "_kind"]) if tag + "_kind" in g.files else default


def cfg_of(g, tag):
    return dict(zip([str(x) for x in g["cfg_layout"]], [float(v) for v in g[tag + "_cfg"]]))


def mode_of(g, tag, kind):
    c = cfg_of(g, tag)
    kern = str(g[tag + "_kernel"])
    if "normal_faceID" in kern:
        return "normal_world" if c["normal_world"] else "normal"
    if c["return_pointcloud"]:
        return "pointcloud_world" if c["world_frame"] else "pointcloud"
    if kind == "lidar":
        return "range"
    return "depth" if c["calculate_depth"] else "range"


def seg_mask(g, tag):
    """pixels whose segmentation value the reference's kernel defines (warp_lidar_kernels.py:49-86 stores a never-assigned variable
    on a miss: recorded by the generator, excluded here)"""
    if tag + "_seg_undefined" in g.files:
        return ~g[tag + "_seg_undefined"]
    return None


def limits_of(g, tag):
    """(min_range, max_range, far_oor, near_oor, normalize) or None where WarpSensor.update applies no limits
    (normal_faceID sensors: warp_sensor.py:193-195)"""
    c = cfg_of(g, tag)
    if "normal_faceID" in str(g[tag + "_kernel"]):
        return None
    return (c["min_range"], c["max_range"], c["far_oor"], c["near_oor"], int(c["normalize"]))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
