"""Stress: stereo / basic camera vs oracle on many random scenes (diagnostic tool, not a pytest)."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo")
import oracle as orc
from scene_util import random_box_scene
from test_gpu_raycast import Scene, _poses

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    n = 4
    sc = random_box_scene(n, 100, seed=seed)
    S = Scene(sc); S.build()
    tris = orc.scene_transform(sc["tri_local"], sc["tri_asset"], sc["asset_state"])
    _, _, _, _, pos, quat = _poses(orc, n, sc, seed + 1000)
    kinv, cx, cy = orc.camera_kinv(96, 54, 87.0)
    for mode in ("depth", "pointcloud_world"):
        ref_px, ref_seg = orc.raycast_stereo_camera(96, 54, kinv, 10.0, -0.095, cx, cy, mode, pos, quat, tris, sc["tri_seg"])
        got_px, got_seg = S.stereo(96, 54, kinv, 10.0, -0.095, cx, cy, orc.MODE[mode], pos, quat)
        if not (np.array_equal(got_seg, ref_seg) and np.array_equal(got_px, ref_px)):
            idx = np.argwhere(got_seg != ref_seg)
            print("MISMATCH seed", seed, mode, "count", len(idx), "first", idx[:3].tolist())
            for e, s_, y, x in idx[:3]:
                print("   got", got_seg[e, s_, y, x], got_px[e, s_, y, x], "ref", ref_seg[e, s_, y, x], ref_px[e, s_, y, x])
            bad += 1
    m_px, m_seg = S.camera(96, 54, kinv, 10.0, cx, cy, 1, pos, quat)
    r_px, r_seg = orc.raycast_camera(96, 54, kinv, 10.0, cx, cy, "depth", pos, quat, tris, sc["tri_seg"])
    if not (np.array_equal(m_seg, r_seg) and np.array_equal(m_px, r_px)):
        print("MONO MISMATCH seed", seed); bad += 1
print("done, mismatching cases:", bad)
