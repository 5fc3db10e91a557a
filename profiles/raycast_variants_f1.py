"""k_raycast variants on the config-3 scene (8192 envs, 64x48, 100 boxes + 6 walls): BASIC depth+seg vs
NORMAL+faceID vs STEREO (second any-hit ray), and the LiDAR kernels of configs 4 / f2.  HIP events."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from aerial_gym_simulator_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda:0"
task = bench.make_task("depth", n, dev, False)
task.reset()
a = torch.rand(n, 4, device=dev) * 2 - 1
for _ in range(5):
    task.step(a)
env = task.sim_env
sen, sc = env.robot_manager.warp_sensor, env.scene
lib, p = env._lib, _lib.dptr
cfg = sen.cfg
N, S, W, H = n, 1, cfg.width, cfg.height
px1 = torch.zeros(N, S, H, W, device=dev)
px3 = torch.zeros(N, S, H, W, 3, device=dev)
seg = torch.zeros(N, S, H, W, dtype=torch.int32, device=dev)


def timeit(name, fn, rays, reps=10):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        s0.record()
        st = env._stream()
        for _ in range(reps):
            _lib.check(fn(st))
        s1.record()
        torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(s1) / reps)
    print(f"{name:44s} {best:8.3f} ms/frame  {rays / best / 1e6:8.2f} G rays/s (primary)")


common = (p(sen.sensor_position), p(sen.sensor_orientation), p(sc.tri_world), p(sc.tri_seg), p(sc.bvh_nodes), sc.num_tris)
rays = N * S * H * W
timeit("camera depth + seg (BASIC)", lambda st: lib.agx_raycast_camera(N, S, W, H, sen.kinv, 10.0, sen.c_x, sen.c_y, 1, *common, p(px1), p(seg), None, st), rays)
timeit("camera pointcloud world + seg (BASIC)", lambda st: lib.agx_raycast_camera(N, S, W, H, sen.kinv, 10.0, sen.c_x, sen.c_y, 3, *common, p(px3), p(seg), None, st), rays)
timeit("camera normal world + faceID (NORMAL)", lambda st: lib.agx_raycast_camera(N, S, W, H, sen.kinv, 10.0, sen.c_x, sen.c_y, 5, *common, p(px3), p(seg), None, st), rays)
timeit("stereo depth + seg, baseline 0.095 (STEREO)", lambda st: lib.agx_raycast_stereo_camera(N, S, W, H, sen.kinv, 10.0, 0.095, sen.c_x, sen.c_y, 1, *common, p(px1), p(seg), None, st), rays)
