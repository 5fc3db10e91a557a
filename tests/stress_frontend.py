"""Diagnostic: repeat test_sensor_front_end_stereo_and_normal_robots with fresh random scenes."""
import sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo")
import random
import oracle as orc
import test_gpu_raycast as T

fails = 0
for it in range(int(sys.argv[1])):
    random.seed(it)
    import torch
    torch.manual_seed(it)
    for robot in ("base_quadrotor_with_stereo_camera", "base_quadrotor_with_faceid_normal_camera"):
        try:
            T.test_sensor_front_end_stereo_and_normal_robots(orc, robot)
        except AssertionError as e:
            fails += 1
            print("FAIL it", it, robot, str(e)[:700])
print("iterations", sys.argv[1], "fails", fails)
