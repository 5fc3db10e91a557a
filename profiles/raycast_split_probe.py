"""k_raycast: how many workgroups share one (env, sensor) image, with the XCD-aware workgroup mapping (csrc/agx_raycast.hip).
agx_set_option("ray_split", n) overrides the launch policy; frames must be bit-identical for every value.  The state is frozen (no env step
between measurements) and the candidates are timed round-robin, several rounds, so that clock drift hits all of them alike.
    python profiles/raycast_split_probe.py [depth|lidar] """
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "depth"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (8192 if which == "depth" else 4096)
task = bench.make_task(which, n, "cuda:0", False)
task.reset()
A = task.task_config.action_space_dim
a = torch.rand(n, A, device="cuda:0") * 2 - 1
for _ in range(3):
    task.step(a)
torch.cuda.synchronize()
sensor = task.sim_env.robot_manager.warp_sensor
g = task.sim_env.global_tensor_dict
splits = [1, 2, 3, 4, 6, 12] if which == "depth" else [1, 2, 4, 8, 16, 32, 64]
if os.environ.get("AGX_PROBE_SPLITS"):  # e.g. a build with another workgroup size (AGX_LIB_PATH): "6,8,12,16,24,48"
    splits = [int(x) for x in os.environ["AGX_PROBE_SPLITS"].split(",")]


def use(sp):
    from aerial_gym_simulator_amd import _lib

    _lib.set_option("ray_split", sp)  # (rounds 4-5 ran this script with the AGX_RAY_SPLIT variable the library read then)


ref, same = None, {}
for sp in splits:
    use(sp)
    g["depth_range_pixels"].fill_(-5.0)
    g["segmentation_pixels"].fill_(-5)
    sensor.raycast(fuse_limits=sensor.limits_fusable())
    torch.cuda.synchronize()
    img = (g["depth_range_pixels"].clone(), g["segmentation_pixels"].clone())
    ref = ref or img
    same[sp] = bool(torch.equal(img[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(img[1], ref[1]))
times = {sp: [] for sp in splits}
for rnd in range(5):
    for sp in splits:
        use(sp)
        times[sp].append(bench.kernel_time_raycast(task, reps=10) * 1e6)
for sp in splits:
    t = sorted(times[sp])
    print(json.dumps({"workload": which, "envs": n, "split": sp, "raycast_us_median": t[len(t) // 2], "raycast_us_min": t[0], "raycast_us_max": t[-1],
                      "frames_identical": same[sp]}), flush=True)
