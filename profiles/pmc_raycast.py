"""Workload for counter passes on k_raycast: BASELINE configs[2] (all obstacles) or configs[3], a few frames.
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -o p -- python profiles/pmc_raycast.py [depth|lidar] [envs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
which = sys.argv[1] if len(sys.argv) > 1 else "depth"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (8192 if which == "depth" else 4096)
if os.environ.get("AGX_PROBE_RAY_SPLIT"):  # (this SCRIPT's knob, measure_r04_c.sh; the library itself reads no environment variable)
    from aerial_gym_simulator_amd import _lib
    _lib.set_option("ray_split", int(os.environ["AGX_PROBE_RAY_SPLIT"]))
t = bench.make_task(which, n, "cuda:0", False)
t.reset()
a = torch.rand(n, t.task_config.action_space_dim, device="cuda:0") * 2 - 1
for _ in range(4):
    t.step(a)
torch.cuda.synchronize()
print("done")
