"""Where the per-step cost of the exchange goes, world of one, one GPU (run plain, or under rocprofv3 --kernel-trace --stats):
    python profiles/push_probe.py none|rows|peer_push|rccl_thread [steps]
`rows`: exchange rows + step_signal bound, nothing gathered (what bench.py's ms_per_step_without_exchange measures)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "peer_push"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
dev = "cuda:0"
from aerial_gym_simulator_amd.sharding import StepGather  # noqa: E402

task = bench.make_task("dynamics", 8192, dev, False)
task.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = [torch.rand(8192, 4, device=dev, generator=g) * 2 - 1 for _ in range(16)]
gb = None
if mode in ("peer_push", "rccl_thread"):
    gb = StepGather(8192, 13, dev, env=task.sim_env, reward=task.rewards, backend=mode)
elif mode == "rows":
    gb0 = StepGather(8192, 13, dev, env=task.sim_env, reward=task.rewards, backend="process_group")
for rep in range(3):
    dt = bench.timed_steps(task, acts, steps, 200, 1, gb, overlap=True)
    print(f"{mode}: {1e6 * dt / steps:.2f} us per step", flush=True)
if gb is not None:
    gb.close()
dist.destroy_process_group()
