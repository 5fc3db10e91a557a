"""Where the per-step exchange time goes on one GPU (world of one): env steps alone, + post only, + post and wait.
    AGX_BENCH_FORCE_DIST unnecessary; run on the GPU box:  python profiles/exchange_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from aerial_gym_simulator_amd import _lib  # noqa: E402
from aerial_gym_simulator_amd.sharding import StepGather  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = "cuda:0"
WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "dynamics"
task = bench.make_task(WORKLOAD, 8192, dev, False)
task.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = [torch.rand(8192, 4, device=dev, generator=g) * 2 - 1 for _ in range(16)]
OBS = task.task_obs["observations"].shape[1]
STEPS = 3000 if WORKLOAD == "dynamics" else 100
gb = StepGather(8192, OBS, dev, env=task.sim_env, reward=task.rewards, backend="rccl_thread")
lib, h = gb._lib, gb._native
env = task.sim_env


def run(mode, steps=None):
    steps = steps or STEPS
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            task.step(acts[i & 15])
            p = env._parity
            s = torch.cuda.current_stream().cuda_stream
            if mode == "record":  # what the post costs the stepping stream: one event record, nothing else
                ev.record()
            elif mode == "api":
                gb.exchange(p, overlap=True)
            elif mode == "api-event":
                gbe.exchange(p, overlap=True)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
    print(f"{mode:10s} host {1e6 * host / steps:6.2f} us/step   total {1e6 * tot / steps:6.2f} us/step")


ev = torch.cuda.Event()
for m in ("none", "record", "api", "none"):
    run(m)
gb.close()
gbe = StepGather(8192, OBS, dev, env=task.sim_env, reward=task.rewards, backend="rccl_thread", ready="event")
for m in ("api-event", "none"):
    run(m)
gbe.close()
dist.destroy_process_group()
