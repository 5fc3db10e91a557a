"""Why a 20-step timed region reads 14-15 us per step where a 2000-step one reads 12.4 (bench.py --steps 20 --warmup 5, the driver's
form): region time against the number of steps, and the bench's own sequence (reset, three regions with synchronised episodes,
de-synchronise, regions of 20) continued for 30 regions."""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import bench

task = bench.make_task("dynamics", 8192, "cuda:0", False)
task.reset()
g = torch.Generator(device="cuda:0").manual_seed(1)
acts = [torch.rand(8192, 4, device="cuda:0", generator=g) * 2 - 1 for _ in range(16)]
out = []
for r in range(3):
    t = bench.timed_steps(task, acts, 20, 5 if r == 0 else 0, 1)
    out.append(("sync", r, t * 1e6 / 20, bench.timed_steps.last_host_s * 1e6 / 20))
bench.desynchronise_episodes(task)
for r in range(30):
    t = bench.timed_steps(task, acts, 20, 5 if r == 0 else 0, 1)
    out.append(("desync", r, t * 1e6 / 20, bench.timed_steps.last_host_s * 1e6 / 20))
for o in out:
    print("%s region %2d  %.2f us/step  host %.2f" % o)
for K in (5, 10, 20, 40, 100, 400):
    ts, hs = [], []
    for r in range(7):
        ts.append(bench.timed_steps(task, acts, K, 0, 1))
        hs.append(bench.timed_steps.last_host_s)
    ts.sort(), hs.sort()
    print("K %3d  region %.1f us  per step %.2f  host per step %.2f" % (K, ts[3] * 1e6, ts[3] * 1e6 / K, hs[3] * 1e6 / K))
