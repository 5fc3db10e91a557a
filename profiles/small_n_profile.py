"""Kernel breakdown in the host / latency-bound regime (256 envs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
wl, n = sys.argv[1], int(sys.argv[2])
t = bench.make_task(wl, n, "cuda:0", False, obstacles="curriculum")
t.reset()
a = torch.rand(n, 4, device="cuda:0") * 2 - 1
for _ in range(200):
    t.step(a)
torch.cuda.synchronize()
