"""Workload for SQ counter passes on k_raycast: config 3 (all obstacles), a few frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
n = 8192
t = bench.make_task("depth", n, "cuda:0", False)
t.reset()
a = torch.rand(n, 4, device="cuda:0") * 2 - 1
for _ in range(4):
    t.step(a)
torch.cuda.synchronize()
print("done")
