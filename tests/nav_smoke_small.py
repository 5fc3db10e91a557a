"""Host-bound regime: step time of the navigation-type tasks at RL-typical env counts."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
for wl in ("depth", "lidar_nav"):
    for n in (256, 1024):
        t = bench.make_task(wl, n, "cuda:0", False, obstacles="curriculum")
        t.reset()
        a = torch.rand(n, 4, device="cuda:0") * 2 - 1
        for _ in range(30):
            t.step(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            t.step(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 200
        print(f"{wl:10s} n={n:5d}: {dt*1e3:.3f} ms/step  {n/dt/1e3:.0f} k env-steps/s")
        del t
