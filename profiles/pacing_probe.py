"""Sustained eager stepping (8192 envs, position task): does it matter how far the host runs ahead of the device?
python profiles/pacing_probe.py   (GPU box)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    device = "cuda:0"
    task = bench.make_task("dynamics", 8192, device, False)
    task.reset()
    g = torch.Generator(device=device).manual_seed(1234)
    actions = [torch.rand(8192, 4, device=device, generator=g) * 2 - 1 for _ in range(16)]
    torch.cuda.synchronize()
    for i in range(200):
        task.step(actions[i % 16])
    torch.cuda.synchronize()
    K = 4000

    def run(label, body):
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            body()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            r = ((t2 - t0) / K * 1e6, (t2 - t1) * 1e6)
            best = r if best is None or r[0] < best[0] else best
        print(json.dumps({"variant": label, "us_per_step": round(best[0], 3), "host_ahead_at_end_us": round(best[1], 1)}), flush=True)

    def plain():
        for i in range(K):
            task.step(actions[i & 15])

    run("free running", plain)
    for M in (10, 20, 50, 100, 400):
        def synced(M=M):
            for i in range(K):
                task.step(actions[i & 15])
                if i % M == M - 1:
                    torch.cuda.synchronize()
        run(f"device synchronize every {M} steps", synced)
    for every, lag in ((8, 2), (16, 2), (16, 4), (32, 2), (64, 2)):
        evs = [torch.cuda.Event() for _ in range(lag + 1)]

        def paced(every=every, lag=lag, evs=evs):
            n = 0
            for i in range(K):
                task.step(actions[i & 15])
                if i % every == every - 1:
                    evs[n % (lag + 1)].record()
                    n += 1
                    if n > lag:
                        evs[(n - lag - 1) % (lag + 1)].synchronize()  # the host stays <= (lag + 1) * every steps ahead
        run(f"event every {every} steps, host at most {(lag + 1) * every} steps ahead", paced)
    run("free running (again)", plain)


if __name__ == "__main__":
    main()
