"""The first 20-step region of a process (bench.py --steps 20 --warmup 5) under different preparations; one mode per
process:  python profiles/short_run_probe2.py MODE   (see MODES)"""
import gc
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from short_run_probe import region  # noqa: E402

MODES = ["baseline", "nogc", "burst", "burst_nosync", "burst32", "burst_late", "warm25", "warm100", "warm5_sync_warm5", "idle", "twice"]


def main():
    mode = sys.argv[1]
    device = "cuda:0"
    task = bench.make_task("dynamics", 8192, device, False)
    task.reset()
    g = torch.Generator(device=device).manual_seed(1234)
    actions = [torch.rand(8192, 4, device=device, generator=g) * 2 - 1 for _ in range(16)]
    def burst(n, sync=True):  # n trivial launches on the stream the steps use
        x = torch.zeros(64, device=device)
        for _ in range(n):
            x.add_(1.0)
        if sync:
            torch.cuda.synchronize()

    if mode == "burst":
        burst(256)
    if mode == "burst_nosync":
        burst(256, False)
    if mode == "burst32":
        burst(32)
    w = {"warm25": 25, "warm100": 100}.get(mode, 5)
    split = {"w2s3": 2, "w4s1": 4, "w1s4": 1, "s5": 0}.get(mode)
    for i in range(w):
        if split is not None and i == split:
            torch.cuda.synchronize()
        task.step(actions[i % 16])
    if mode == "warm5_sync_warm5":
        torch.cuda.synchronize()
        for i in range(5):
            task.step(actions[i % 16])
    if mode == "burst_late":
        burst(256)
    if mode == "nogc":
        gc.collect()
        gc.disable()
    if mode == "idle":
        torch.cuda.synchronize()
        time.sleep(0.2)
    out = []
    for _ in range(3 if mode == "twice" else 1):
        tot, launch, sync = region(task, actions, 20)
        out.append({"region_us": round(tot, 1), "launch_loop_us": round(launch, 1), "closing_sync_us": round(sync, 1)})
    print(json.dumps({"mode": mode, "warmup": w, "regions": out}), flush=True)


if __name__ == "__main__":
    main()
