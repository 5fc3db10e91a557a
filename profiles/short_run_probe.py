"""Where do the ~0.09 ms go that a 20-step timed region (the driver's `bench.py --steps 20 --warmup 5`) shows on top of
20 x 12.4 us?  Times the same region (barrier-less world of one: synchronize, K steps, synchronize) under different
histories of the device, and splits it into the host's launch loop and the closing synchronize.

    python profiles/short_run_probe.py > gpurun_out/short_run_probe.txt
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def region(task, actions, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        task.step(actions[i % len(actions)])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6


def main():
    device = "cuda:0"
    task = bench.make_task("dynamics", 8192, device, False)
    task.reset()
    g = torch.Generator(device=device).manual_seed(1234)
    actions = [torch.rand(8192, 4, device=device, generator=g) * 2 - 1 for _ in range(16)]
    spin = torch.empty(1 << 26, device=device)

    def report(label, steps=20, reps=1):
        for _ in range(reps):
            tot, launch, sync = region(task, actions, steps)
            print(json.dumps({"history": label, "steps": steps, "region_us": round(tot, 1), "launch_loop_us": round(launch, 1),
                              "closing_sync_us": round(sync, 1), "us_per_step": round(tot / steps, 2),
                              "env_steps_per_s": round(8192 * steps / tot * 1e6)}), flush=True)

    for i in range(5):
        task.step(actions[i])
    report("5 warm-up steps after task creation (the driver's run)")
    report("immediately again", reps=3)
    for i in range(2000):
        task.step(actions[i % 16])
    report("right after 2000 steps", reps=3)
    torch.cuda.synchronize()
    time.sleep(0.5)
    report("after 0.5 s of idle device", reps=2)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        spin.add_(1.0)
    report("after 0.3 s of a streaming kernel loop (not synchronized before the region's own synchronize)", reps=2)
    for k in (1, 5, 20, 100, 500, 2000):
        report("steady state, region length sweep", steps=k)
    # the launch loop alone, queue kept shallow: host cost per step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        task.step(actions[i % 16])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(json.dumps({"host_us_per_step_call": round((t1 - t0) / 200 * 1e6, 2)}), flush=True)
    # an empty region: synchronize -> synchronize
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.cuda.synchronize()
        print(json.dumps({"empty_region_us": round((time.perf_counter() - t0) * 1e6, 2)}), flush=True)
    # one step, then synchronize: launch-to-completion latency of a single step from an idle queue
    for _ in range(3):
        tot, launch, sync = region(task, actions, 1)
        print(json.dumps({"single_step_from_idle_us": round(tot, 1), "launch": round(launch, 1), "sync": round(sync, 1)}), flush=True)


if __name__ == "__main__":
    main()
