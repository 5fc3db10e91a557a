"""hipGraph replay of the position-task step (two steps per graph: one per reset-flag parity) against
eager launches.  python profiles/graph_probe.py   (on the GPU box)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
task = bench.make_task("dynamics", N, dev, False)
task.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = [torch.rand(N, 4, device=dev, generator=g) * 2 - 1 for _ in range(16)]
static = torch.zeros(N, 4, device=dev)


def eager(steps):
    for i in range(steps):
        task.step(acts[i & 15])


def timed(fn, steps, label):
    fn(200)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{label:34s} host {1e6 * host / steps:6.2f} us/step   total {1e6 * tot / steps:6.2f} us/step", flush=True)


timed(eager, 4000, "eager, 2 launches per step")
for per_graph in (2, 8, 32):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        task.step(static)
        task.step(static)  # back at the parity the capture starts from
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(per_graph):
            task.step(static)

    def replay(steps, copy):
        for i in range(steps // per_graph):
            if copy:
                static.copy_(acts[i & 15])
            graph.replay()

    timed(lambda s: replay(s, False), 4000, f"graph of {per_graph} steps, fixed actions")
    if per_graph == 2:
        timed(lambda s: replay(s, True), 4000, f"graph of {per_graph} steps + action copy")
    del graph
