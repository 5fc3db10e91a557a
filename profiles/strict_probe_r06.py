"""Round 6: where the strict-RNG step's 62 us go, and what the candidate replacements cost (one MI355X).
    python profiles/strict_probe_r06.py  -> JSON lines
 * the reset-flag read as rounds 1-5 did it (`int(flag[parity].item())`);
 * the seven `uniform_` calls of the strict reset (env_manager._draw_reset_randoms) eager, and captured once in a torch CUDAGraph
   and replayed (the generator's offset advances by the same amount, the numbers are the same: checked here);
 * a pinned host word written by a kernel and polled by the host (torch only: `copy_(non_blocking)` into pinned memory + poll)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dev = "cuda:0"
N = 8192
out = {}


def timeit(fn, reps=300, sync_each=False):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        if sync_each:
            torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    return {"host_us": host * 1e6, "total_us": (time.perf_counter() - t0) / reps * 1e6}


flag = torch.zeros(2, dtype=torch.int32, device=dev)
out["flag_item"] = timeit(lambda: int(flag[1].item()))
bufs = [torch.zeros(N, c, device=dev) for c in (3, 3, 13, 4, 4, 4, 4)]


def draws():
    for b in bufs:
        b.uniform_(0.0, 1.0)


out["draws_eager"] = timeit(draws)
# graph capture of the same seven calls
torch.manual_seed(123)
draws()
ref1 = [b.clone() for b in bufs]
draws()
ref2 = [b.clone() for b in bufs]
state_after_eager = torch.cuda.get_rng_state(dev).clone()
torch.manual_seed(123)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        draws()
torch.cuda.synchronize()
torch.manual_seed(123)
g.replay()
torch.cuda.synchronize()
same1 = all(torch.equal(a, b) for a, b in zip(ref1, bufs))
g.replay()
torch.cuda.synchronize()
same2 = all(torch.equal(a, b) for a, b in zip(ref2, bufs))
out["graph_replay_equals_eager"] = {"first": same1, "second": same2,
                                    "generator_state_equal": bool(torch.equal(state_after_eager, torch.cuda.get_rng_state(dev)))}
out["draws_graph"] = timeit(g.replay)
# pinned word + poll
pin = torch.zeros(2, dtype=torch.int32).pin_memory()
seq = torch.zeros(2, dtype=torch.int32, device=dev)
pin_np = pin.numpy()
counter = [0]


def publish_and_poll():
    counter[0] += 1
    seq.fill_(counter[0])
    pin.copy_(seq, non_blocking=True)
    while pin_np[0] != counter[0]:
        pass


out["pinned_copy_poll"] = timeit(publish_and_poll)
print(json.dumps(out))

# the strict task step as it is
import bench  # noqa: E402

for desync in (False, True):
    ts = bench.make_task("dynamics", N, dev, True)
    ts.reset()
    if desync:
        bench.desynchronise_episodes(ts)
    gen = torch.Generator(device=dev).manual_seed(1)
    acts = [torch.rand(N, 4, device=dev, generator=gen) * 2 - 1 for _ in range(8)]
    dt = bench.timed_steps(ts, acts, 300, 30, 1)
    print(json.dumps({"strict_step_us": dt / 300 * 1e6, "desynchronised": desync, "host_us": bench.timed_steps.last_host_s / 300 * 1e6}))
    del ts
