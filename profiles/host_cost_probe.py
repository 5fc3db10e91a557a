"""Host cost of one position-task step (8192 envs), piece by piece.  python profiles/host_cost_probe.py (GPU box)"""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aerial_gym_simulator_amd import _lib  # noqa: E402


def per_call(fn, n=20000):
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return round((time.perf_counter() - t0) / n * 1e6, 3)


def main():
    device = "cuda:0"
    task = bench.make_task("dynamics", 8192, device, False)
    task.reset()
    g = torch.Generator(device=device).manual_seed(1234)
    actions = [torch.rand(8192, 4, device=device, generator=g) * 2 - 1 for _ in range(16)]
    for i in range(50):
        task.step(actions[i % 16])
    torch.cuda.synchronize()
    env = task.sim_env
    a = actions[0]
    out = {}
    out["current_stream_us"] = per_call(lambda: _lib.current_stream(device))
    out["data_ptr_us"] = per_call(lambda: a.data_ptr())
    out["fast_path_conditions_us"] = per_call(lambda: (a.dtype is torch.float32 and a.is_contiguous() and a.is_cuda and a.shape == task._action_shape))
    lib = env._lib
    ver = lib.agx_abi_version
    out["ctypes_noarg_call_us"] = per_call(lambda: ver())
    real = task._plan_fn
    task._plan_fn = lambda plan, ptr, stream: 0
    out["task_step_python_only_us"] = per_call(lambda: task.step(a))
    task._plan_fn = real
    stream = env._stream()
    ptr = a.data_ptr()
    plan = task._plan
    # C side (2 launches) from an idle queue, bursts of 8 / 20: no throttling possible
    for burst in (8, 20, 40):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            real(plan, ptr, stream)
            t1 = time.perf_counter()
            for _ in range(burst - 1):
                real(plan, ptr, stream)
            t2 = time.perf_counter()
            ts.append(((t1 - t0) * 1e6, (t2 - t1) / (burst - 1) * 1e6))
        ts.sort(key=lambda x: x[1])
        out[f"plan_fn_burst{burst}"] = {"first_call_us": round(ts[2][0], 2), "next_calls_us_each": round(ts[2][1], 2)}
    # the same through task.step
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        task.step(a)
        t1 = time.perf_counter()
        for _ in range(19):
            task.step(a)
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e6, (t2 - t1) / 19 * 1e6))
    ts.sort(key=lambda x: x[1])
    out["task_step_burst20"] = {"first_call_us": round(ts[2][0], 2), "next_calls_us_each": round(ts[2][1], 2)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
