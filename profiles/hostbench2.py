import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = 8192; dev = "cuda:0"
task = bench.make_task("dynamics", n, dev, False); task.reset()
acts = [torch.rand(n, 4, device=dev) * 2 - 1 for _ in range(16)]
def run(name, fn, steps=3000):
    for i in range(100): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): fn(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:40s} enqueue {1e6*(t1-t0)/steps:6.2f} us/step   total {1e6*(t2-t0)/steps:6.2f} us/step", flush=True)
plan, fn, env = task._plan, task._plan_fn, task.sim_env
st = env._stream(); a0 = acts[0]; ptr = a0.data_ptr()
for rep in range(3):
    run("plan_fn", lambda i: fn(plan, ptr, st))
    run("task.step", lambda i: task.step(acts[i % 16]))
real = task._plan_fn
task._plan_fn = lambda *a: 0
run("task.step with no-op launch (pure python)", lambda i: task.step(acts[i % 16]))
task._plan_fn = real
