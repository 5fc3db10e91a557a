"""Where does a 8192-env task.step() spend its time?  (host enqueue vs GPU execution)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = 8192
dev = "cuda:0"
task = bench.make_task("dynamics", n, dev, False); task.reset()
acts = [torch.rand(n, 4, device=dev) * 2 - 1 for _ in range(16)]
def run(name, fn, steps=3000):
    for i in range(200): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): fn(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:40s} enqueue {1e6*(t1-t0)/steps:6.2f} us/step   total {1e6*(t2-t0)/steps:6.2f} us/step")
run("task.step (bench loop)", lambda i: task.step(acts[i % 16]))
a0 = acts[0]
run("task.step (same tensor)", lambda i: task.step(a0))
plan, fn, env = task._plan, task._plan_fn, task.sim_env
st = env._stream(); ptr = a0.data_ptr()
run("plan_fn only (cached stream+ptr)", lambda i: fn(plan, ptr, st))
run("torch.cuda.current_stream only", lambda i: env._stream())
lib = env._lib
run("agx_update_states only", lambda i: lib.agx_update_states(env._buffers, n, st))
run("2x agx_update_states", lambda i: (lib.agx_update_states(env._buffers, n, st), lib.agx_update_states(env._buffers, n, st)))

def host_cost(name, fn, reps=200):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(reps): fn(i)
        t1 = time.perf_counter(); torch.cuda.synchronize()
        best = min(best, 1e6 * (t1 - t0) / reps)
    print(f"{name:40s} host-only {best:6.2f} us/call (queue never full)")
a_ptr = a0.data_ptr()
host_cost("agx_env_step", lambda i: lib.agx_env_step(env._params, env._buffers, n, a_ptr, 1, env.task_args, st))
host_cost("agx_post_step_position", lambda i: lib.agx_post_step_position(env._params, env._buffers, n, env._reset_args, plan.target, plan.obs, st))
host_cost("plan_fn (both)", lambda i: fn(plan, ptr, st))
host_cost("agx_update_states", lambda i: lib.agx_update_states(env._buffers, n, st))
host_cost("agx_obs_position", lambda i: lib.agx_obs_position(env._buffers, n, plan.target, plan.obs, st))

B, P, R = env._buffers, env._params, env._reset_args
def pair(i):
    lib.agx_env_step(P, B, n, a_ptr, 1, env.task_args, st); lib.agx_post_step_position(P, B, n, R, plan.target, plan.obs, st)
run("pair via two ctypes calls (no parity)", pair)
def pair_obs(i):
    lib.agx_env_step(P, B, n, a_ptr, 1, env.task_args, st); lib.agx_obs_position(B, n, plan.target, plan.obs, st)
run("env_step + obs_position", pair_obs)
run("env_step only", lambda i: lib.agx_env_step(P, B, n, a_ptr, 1, env.task_args, st))
run("env_step only, no task", lambda i: lib.agx_env_step(P, B, n, a_ptr, 1, None, st))
for reps in (300, 3000):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s0.record()
    for i in range(reps): pair(i)
    s1.record(); torch.cuda.synchronize()
    print("events, no blocker, reps", reps, s0.elapsed_time(s1) / reps * 1e3, "us/pair")

def pair_toggle(i):
    B.flag_parity ^= 1
    lib.agx_env_step(P, B, n, a_ptr, 1, env.task_args, st); lib.agx_post_step_position(P, B, n, R, plan.target, plan.obs, st)
run("pair + python parity toggle", pair_toggle)
run("plan_fn again", lambda i: fn(plan, ptr, st))
g = env.global_tensor_dict
print("flags", g["reset_flag"].tolist(), "mask", int(g["reset_mask"].sum()), "max steps", int(g["sim_steps"].max()), "episode_len", task.task_config.episode_len_steps, env.task_args.episode_len)
run("task.step (bench loop) AGAIN", lambda i: task.step(acts[i % 16]))
task.reset()
run("task.step after reset", lambda i: task.step(acts[i % 16]))
run("task.step after reset 2", lambda i: task.step(acts[i % 16]))
