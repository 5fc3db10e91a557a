import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from aerial_gym_simulator_amd import _lib
n = 8192; dev = "cuda:0"
task = bench.make_task("dynamics", n, dev, False); task.reset()
env = task.sim_env; lib, P, B, R = env._lib, env._params, env._buffers, env._reset_args
g = env.global_tensor_dict
acts = [torch.rand(n, 4, device=dev) * 2 - 1 for _ in range(16)]
blocker = torch.randn(4096, 4096, device=dev)
plan = task._plan
def kt(fn, reps=100):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for _ in range(6): blocker @ blocker
    s0.record(); st = env._stream()
    for _ in range(reps): fn(st)
    s1.record(); torch.cuda.synchronize()
    return s0.elapsed_time(s1) / reps * 1e3
def stats(tag):
    s = g["robot_state_tensor"]
    print(tag, "finite", bool(torch.isfinite(s).all()), "|p|max %.2f |v|max %.2f |w|max %.2f" % (float(g["robot_position"].abs().max()), float(g["robot_linvel"].abs().max()), float(g["robot_angvel"].abs().max())),
          "euler max %.3f" % float(g["robot_euler_angles"].abs().max()), "steps max", int(g["sim_steps"].max()), "flags", g["reset_flag"].tolist())
a0 = acts[0]; p0 = _lib.dptr(a0)
stats("after reset")
print(" env_step %.2f  post %.2f" % (kt(lambda st: lib.agx_env_step(P, B, n, p0, 1, env.task_args, st)), kt(lambda st: lib.agx_post_step_position(P, B, n, R, plan.target, plan.obs, st))))
for i in range(300): task.step(acts[i % 16])
stats("after 300 varying-action steps")
print(" env_step %.2f  post %.2f" % (kt(lambda st: lib.agx_env_step(P, B, n, p0, 1, None, st)), kt(lambda st: lib.agx_post_step_position(P, B, n, R, plan.target, plan.obs, st))))
for i in range(300): task.step(acts[i % 16])
stats("after 600")
print(" env_step(no task) %.2f env_step(task) %.2f post %.2f" % (kt(lambda st: lib.agx_env_step(P, B, n, p0, 1, None, st)), kt(lambda st: lib.agx_env_step(P, B, n, p0, 1, env.task_args, st)), kt(lambda st: lib.agx_post_step_position(P, B, n, R, plan.target, plan.obs, st))))
stats("end")
