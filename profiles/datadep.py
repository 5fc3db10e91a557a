"""Is k_env_step's duration data dependent?  hover state + constant action vs fresh random state + random action."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from aerial_gym_simulator_amd import _lib
n = 8192; dev = "cuda:0"
task = bench.make_task("dynamics", n, dev, False); task.reset()
env = task.sim_env; lib, P, B = env._lib, env._params, env._buffers
g = env.global_tensor_dict
blocker = torch.randn(4096, 4096, device=dev)
def ktime(a, reps=200, restore=None):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        if restore is not None:
            g["robot_state_soa"].copy_(restore[0]); env.robot_manager.robot.control_allocator.motor_model.thrust_soa.copy_(restore[1])
        torch.cuda.synchronize()
        for _ in range(6): blocker @ blocker
        s0.record(); st = env._stream(); p = _lib.dptr(a)
        for _ in range(reps): lib.agx_env_step(P, B, n, p, 1, env.task_args, st)
        s1.record(); torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(s1) / reps * 1e3)
    return best
a = torch.rand(n, 4, device=dev) * 2 - 1
snap = (g["robot_state_soa"].clone(), env.robot_manager.robot.control_allocator.motor_model.thrust_soa.clone())
print("fresh random state, 20 steps   :", ktime(a, reps=20, restore=snap))
print("fresh random state, 200 steps  :", ktime(a, reps=200, restore=snap))
for _ in range(3000): lib.agx_env_step(P, B, n, _lib.dptr(a), 1, None, env._stream())
torch.cuda.synchronize()
print("settled hover, 200 steps       :", ktime(a, reps=200))
a2 = torch.rand(n, 4, device=dev) * 2 - 1
print("new setpoint from hover, 30    :", ktime(a2, reps=30))
st = g["robot_state_tensor"]
print("state stats: |v| max", float(g["robot_linvel"].abs().max()), "|w| max", float(g["robot_angvel"].abs().max()), "thrust min/max",
      float(env.robot_manager.robot.control_allocator.motor_model.thrust_soa.min()), float(env.robot_manager.robot.control_allocator.motor_model.thrust_soa.max()))
