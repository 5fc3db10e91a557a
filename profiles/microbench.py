"""Per-kernel launch durations at 8192 envs (HIP events, queue pre-filled behind a blocker)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from aerial_gym_simulator_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda:0"
task = bench.make_task("dynamics", n, dev, False)
task.reset()
env = task.sim_env
a = torch.rand(n, 4, device=dev) * 2 - 1
for _ in range(10):
    task.step(a)
lib, P, B, R = env._lib, env._params, env._buffers, env._reset_args
tgt, obs, rew = _lib.dptr(task.target_soa), _lib.dptr(task.task_obs["observations"]), _lib.dptr(task.rewards)
blocker = torch.randn(4096, 4096, device=dev)


def timeit(name, fn, reps=300):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        for _ in range(6):
            blocker @ blocker
        s0.record()
        st = env._stream()
        for _ in range(reps):
            fn(st)
        s1.record()
        torch.cuda.synchronize()
        best = min(best, s0.elapsed_time(s1) / reps * 1e3)
    print(f"{name:34s} {best:8.2f} us/launch")


env.global_tensor_dict["reset_flag"].zero_()
timeit("agx_obs_position", lambda st: lib.agx_obs_position(B, n, tgt, obs, st))
timeit("agx_reset_masked (flag 0)", lambda st: lib.agx_reset_masked(P, B, n, R, st))
timeit("agx_post_step_position (flag 0)", lambda st: lib.agx_post_step_position(P, B, n, R, tgt, obs, st))
timeit("agx_update_states", lambda st: lib.agx_update_states(B, n, st))
timeit("agx_reward_position", lambda st: lib.agx_reward_position(B, n, tgt, 100000, 1, rew, st))
timeit("agx_env_step k=1 (fused reward)", lambda st: lib.agx_env_step(P, B, n, _lib.dptr(a), 1, env.task_args, st))
timeit("agx_env_step k=1 (no task)", lambda st: lib.agx_env_step(P, B, n, _lib.dptr(a), 1, None, st))
ap = _lib.dptr(a)
timeit("PAIR env_step + post_step (flag 0)", lambda st: (lib.agx_env_step(P, B, n, ap, 1, env.task_args, st), lib.agx_post_step_position(P, B, n, R, tgt, obs, st)))
timeit("PAIR env_step + obs_position", lambda st: (lib.agx_env_step(P, B, n, ap, 1, env.task_args, st), lib.agx_obs_position(B, n, tgt, obs, st)))
timeit("PAIR env_step + update_states", lambda st: (lib.agx_env_step(P, B, n, ap, 1, env.task_args, st), lib.agx_update_states(B, n, st)))
timeit("PAIR update_states + obs_position", lambda st: (lib.agx_update_states(B, n, st), lib.agx_obs_position(B, n, tgt, obs, st)))
env.global_tensor_dict["reset_flag"].fill_(1)
env.global_tensor_dict["reset_mask"].fill_(1)
timeit("agx_post_step_position (all reset)", lambda st: (env.global_tensor_dict["reset_flag"].fill_(1), lib.agx_post_step_position(P, B, n, R, tgt, obs, st)))
