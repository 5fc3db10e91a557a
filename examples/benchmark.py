"""FPS of the raw EnvManager loop (reference: aerial_gym/examples/benchmark.py:17-100)."""
import argparse
import time

import torch

import aerial_gym_simulator_amd  # noqa: F401
from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rendering", action="store_true", help="env_with_obstacles + depth camera instead of empty_env")
    ap.add_argument("--num-envs", type=int, default=None)
    ap.add_argument("--steps", type=int, default=2000)
    args = ap.parse_args()
    torch.manual_seed(0)
    if args.rendering:
        env = SimBuilder().build_env(sim_name="base_sim", env_name="env_with_obstacles", robot_name="base_quadrotor_with_camera",
                                     controller_name="lee_velocity_control", args=None, device="cuda:0", num_envs=args.num_envs or 16,
                                     headless=True, use_warp=True)
    else:
        env = SimBuilder().build_env(sim_name="base_sim", env_name="empty_env", robot_name="base_quadrotor", controller_name="no_control",
                                     args=None, device="cuda:0", num_envs=args.num_envs or 256, headless=True, use_warp=True)
    n = env.num_envs
    env.reset()
    actions = 0.295 * torch.ones((n, env.num_robot_actions), device="cuda:0")  # hover thrust per motor for no_control
    if args.rendering:
        actions = torch.zeros((n, 4), device="cuda:0")
    for _ in range(100):
        env.step(actions=actions)
        if args.rendering:
            env.render(render_components="sensors")
        env.reset_terminated_and_truncated_envs()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        env.step(actions=actions)
        if args.rendering:
            env.render(render_components="sensors")
        env.reset_terminated_and_truncated_envs()
    torch.cuda.synchronize()
    dt = time.time() - t0
    k = env.cfg.env.num_physics_steps_per_env_step_mean
    print(f"{n} envs, {args.steps} steps in {dt:.3f} s: {n * args.steps / dt:,.0f} env-steps/s, "
          f"{n * args.steps * k / dt:,.0f} physics steps/s, real-time factor {n * args.steps * k * env.global_tensor_dict['dt'] / dt:,.0f}")
