"""Lee position control of 64 quadrotors towards random set-points (reference: examples/position_control_example.py)."""
import os
import torch

import aerial_gym_simulator_amd  # noqa: F401
from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

if __name__ == "__main__":
    env = SimBuilder().build_env(sim_name="base_sim", env_name="empty_env", robot_name="base_quadrotor", controller_name="lee_position_control",
                                 args=None, device="cuda:0", num_envs=64, headless=True, use_warp=False)
    actions = torch.zeros((env.num_envs, 4), device="cuda:0")
    env.reset()
    g = env.get_obs()
    for i in range(int(os.environ.get("AGX_EXAMPLE_STEPS", 3000))):
        if i % 1000 == 0:
            actions[:, 0:3] = 0.6 * (torch.rand_like(actions[:, 0:3]) * 2 - 1)   # stay inside the +-1 m env
            actions[:, 3] = torch.pi * (torch.rand_like(actions[:, 3]) * 2 - 1)
            env.reset()
        env.step(actions=actions)
        if i % 250 == 249:
            err = (g["robot_position"] - actions[:, 0:3]).norm(dim=1)
            print(f"step {i + 1}: mean position error {float(err.mean()):.3f} m, max {float(err.max()):.3f} m")
