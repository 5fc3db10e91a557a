"""The task interface an RL library drives (reference: examples/rl_env_example.py)."""
import os
import time

import torch

import aerial_gym_simulator_amd  # noqa: F401
from aerial_gym_simulator_amd.registry.task_registry import task_registry

if __name__ == "__main__":
    task = task_registry.make_task("position_setpoint_task", seed=0, num_envs=4096, headless=True)
    task.reset()
    actions = torch.zeros((task.num_envs, task.task_config.action_space_dim), device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(int(os.environ.get("AGX_EXAMPLE_STEPS", 5000))):
        obs, reward, terminated, truncated, info = task.step(actions)
        if i % 1000 == 999:
            print(f"step {i + 1}: mean reward {float(reward.mean()):.3f}, resets so far "
                  f"{int(task.obs_dict['episode_count'].sum())}, obs {tuple(obs['observations'].shape)}")
    torch.cuda.synchronize()
    print(f"{task.num_envs * 5000 / (time.time() - t0):,.0f} env-steps/s")
