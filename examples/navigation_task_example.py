"""Depth-camera navigation task with random actions (reference: examples/navigation_task_example.py)."""
import os
import time

import torch

import aerial_gym_simulator_amd  # noqa: F401
from aerial_gym_simulator_amd.registry.task_registry import task_registry

if __name__ == "__main__":
    task = task_registry.make_task("navigation_task", seed=42, num_envs=1024, headless=True, use_warp=True)
    task.reset()
    n = task.num_envs
    t0 = time.time()
    for i in range(int(os.environ.get("AGX_EXAMPLE_STEPS", 1000))):
        actions = torch.rand((n, 4), device="cuda:0") * 2 - 1
        obs, reward, terminated, truncated, info = task.step(actions)
        if i % 200 == 199:
            depth = task.obs_dict["depth_range_pixels"]
            print(f"step {i + 1}: reward {float(reward.mean()):+.2f}, crashes {int(terminated.sum())}, timeouts {int(truncated.sum())}, "
                  f"curriculum level {task.curriculum_level}, depth image {tuple(depth.shape)} min {float(depth.min()):.2f}")
    torch.cuda.synchronize()
    print(f"{n * 1000 / (time.time() - t0):,.0f} env-steps/s")
