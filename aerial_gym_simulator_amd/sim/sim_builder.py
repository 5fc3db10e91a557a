"""aerial_gym/sim/sim_builder.py:22-48"""
import torch

from ..env_manager.env_manager import EnvManager


class SimBuilder:
    def __init__(self):
        self.sim_name = self.env_name = self.robot_name = self.env = None

    def delete_env(self):
        del self.env
        self.env = None
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
            torch.cuda.synchronize()

    def build_env(self, sim_name, env_name, robot_name, controller_name, device, args=None, num_envs=None,
                  use_warp=None, headless=None):
        self.sim_name, self.env_name, self.robot_name = sim_name, env_name, robot_name
        self.env = EnvManager(sim_name=sim_name, env_name=env_name, robot_name=robot_name,
                              controller_name=controller_name, args=args, device=device, num_envs=num_envs,
                              use_warp=use_warp, headless=headless)
        return self.env
