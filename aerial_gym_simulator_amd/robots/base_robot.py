"""aerial_gym/robots/base_robot.py:10-63 (host mirror)."""
from abc import ABC, abstractmethod

from ..registry.controller_registry import controller_registry


class BaseRobot(ABC):
    def __init__(self, robot_config, controller_name, env_config, device):
        self.cfg = robot_config
        self.num_envs = env_config.env.num_envs
        self.device = device
        self.controller_name = controller_name
        self.controller, self.controller_config = controller_registry.make_controller(
            controller_name, self.num_envs, self.device
        )
        if controller_name == "no_control":
            self.controller_config.num_actions = self.cfg.control_allocator_config.num_motors
        self.num_actions = self.controller_config.num_actions

    @abstractmethod
    def init_tensors(self, global_tensor_dict):
        g = global_tensor_dict
        self.dt, self.gravity = g["dt"], g["gravity"]
        self.robot_state = g["robot_state_tensor"]
        self.robot_position, self.robot_orientation = g["robot_position"], g["robot_orientation"]
        self.robot_linvel, self.robot_angvel = g["robot_linvel"], g["robot_angvel"]
        # tensors for robot forces and torques (base_robot.py:46-48): [N, num_bodies, 3], each body's wrench in ITS frame
        self.robot_force_tensors, self.robot_torque_tensors = g["robot_force_tensor"], g["robot_torque_tensor"]
        self.env_bounds_min, self.env_bounds_max = g["env_bounds_min"], g["env_bounds_max"]

    @abstractmethod
    def reset(self):
        ...

    @abstractmethod
    def reset_idx(self, env_ids):
        ...

    @abstractmethod
    def step(self):
        ...
