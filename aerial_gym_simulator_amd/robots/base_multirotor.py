"""Host mirror of aerial_gym/robots/base_multirotor.py.

Owns the robot-side tensors (derived state, motor model, gains) and the constant block
(`AgxRobotParams`).  Everything `BaseMultirotor.step` does in the reference
(update_states, clip, controller, allocation, motor model, drag, disturbance) is fused
with the rigid-body integration in agx_dynamics_substeps, launched by EnvManager.
"""
import torch

from ..control.control_allocation import ControlAllocator
from ..tensors import aos_view, soa
from ..utils.logging import CustomLogger
from .base_robot import BaseRobot
from .robot_model import pack_robot_params, robot_params_dict

logger = CustomLogger("base_multirotor")


class BaseMultirotor(BaseRobot):
    def __init__(self, robot_config, controller_name, env_config, device):
        super().__init__(robot_config, controller_name, env_config, device)
        self.force_application_level = self.cfg.control_allocator_config.force_application_level
        self.output_mode = "forces" if controller_name == "no_control" else "wrench"
        if self.force_application_level == "root_link" and controller_name == "no_control":
            raise ValueError("Force application level 'root_link' cannot be used with 'no_control'.")
        self.control_allocator = None
        self._env_binding = None

    def init_tensors(self, global_tensor_dict):
        super().init_tensors(global_tensor_dict)
        g, N, dev = global_tensor_dict, self.num_envs, self.device
        self.derived_soa = soa(16, N, dev)
        self.robot_euler_angles = aos_view(self.derived_soa, 0, 3)
        self.robot_vehicle_orientation = aos_view(self.derived_soa, 3, 7)
        self.robot_vehicle_linvel = aos_view(self.derived_soa, 7, 10)
        self.robot_body_linvel = aos_view(self.derived_soa, 10, 13)
        self.robot_body_angvel = aos_view(self.derived_soa, 13, 16)
        g["robot_derived_soa"] = self.derived_soa
        g["robot_euler_angles"] = self.robot_euler_angles
        g["robot_vehicle_orientation"] = self.robot_vehicle_orientation
        g["robot_vehicle_linvel"] = self.robot_vehicle_linvel
        g["robot_body_linvel"] = self.robot_body_linvel
        g["robot_body_angvel"] = self.robot_body_angvel
        g["num_robot_actions"] = self.controller_config.num_actions
        # what the reference's controllers read in init_tensors (controllers/base_controller.py:11-24)
        from .robot_model import composite_body

        mass, _, inertia = composite_body(self.cfg.robot_model)
        g.setdefault("robot_mass", torch.full((N,), float(mass), dtype=torch.float32, device=dev))
        g.setdefault("robot_inertia", torch.tensor(inertia, dtype=torch.float32, device=dev).view(1, 3, 3).expand(N, 3, 3))
        self.controller.init_tensors(g)
        self.min_init_state = [float(x) for x in self.cfg.init_config.min_init_state]
        self.max_init_state = [float(x) for x in self.cfg.init_config.max_init_state]
        self.max_force_and_torque_disturbance = [float(x) for x in self.cfg.disturbance.max_force_and_torque_disturbance]
        self.control_allocator = ControlAllocator(
            num_envs=N, dt=self.dt, config=self.cfg.control_allocator_config, device=dev,
            random_source=g["random_source"],
        )
        kind = getattr(self.controller, "KIND", None)
        # A class registered through controller_registry.register_controller that is not one of the built-in laws (no KIND):
        # the reference's plug-in contract -- __init__(config, num_envs, device), init_tensors(global_tensor_dict),
        # __call__(action) -> wrench [N, 6] (base_lee_controller.py:23-118).  It is evaluated by the host, in torch, once
        # per physics sub-step on freshly updated state tensors; the kernel takes its output (AGX_CTRL_WRENCH).
        self.external_controller = kind is None
        if self.external_controller:
            if self.output_mode != "wrench":
                raise ValueError("an external controller must return a body wrench [N, 6]")
            kind = "wrench"
        self.params_dict = robot_params_dict(self.cfg, self.controller_config, kind, g["sim_config"])
        self.params = pack_robot_params(self.params_dict)
        # A robot CLASS (robot_registry.register) that overrides step(): the reference's plug-in contract (base_robot.py:10-63).
        # It is called by the host once per physics sub-step; what it leaves in the per-body tensors is integrated.
        self.external_robot = type(self).step is not BaseMultirotor.step

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))

    def reset_idx(self, env_ids):
        if len(env_ids) == 0:
            return
        if self._env_binding is None:
            raise RuntimeError("robot is not bound to an EnvManager yet")
        self._env_binding.reset_robots(env_ids)

    def update_states(self):
        self._env_binding.update_states()

    def step(self, action_tensor):
        """base_multirotor.py:296-307 as ONE launch (agx_robot_step): update_states, clip, controller, allocation + motor model
        (the motor thrusts advance), `robot_force_tensors` / `robot_torque_tensors` written, drag and disturbance added to the
        root body.  EnvManager.step does NOT go through here for a robot whose class leaves step() alone (controller ..
        integration are one fused launch then); a registered subclass that overrides step() is called once per physics
        sub-step like the reference's (robot_manager.py:486-489) and typically calls this through super().step(action)."""
        if self._env_binding is None:
            raise RuntimeError("robot is not bound to an EnvManager yet")
        if action_tensor.shape[0] != self.num_envs:
            raise ValueError("Action tensor does not have the correct number of environments")
        self._env_binding.robot_step(action_tensor)
