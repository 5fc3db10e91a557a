"""Controller plug-ins (aerial_gym/control/controllers/*.py).

A controller object owns its per-env gains (SoA [12, N]: K_pos, K_vel, K_rot, K_angvel) and
selects the controller law evaluated inside the fused dynamics kernel.  Calling the object
(`controller(action)`) evaluates just the control law on the current state through
agx_controller_wrench and returns the wrench [N, 6], like the reference's `update()`.
"""
import torch

from .. import _lib
from ..tensors import aos_view, soa


class BaseController:
    KIND = "none"

    def __init__(self, control_config, num_envs, device, mode="robot"):
        self.cfg, self.num_envs, self.device, self.mode = control_config, num_envs, device, mode

    def init_tensors(self, global_tensor_dict):
        self.global_tensor_dict = global_tensor_dict

    def __call__(self, *args, **kwargs):
        return self.update(*args, **kwargs)

    def reset_commands(self):
        pass

    def reset(self):
        self.reset_idx(None)

    def reset_idx(self, env_ids):
        self.randomize_params(env_ids)

    def randomize_params(self, env_ids):
        pass


class NoControl(BaseController):  # no_control.py
    KIND = "none"

    def update(self, command_actions):
        return command_actions


class BaseLeeController(BaseController):  # base_lee_controller.py:23-118
    GAIN_ROWS = (("K_pos", 0), ("K_vel", 3), ("K_rot", 6), ("K_angvel", 9))

    def init_tensors(self, global_tensor_dict):
        super().init_tensors(global_tensor_dict)
        N, dev, cfg = self.num_envs, self.device, self.cfg
        self.gains_soa = soa(12, N, dev)
        self.gains_min = [0.0] * 12
        self.gains_max = [0.0] * 12
        for name, row in self.GAIN_ROWS:
            lo, hi = getattr(cfg, name + "_tensor_min"), getattr(cfg, name + "_tensor_max")
            self.gains_min[row:row + 3] = [float(x) for x in lo]
            self.gains_max[row:row + 3] = [float(x) for x in hi]
        mid = (torch.tensor(self.gains_max) + torch.tensor(self.gains_min)) / 2.0
        self.gains_soa[:] = mid.to(dev).unsqueeze(1)
        self.K_pos_tensor_current = aos_view(self.gains_soa, 0, 3)
        self.K_linvel_tensor_current = aos_view(self.gains_soa, 3, 6)
        self.K_rot_tensor_current = aos_view(self.gains_soa, 6, 9)
        self.K_angvel_tensor_current = aos_view(self.gains_soa, 9, 12)
        self.wrench_soa = soa(6, N, dev)
        self.wrench_command = aos_view(self.wrench_soa)
        global_tensor_dict["controller_gains_soa"] = self.gains_soa
        global_tensor_dict["controller_wrench_soa"] = self.wrench_soa
        self._env_binding = None  # (params struct, buffers struct) set by EnvManager

    def set_controller_gains(self, K_pos, K_vel, K_rot, K_angvel):
        if not getattr(self, "_per_env_gains_bound", True):
            raise RuntimeError(
                "gains are bound as constants (randomize_params is False): build the env with "
                "args={'per_env_gains': True} to set per-env gains at run time"
            )
        self.K_pos_tensor_current[:] = K_pos
        self.K_linvel_tensor_current[:] = K_vel
        self.K_rot_tensor_current[:] = K_rot
        self.K_angvel_tensor_current[:] = K_angvel

    def randomize_params(self, env_ids):
        """base_lee_controller.py:101-118; the draw happens inside the env reset path."""
        if not self.cfg.randomize_params or self._env_binding is None:
            return
        self._env_binding.randomize_controller_gains(env_ids)

    def update(self, command_actions):
        if self._env_binding is None:
            raise RuntimeError("controller is not bound to an EnvManager yet")
        return self._env_binding.controller_wrench(command_actions)


class LeePositionController(BaseLeeController):
    KIND = "position"


class LeeVelocityController(BaseLeeController):
    KIND = "velocity"


class LeeAttitudeController(BaseLeeController):
    KIND = "attitude"


class LeeRatesController(BaseLeeController):
    KIND = "rates"


class LeeAccelerationController(BaseLeeController):
    KIND = "acceleration"


class LeeVelocitySteeringAngleController(BaseLeeController):
    KIND = "velocity_steering"


class FullyActuatedController(BaseLeeController):
    KIND = "fully_actuated"
