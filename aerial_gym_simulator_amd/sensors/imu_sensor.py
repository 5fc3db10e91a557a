"""IMUSensor of the reference (aerial_gym/sensors/imu_sensor.py) on agx_imu_update / agx_imu_reset.

The reference reads Isaac Gym's force sensor on the base link every physics sub-step; here the env-step
kernel leaves the net applied force of its last sub-step in `body_force` and one launch per env step
advances the bias random walk by the k sub-steps and produces the measurement of the last one."""
import math

import torch

from .. import _lib


class IMUSensor:
    def __init__(self, sensor_config, num_envs, device):
        self.cfg, self.num_envs, self.device = sensor_config, num_envs, device
        self.world_frame = sensor_config.world_frame
        self.gravity_compensation = sensor_config.gravity_compensation

    def init_tensors(self, global_tensor_dict):
        g, N, dev, cfg = global_tensor_dict, self.num_envs, self.device, self.cfg
        self.g = g
        self.dt = g["dt"]
        self.sqrt_dt = math.sqrt(self.dt)
        self.bias = torch.zeros(N, 6, device=dev)
        self.imu_meas = torch.zeros(N, 6, device=dev)
        self.sensor_quats = torch.zeros(N, 4, device=dev)
        self.sensor_quats[:, 3] = 1.0
        self.body_force = torch.zeros(3, N, device=dev)  # [3][N], written by agx_env_step
        g["imu_measurement"] = self.imu_meas
        g["imu_body_force_soa"] = self.body_force
        A = _lib.AgxImuArgs()
        for i in range(6):
            A.bias_std[i], A.noise_std[i] = cfg.bias_std[i], cfg.imu_noise_std[i]
            A.max_value[i], A.max_bias_init[i] = cfg.max_measurement_value[i], cfg.max_bias_init_value[i]
        grav = g["sim_config"].sim.gravity
        for i in range(3):
            A.min_rot[i], A.max_rot[i] = math.radians(cfg.min_euler_rotation_deg[i]), math.radians(cfg.max_euler_rotation_deg[i])
            A.g_world[i] = grav[i] * (1 - int(self.gravity_compensation))
        A.sqrt_dt = self.sqrt_dt
        A.mass = float(g["robot_mass"][0])
        A.world_frame, A.enable_noise, A.enable_bias = int(cfg.world_frame), int(cfg.enable_noise), int(cfg.enable_bias)
        self._args = A
        self._u_bias = torch.zeros(N, 6, device=dev)
        self._u_rot = torch.zeros(N, 3, device=dev)
        self._z = None

    # strict_rng: rand_like(bias) then rand_like(min_rot) over ALL envs (imu_sensor.py:144-153)
    def draw_reset_randoms(self, env_ids):
        rs = self.g["random_source"]
        rs.rand_into(self._u_bias, tag="imu_bias_init")
        rs.rand_into(self._u_rot, tag="imu_mount")

    def reset_masked(self):
        env = self.g["env_manager"]
        p = _lib.dptr
        strict = bool(self.g.get("strict_rng", False))
        _lib.check(
            env._lib.agx_imu_reset(env._buffers, self.num_envs, self._args, p(self._u_bias) if strict else None,
                                   p(self._u_rot) if strict else None, p(self.bias), p(self.sensor_quats), env._stream()),
            "agx_imu_reset",
        )

    def update(self, k_substeps):
        """k x IMUSensor.update() (called once per physics sub-step in the reference, robot_manager.py:491-495)"""
        env = self.g["env_manager"]
        p = _lib.dptr
        zn = zb = None
        if bool(self.g.get("strict_rng", False)) and k_substeps > 0:
            # per sub-step: sample_noise -> randn(N, 6), update_bias -> randn(N, 6)
            rs, N = self.g["random_source"], self.num_envs
            if self._z is None or self._z.shape[0] != 2 * k_substeps:
                self._z = torch.zeros(2 * k_substeps, N, 6, device=self.device)
            rs.normal_into(self._z, tag="imu_normals")
            zn = p(self._z[2 * (k_substeps - 1)])
            self._zb = self._z[1::2].contiguous()
            zb = p(self._zb)
        _lib.check(
            env._lib.agx_imu_update(env._buffers, self.num_envs, int(k_substeps), self._args, p(self.sensor_quats), zn, zb,
                                    p(self.bias), p(self.imu_meas), env._stream()),
            "agx_imu_update",
        )

    def get_observation(self):
        return self.imu_meas
