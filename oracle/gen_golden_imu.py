"""TEST INFRASTRUCTURE -- golden vectors for the IMU sensor, by running the reference's
aerial_gym/sensors/imu_sensor.py (build container only).   python oracle/gen_golden_imu.py

imu_sensor.npz: IMUSensor.update() over a chain of physics sub-steps (bias random walk + noise), in both
frame conventions, plus reset_idx; the normal / uniform draws are replayed from the same seeds and stored."""
import math
import os

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  FIRST: it switches TorchScript off before torch is imported (cr_torch.py)

import numpy as np
import torch

import ref_shells

OUT = gg.OUT


def main():
    ref_shells.install()
    imu = ref_shells.ref("sensors.imu_sensor")
    from aerial_gym.config.sensor_config.imu_config.base_imu_config import BaseImuConfig

    rng = torch.Generator().manual_seed(99)
    n, K = 64, 5
    out = {}
    for tag, world_frame in (("body", False), ("world", True)):
        class Cfg(BaseImuConfig):
            pass

        Cfg.world_frame = world_frame
        Cfg.randomize_placement = True
        state = gg.random_state(n, rng)
        g = {"robot_position": state[:, 0:3], "robot_orientation": state[:, 3:7], "gravity": torch.tensor([0.0, 0.0, -9.81]).expand(n, -1),
             "dt": 0.01, "robot_mass": torch.full((n,), 1.24), "robot_linvel": state[:, 7:10], "robot_angvel": state[:, 10:13],
             "robot_body_angvel": torch.randn(n, 3, generator=rng), "robot_body_linvel": torch.randn(n, 3, generator=rng),
             "robot_euler_angles": torch.zeros(n, 3), "force_sensor_tensor": torch.zeros(n, 6)}
        torch.manual_seed(7)
        s = imu.IMUSensor(Cfg, n, "cpu")
        s.init_tensors(g)
        torch.manual_seed(8)
        s.reset()
        torch.manual_seed(8)  # replay reset(): rand_like(bias) then torch_rand_float_tensor(min, max) = rand_like
        u_bias = torch.rand(n, 6)
        u_rot = torch.rand(n, 3)
        forces, wbodies, quats, meas, z_noise, z_bias = [], [], [], [], [], []
        bias0 = s.bias.clone()
        for k in range(K):
            g["force_sensor_tensor"][:, 0:3] = torch.randn(n, 3, generator=rng) * 6.0
            g["robot_body_angvel"][:] = torch.randn(n, 3, generator=rng)
            g["robot_orientation"][:] = gg.random_state(n, rng)[:, 3:7]
            if k == K - 1:
                g["force_sensor_tensor"][:8, 0:3] *= 40.0  # exercise the +-100 / +-10 clamps
                g["robot_body_angvel"][:8] *= 12.0
            forces.append(g["force_sensor_tensor"][:, 0:3].clone())
            wbodies.append(g["robot_body_angvel"].clone())
            quats.append(g["robot_orientation"].clone())
            torch.manual_seed(100 + k)
            s.update()
            torch.manual_seed(100 + k)  # sample_noise then update_bias: two randn((n, 6))
            z_noise.append(torch.randn(n, 6))
            z_bias.append(torch.randn(n, 6))
            meas.append(s.imu_meas.clone())
        st = lambda l: torch.stack(l).numpy()  # noqa: E731
        out.update({f"{tag}_force": st(forces), f"{tag}_wbody": st(wbodies), f"{tag}_quat": st(quats), f"{tag}_meas": st(meas),
                    f"{tag}_z_noise": st(z_noise), f"{tag}_z_bias": st(z_bias), f"{tag}_bias0": bias0.numpy(),
                    f"{tag}_bias_end": s.bias.numpy().copy(), f"{tag}_sensor_quat": s.sensor_quats.numpy().copy(),
                    f"{tag}_u_bias": u_bias.numpy(), f"{tag}_u_rot": u_rot.numpy()})
    cfg = BaseImuConfig
    out.update(mass=np.float32(1.24), dt=np.float32(0.01), bias_std=np.array(cfg.bias_std, np.float32),
               noise_std=np.array(cfg.imu_noise_std, np.float32), max_value=np.array(cfg.max_measurement_value, np.float32),
               max_bias_init=np.array(cfg.max_bias_init_value, np.float32),
               min_rot_deg=np.array(cfg.min_euler_rotation_deg, np.float32), max_rot_deg=np.array(cfg.max_euler_rotation_deg, np.float32))
    np.savez(os.path.join(OUT, "imu_sensor.npz"), **out)
    print("imu_sensor: ok  meas[0,0] =", out["body_meas"][0, 0], " |bias| end", np.abs(out["body_bias_end"]).max())


if __name__ == "__main__":
    main()
