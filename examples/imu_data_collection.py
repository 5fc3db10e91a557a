"""IMU stream of a hovering quadrotor (reference: examples/imu_data_collection.py)."""
import os
import torch

import aerial_gym_simulator_amd  # noqa: F401
from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

if __name__ == "__main__":
    env = SimBuilder().build_env(sim_name="base_sim", env_name="empty_env", robot_name="base_quadrotor_with_imu",
                                 controller_name="lee_position_control", args=None, device="cuda:0", num_envs=4, headless=True, use_warp=False)
    env.reset()
    g = env.get_obs()
    target = torch.cat([g["robot_position"].clone(), torch.zeros(env.num_envs, 1, device="cuda:0")], dim=1)
    log = []
    for i in range(int(os.environ.get("AGX_EXAMPLE_STEPS", 2000))):
        env.step(actions=target)
        log.append(g["imu_measurement"][0].clone())
    data = torch.stack(log)[500:]
    print("accelerometer mean", data[:, 0:3].mean(0).tolist(), "std", data[:, 0:3].std(0).tolist())
    print("gyro          mean", data[:, 3:6].mean(0).tolist(), "std", data[:, 3:6].std(0).tolist())
