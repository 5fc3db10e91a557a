"""Lee controller gains (aerial_gym/config/controller_config/*.py)."""
import numpy as np


class lee_controller_config:  # lee_controller_config.py:19-45
    num_actions = 4
    max_inclination_angle_rad = np.pi / 3.0
    max_yaw_rate = np.pi / 3.0
    K_pos_tensor_max, K_pos_tensor_min = [3.0, 3.0, 2.0], [2.0, 2.0, 1.0]
    K_vel_tensor_max, K_vel_tensor_min = [3.0, 3.0, 3.0], [2.0, 2.0, 2.0]
    K_rot_tensor_max, K_rot_tensor_min = [1.2, 1.2, 0.6], [0.8, 0.8, 0.4]
    K_angvel_tensor_max, K_angvel_tensor_min = [0.2, 0.2, 0.2], [0.1, 0.1, 0.1]
    randomize_params = False


class lee_controller_config_octarotor:  # lee_controller_config_octarotor.py
    num_actions = 4
    max_inclination_angle_rad = np.pi / 3.0
    max_yaw_rate = np.pi / 3.0
    K_pos_tensor_max, K_pos_tensor_min = [3.0, 3.0, 2.0], [2.0, 2.0, 1.0]
    K_vel_tensor_max, K_vel_tensor_min = [3.0, 3.0, 3.0], [2.0, 2.0, 2.0]
    K_rot_tensor_max, K_rot_tensor_min = [10.2, 10.2, 5.6], [10.8, 10.8, 5.4]
    K_angvel_tensor_max, K_angvel_tensor_min = [2.2, 2.2, 2.2], [2.1, 2.1, 2.1]
    randomize_params = True


class magpie_controller_config:  # magpie_controller_config.py:4-43
    num_actions = 4
    max_inclination_angle_rad = np.pi / 3.0
    max_yaw_rate = np.pi / 3.0
    K_pos_tensor_max, K_pos_tensor_min = [2.0, 2.0, 1.0], [2.0, 2.0, 1.0]
    K_vel_tensor_max, K_vel_tensor_min = [3.3, 3.3, 2.6], [2.7, 2.7, 2.3]
    K_rot_tensor_max = [12.9453125, 12.9453125, 0.32499998807907104]
    K_rot_tensor_min = [8.9453125, 8.9453125, 0.32499998807907104]
    K_angvel_tensor_max = [0.8910937666893005, 0.8910937666893005, 0.048818358927965164]
    K_angvel_tensor_min = [0.65910937666893005, 0.65910937666893005, 0.028818358927965164]
    randomize_params = True


class lmf2_controller_config:  # lmf2_controller_config.py:4-43 (K_vel z: max 1.3 < min 1.7 in the reference, kept as written)
    num_actions = 4
    max_inclination_angle_rad = np.pi / 3.0
    max_yaw_rate = np.pi / 3.0
    K_pos_tensor_max, K_pos_tensor_min = [2.0, 2.0, 1.0], [2.0, 2.0, 1.0]
    K_vel_tensor_max, K_vel_tensor_min = [3.3, 3.3, 1.3], [2.7, 2.7, 1.7]
    K_rot_tensor_max, K_rot_tensor_min = [1.85, 1.85, 0.4], [1.6, 1.6, 0.25]
    K_angvel_tensor_max, K_angvel_tensor_min = [0.5, 0.5, 0.09], [0.4, 0.4, 0.075]
    randomize_params = True


class fully_actuated_controller_config:  # fully_actuated_controller_rov.py
    num_actions = 7
    max_inclination_angle_rad = np.pi / 3.0
    max_yaw_rate = np.pi / 3.0
    K_pos_tensor_max, K_pos_tensor_min = [1.0, 1.0, 1.0], [1.0, 1.0, 1.0]
    K_vel_tensor_max, K_vel_tensor_min = [8.0, 8.0, 8.0], [8.0, 8.0, 8.0]
    K_rot_tensor_max, K_rot_tensor_min = [2.2, 2.2, 2.6], [2.2, 2.2, 2.6]
    K_angvel_tensor_max, K_angvel_tensor_min = [2.2, 2.2, 2.2], [2.1, 2.1, 2.1]
    randomize_params = True


class no_control_config:  # no_control_config.py (num_actions is overwritten with num_motors)
    num_actions = 4
