"""Task configs (aerial_gym/config/task_config/{position_setpoint,navigation}_task_config.py)."""
import torch


class position_setpoint_task_config:
    seed = 1
    sim_name = "base_sim"
    env_name = "empty_env"
    robot_name = "base_quadrotor"
    controller_name = "lee_attitude_control"
    args = {}
    num_envs = 4096
    use_warp = False
    headless = True
    device = "cuda:0"
    observation_space_dim = 13
    privileged_observation_space_dim = 0
    action_space_dim = 4
    episode_len_steps = 500
    return_state_before_reset = False
    reward_parameters = {}  # the reference's table is unused by its compute_reward (hard-coded constants)


class navigation_task_config:
    seed = -1
    sim_name = "base_sim"
    env_name = "env_with_random_boxes"
    robot_name = "base_quadrotor_with_camera_64x48"
    controller_name = "lee_velocity_control"
    args = {}
    num_envs = 1024
    use_warp = True
    headless = True
    device = "cuda:0"
    # the reference packs 13 + 4 state/action entries + 64 VAE latents; the VAE encoder is a
    # dense conv net outside the simulation hot path, so `use_vae = False` gives 17 + pooled depth
    observation_space_dim = 13 + 4 + 64
    privileged_observation_space_dim = 0
    action_space_dim = 4
    episode_len_steps = 100
    return_state_before_reset = False
    target_min_ratio = [0.90, 0.1, 0.1]
    target_max_ratio = [0.94, 0.90, 0.90]

    reward_parameters = {
        "pos_reward_magnitude": 5.0,
        "pos_reward_exponent": 1.0 / 3.5,
        "very_close_to_goal_reward_magnitude": 5.0,
        "very_close_to_goal_reward_exponent": 2.0,
        "getting_closer_reward_multiplier": 10.0,
        "x_action_diff_penalty_magnitude": 0.8,
        "x_action_diff_penalty_exponent": 3.333,
        "z_action_diff_penalty_magnitude": 0.8,
        "z_action_diff_penalty_exponent": 5.0,
        "yawrate_action_diff_penalty_magnitude": 0.8,
        "yawrate_action_diff_penalty_exponent": 3.33,
        "x_absolute_action_penalty_magnitude": 0.1,
        "x_absolute_action_penalty_exponent": 0.3,
        "z_absolute_action_penalty_magnitude": 1.5,
        "z_absolute_action_penalty_exponent": 1.0,
        "yawrate_absolute_action_penalty_magnitude": 1.5,
        "yawrate_absolute_action_penalty_exponent": 2.0,
        "collision_penalty": -100.0,
    }
    REWARD_PARAMETER_ORDER = tuple(reward_parameters.keys())

    class vae_config:
        use_vae = False
        latent_dims = 64  # min-pooled 8x8 depth grid takes the latents' place (see navigation_task.py here)
        image_res = (270, 480)
        interpolation_mode = "nearest"
        return_sampled_latent = True

    class curriculum:
        min_level = 15
        max_level = 50
        check_after_log_instances = 2048
        increase_step = 2
        decrease_step = 1
        success_rate_for_increase = 0.7
        success_rate_for_decrease = 0.6

    @staticmethod
    def action_transformation_function(action):
        """navigation_task_config.py:87-117: (speed, inclination, yaw-rate) -> velocity command."""
        # the reference's arithmetic with fewer launches (9 instead of 17): `* max_speed / 2.0` is `* 2.0 / 2.0`, exact in
        # binary floating point, and the products are written straight into their column
        a = torch.clamp(action, -1.0, 1.0)
        max_yawrate, max_inclination = torch.pi / 3, torch.pi / 4
        speed = a[:, 0] + 1.0
        inclination = max_inclination * a[:, 1]
        out = torch.zeros((a.shape[0], 4), device=a.device)
        torch.mul(speed, torch.cos(inclination), out=out[:, 0])
        torch.mul(speed, torch.sin(inclination), out=out[:, 2])
        torch.mul(a[:, 2], max_yawrate, out=out[:, 3])
        return out

    @staticmethod
    def action_transformation_function_as_written(action):
        """the same function in the reference's own wording (navigation_task_config.py:87-117); tests compare the two"""
        a = torch.clamp(action, -1.0, 1.0)
        max_speed, max_yawrate, max_inclination = 2.0, torch.pi / 3, torch.pi / 4
        speed = a[:, 0] + 1.0
        out = torch.zeros((a.shape[0], 4), device=a.device)
        out[:, 0] = speed * torch.cos(max_inclination * a[:, 1]) * max_speed / 2.0
        out[:, 2] = speed * torch.sin(max_inclination * a[:, 1]) * max_speed / 2.0
        out[:, 3] = a[:, 2] * max_yawrate
        return out


class fully_actuated_lidar_navigation_task_config(navigation_task_config):
    """BASELINE configs[3] as written: "fully-actuated octarotor + 32-beam x 512 LiDAR depth+seg".  Assembled from
    reference pieces only: NavigationTask (reward / obs / curriculum), `base_octarotor` (base_octarotor_config.py)
    with a 32 x 512 range + segmentation LiDAR (base_lidar_config.py), and FullyActuatedController
    (fully_actuated_control.py:14-32, registered as `rov_fully_actuated_control`, control/__init__.py:98-100), whose
    command is 7-D: world-frame position set-point + orientation set-point (xyzw).  The policy's 4-D action maps to
    that command the way the reference's tasks map theirs (a static function of the action alone)."""

    robot_name = "base_octarotor_with_lidar_32x512"
    controller_name = "rov_fully_actuated_control"
    action_space_dim = 4

    @staticmethod
    def action_transformation_function(action):
        """(x, y, z, yaw) in [-1, 1] -> position set-point within +-(5, 5, 2.5) m, level attitude at yaw * pi."""
        a = torch.clamp(action, -1.0, 1.0)
        out = torch.zeros((a.shape[0], 7), device=a.device)
        torch.mul(a[:, 0:2], 5.0, out=out[:, 0:2])
        torch.mul(a[:, 2], 2.5, out=out[:, 2])
        half = 0.5 * torch.pi * a[:, 3]
        torch.sin(half, out=out[:, 5])
        torch.cos(half, out=out[:, 6])
        return out


class lidar_navigation_task_config:  # lidar_navigation_task_config.py:5-108
    seed = -1
    sim_name = "base_sim"
    env_name = "env_with_lidar_nav_obstacles"
    robot_name = "magpie"
    controller_name = "magpie_acceleration_control"
    args = {}
    num_envs = 512
    use_warp = True
    headless = True
    device = "cuda:0"
    observation_space_dim = 13 + 4 + 16 * 20  # root state + actions + 3 x 6 min-pooled 48 x 120 LiDAR image
    privileged_observation_space_dim = 0
    action_space_dim = 4
    episode_len_steps = 110
    return_state_before_reset = False
    target_min_ratio = [0.90, 0.15, 0.15]
    target_max_ratio = [0.92, 0.80, 0.80]
    lidar_pool = (3, 6)        # max_pool2d kernel of process_image_observation (:346-347)
    lidar_low_noise_row0 = 10  # ds_lidar_data[:, 10:] (:302-308)

    reward_parameters = {
        "pos_reward_magnitude": 3.0,
        "pos_reward_exponent": 1.0,
        "very_close_to_goal_reward_magnitude": 5.0,
        "very_close_to_goal_reward_exponent": 8.0,
        "vel_direction_component_reward_magnitude": 1.0,
        "x_action_diff_penalty_magnitude": 0.3,
        "x_action_diff_penalty_exponent": 5.0,
        "y_action_diff_penalty_magnitude": 0.3,
        "y_action_diff_penalty_exponent": 5.0,
        "z_action_diff_penalty_magnitude": 0.3,
        "z_action_diff_penalty_exponent": 5.0,
        "yawrate_action_diff_penalty_magnitude": 0.3,
        "yawrate_action_diff_penalty_exponent": 5.0,
        "x_absolute_action_penalty_magnitude": 0.1,
        "x_absolute_action_penalty_exponent": 0.3,
        "y_absolute_action_penalty_magnitude": 0.1,
        "y_absolute_action_penalty_exponent": 0.3,
        "z_absolute_action_penalty_magnitude": 0.15,
        "z_absolute_action_penalty_exponent": 1.0,
        "yawrate_absolute_action_penalty_magnitude": 0.15,
        "yawrate_absolute_action_penalty_exponent": 2.0,
        "collision_penalty": -10.0,
    }
    REWARD_PARAMETER_ORDER = tuple(reward_parameters.keys())

    class vae_config:
        use_vae = False

    class curriculum:
        min_level = 25
        max_level = 70
        check_after_log_instances = 2048
        increase_step = 2
        decrease_step = 1
        success_rate_for_increase = 0.7
        success_rate_for_decrease = 0.6

    @staticmethod
    def action_transformation_function(action):
        """lidar_navigation_task_config.py:98-108: +-2 m/s^2 acceleration command, +-pi/3 rad/s yaw rate."""
        a = torch.clamp(action, -1.0, 1.0)
        out = torch.empty((a.shape[0], 4), device=a.device)
        torch.mul(a[:, 0:3], 2, out=out[:, 0:3])
        torch.mul(a[:, 3], torch.pi / 3, out=out[:, 3])
        return out


# the built-in transformations have a one-launch device form (agx_action_transform); a task uses it when the config still
# carries THE function object below (anything a user puts there instead runs as the torch code it is)
navigation_task_config.action_transformation_function.agx_kind = (1, 4)
fully_actuated_lidar_navigation_task_config.action_transformation_function.agx_kind = (3, 7)
lidar_navigation_task_config.action_transformation_function.agx_kind = (2, 4)
