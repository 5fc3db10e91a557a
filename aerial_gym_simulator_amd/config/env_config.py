"""Environment configs (aerial_gym/config/env_config/{empty_env,env_with_obstacles}.py)."""
from . import asset_config as A


class EmptyEnvCfg:
    class env:
        num_envs = 3
        num_env_actions = 0
        env_spacing = 1.0
        num_physics_steps_per_env_step_mean = 1
        num_physics_steps_per_env_step_std = 0
        render_viewer_every_n_steps = 10
        collision_force_threshold = 0.010
        manual_camera_trigger = False
        reset_on_collision = True
        create_ground_plane = False
        sample_timestep_for_latency = True
        perturb_observations = True
        keep_same_env_for_num_episodes = 1
        write_to_sim_at_every_timestep = False
        use_warp = False
        e_s = env_spacing
        lower_bound_min = lower_bound_max = [-e_s, -e_s, -e_s]
        upper_bound_min = upper_bound_max = [e_s, e_s, e_s]

    class env_config:
        include_asset_type = {}
        asset_type_to_dict_map = {}


_WALLS = {
    "left_wall": A.left_wall,
    "right_wall": A.right_wall,
    "back_wall": A.back_wall,
    "front_wall": A.front_wall,
    "bottom_wall": A.bottom_wall,
    "top_wall": A.top_wall,
}


class EnvWithObstaclesCfg:
    class env:
        num_envs = 64
        num_env_actions = 4
        env_spacing = 5.0
        num_physics_steps_per_env_step_mean = 10
        num_physics_steps_per_env_step_std = 0
        render_viewer_every_n_steps = 1
        reset_on_collision = True
        collision_force_threshold = 0.05
        create_ground_plane = False
        sample_timestep_for_latency = True
        perturb_observations = True
        keep_same_env_for_num_episodes = 1
        write_to_sim_at_every_timestep = False
        use_warp = True
        lower_bound_min, lower_bound_max = [-2.0, -4.0, -3.0], [-1.0, -2.5, -2.0]
        upper_bound_min, upper_bound_max = [9.0, 2.5, 2.0], [10.0, 4.0, 3.0]

    class env_config:
        include_asset_type = dict({"panels": True, "objects": True}, **{k: True for k in _WALLS})
        asset_type_to_dict_map = dict({"panels": A.panel_asset_params, "objects": A.object_asset_params}, **_WALLS)


class EnvWithRandomBoxesCfg(EnvWithObstaclesCfg):
    """BASELINE configs 3-5: 100 random boxes + the 6 wall slabs."""

    class env(EnvWithObstaclesCfg.env):
        pass

    class env_config:
        include_asset_type = dict({"boxes": True}, **{k: True for k in _WALLS})
        asset_type_to_dict_map = dict({"boxes": A.random_box_asset_params}, **_WALLS)


class EnvWithLidarNavObstaclesCfg(EnvWithObstaclesCfg):
    """env_with_lidar_nav_obstacles.py:20-82: 15 panels + 70 objects + 6 walls in a 10-15 m cube."""

    class env(EnvWithObstaclesCfg.env):
        lower_bound_min, lower_bound_max = [-7.50, -7.50, -5.0], [-5.0, -5.0, -3.0]
        upper_bound_min, upper_bound_max = [5.0, 5.0, 3.0], [7.5, 7.5, 5.0]

    class env_config:
        include_asset_type = dict({"panels": True, "objects": True}, **{k: True for k in _WALLS})
        asset_type_to_dict_map = dict({"panels": A.lidar_nav_panel_asset_params, "objects": A.lidar_nav_object_asset_params},
                                      **A.lidar_nav_walls)


class DynamicEnvironmentCfg(EnvWithObstaclesCfg):
    """dynamic_environment.py: 35 free objects, no walls, obstacle twists as env actions (6 per obstacle)."""

    class env(EnvWithObstaclesCfg.env):
        num_env_actions = 6
        create_ground_plane = True  # (no effect here: there is no ground contact model)
        write_to_sim_at_every_timestep = True
        lower_bound_min, lower_bound_max = [-2.0, -4.0, 0.0], [-1.0, -2.5, 0.0]
        upper_bound_min, upper_bound_max = [9.0, 2.5, 4.0], [10.0, 4.0, 5.0]

    class env_config:
        include_asset_type = {"objects": True}
        asset_type_to_dict_map = {"objects": A.object_asset_params}


class ForestEnvCfg(EnvWithObstaclesCfg):
    """forest_env.py: a multi-link cylinder tree + 35 objects over a floor slab, 10 x 10 x 4 m."""

    class env(EnvWithObstaclesCfg.env):
        collision_force_threshold = 0.005
        lower_bound_min = lower_bound_max = [-5.0, -5.0, -1.0]
        upper_bound_min = upper_bound_max = [5.0, 5.0, 3.0]

    class env_config:
        include_asset_type = {"trees": True, "objects": True, "bottom_wall": True}
        asset_type_to_dict_map = {"trees": A.tree_asset_params, "objects": A.object_asset_params, "bottom_wall": A.bottom_wall}
