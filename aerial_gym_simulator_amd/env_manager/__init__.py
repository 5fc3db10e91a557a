"""Env / sim config registration (aerial_gym/env_manager/__init__.py, aerial_gym/sim/__init__.py)."""
from ..config.env_config import DynamicEnvironmentCfg, EmptyEnvCfg, ForestEnvCfg, EnvWithLidarNavObstaclesCfg, EnvWithObstaclesCfg, EnvWithRandomBoxesCfg
from ..config.sim_config import BaseSimConfig
from ..registry.env_registry import env_config_registry
from ..registry.sim_registry import sim_config_registry

env_config_registry.register("empty_env", EmptyEnvCfg)
env_config_registry.register("env_with_obstacles", EnvWithObstaclesCfg)
env_config_registry.register("env_with_random_boxes", EnvWithRandomBoxesCfg)
env_config_registry.register("env_with_lidar_nav_obstacles", EnvWithLidarNavObstaclesCfg)
env_config_registry.register("dynamic_env", DynamicEnvironmentCfg)
env_config_registry.register("forest_env", ForestEnvCfg)
sim_config_registry.register("base_sim", BaseSimConfig)
