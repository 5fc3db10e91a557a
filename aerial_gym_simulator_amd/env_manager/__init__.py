"""Env / sim config registration (aerial_gym/env_manager/__init__.py, aerial_gym/sim/__init__.py)."""
from ..config.env_config import DynamicEnvironmentCfg, EmptyEnvCfg, ForestEnvCfg, EnvWithLidarNavObstaclesCfg, EnvWithObstaclesCfg, EnvWithRandomBoxesCfg
from ..config.sim_config import BaseSimConfig, BaseSimHeadlessConfig, BaseSimNoGravityConfig, SimCfg2Ms, SimCfg4Ms
from ..registry.env_registry import env_config_registry
from ..registry.sim_registry import sim_config_registry

env_config_registry.register("empty_env", EmptyEnvCfg)
env_config_registry.register("env_with_obstacles", EnvWithObstaclesCfg)
env_config_registry.register("env_with_random_boxes", EnvWithRandomBoxesCfg)
env_config_registry.register("env_with_lidar_nav_obstacles", EnvWithLidarNavObstaclesCfg)
env_config_registry.register("dynamic_env", DynamicEnvironmentCfg)
env_config_registry.register("forest_env", ForestEnvCfg)
sim_config_registry.register("base_sim", BaseSimConfig)
sim_config_registry.register("base_sim_headless", BaseSimHeadlessConfig)  # aerial_gym/sim/__init__.py:13-16
sim_config_registry.register("base_sim_2ms", SimCfg2Ms)
sim_config_registry.register("base_sim_4ms", SimCfg4Ms)
sim_config_registry.register("base_sim_no_gravity", BaseSimNoGravityConfig)  # (the reference leaves this one to the example that uses it)
