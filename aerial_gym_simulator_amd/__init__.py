"""aerial_gym_simulator_amd -- MI355X-native (gfx950, HIP) implementation of the Aerial Gym
per-env simulation step behind the reference's Task / EnvManager / registry API.

    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    task = task_registry.make_task("position_setpoint_task", num_envs=8192)
    obs, rew, term, trunc, info = task.step(actions)
"""
import os

AERIAL_GYM_DIRECTORY = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))

from . import control, env_manager, robots, task  # noqa: E402,F401  (populate the registries)
from .registry.registries import (  # noqa: E402,F401
    controller_registry,
    env_config_registry,
    robot_registry,
    sim_config_registry,
    task_registry,
)
