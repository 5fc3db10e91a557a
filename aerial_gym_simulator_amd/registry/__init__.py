from .registries import (  # noqa: F401
    controller_registry,
    env_config_registry,
    robot_registry,
    sim_config_registry,
    task_registry,
)
