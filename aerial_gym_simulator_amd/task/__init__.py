"""Task registration (aerial_gym/task/__init__.py)."""
from ..config.task_config import (
    fully_actuated_lidar_navigation_task_config,
    lidar_navigation_task_config,
    navigation_task_config,
    position_setpoint_task_config,
)
from ..registry.task_registry import task_registry
from .lidar_navigation_task import LiDARNavigationTask
from .navigation_task import NavigationTask
from .position_setpoint_task import PositionSetpointTask

task_registry.register_task("position_setpoint_task", PositionSetpointTask, position_setpoint_task_config)
task_registry.register_task("navigation_task", NavigationTask, navigation_task_config)
task_registry.register_task("lidar_navigation_task", LiDARNavigationTask, lidar_navigation_task_config)
# BASELINE configs[3] as written (fully-actuated octarotor + 32 x 512 LiDAR): the reference's NavigationTask on that robot
task_registry.register_task("navigation_task_fully_actuated_lidar", NavigationTask, fully_actuated_lidar_navigation_task_config)
