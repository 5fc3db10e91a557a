"""Task registration (aerial_gym/task/__init__.py)."""
from ..config.task_config import navigation_task_config, position_setpoint_task_config
from ..registry.task_registry import task_registry
from .navigation_task import NavigationTask
from .position_setpoint_task import PositionSetpointTask

task_registry.register_task("position_setpoint_task", PositionSetpointTask, position_setpoint_task_config)
task_registry.register_task("navigation_task", NavigationTask, navigation_task_config)
