"""Controller registration, same names as aerial_gym/control/__init__.py:42-100."""
from ..config.controller_config import (
    fully_actuated_controller_config,
    lee_controller_config,
    lee_controller_config_octarotor,
    lmf2_controller_config,
    magpie_controller_config,
    no_control_config,
)
from ..registry.controller_registry import controller_registry
from .control_allocation import ControlAllocator  # noqa: F401
from .controllers import (
    FullyActuatedController,
    LeeAccelerationController,
    LeeAttitudeController,
    LeePositionController,
    LeeRatesController,
    LeeVelocityController,
    LeeVelocitySteeringAngleController,
    NoControl,
)
from .motor_model import MotorModel  # noqa: F401

controller_registry.register_controller("no_control", NoControl, no_control_config)
_LEE = (
    ("position", LeePositionController),
    ("velocity", LeeVelocityController),
    ("attitude", LeeAttitudeController),
    ("rates", LeeRatesController),
    ("acceleration", LeeAccelerationController),
)


def register_robot_controllers(robot_name=None, controller_config=None):
    for kind, cls in _LEE:
        controller_registry.register_controller(f"{robot_name}_{kind}_control", cls, controller_config)


register_robot_controllers("lee", lee_controller_config)
register_robot_controllers("octarotor", lee_controller_config_octarotor)
register_robot_controllers("magpie", magpie_controller_config)
register_robot_controllers("lmf2", lmf2_controller_config)  # control/__init__.py:91-93
controller_registry.register_controller(
    "lee_velocity_steering_angle_control", LeeVelocitySteeringAngleController, lee_controller_config
)
controller_registry.register_controller(
    "rov_fully_actuated_control", FullyActuatedController, fully_actuated_controller_config
)
