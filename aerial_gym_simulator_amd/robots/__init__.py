"""Robot registration (names of aerial_gym/robots/__init__.py:37-66 that are in scope)."""
from ..config.robot_config import (
    BaseOctarotorCfg,
    BaseOctarotorWithLidar32x512Cfg,
    BaseQuadCfg,
    BaseQuadRootLinkControlCfg,
    BaseQuadWithCamera64x48Cfg,
    BaseQuadWithCameraCfg,
    BaseQuadWithCameraImuCfg,
    BaseQuadWithImuCfg,
    BaseQuadWithFaceIDNormalCameraCfg,
    BaseQuadWithLidarCfg,
    BaseQuadWithStereoCameraCfg,
    LMF2Cfg,
    LMF2With64x48CameraCfg,
    MagpieCfg,
)
from ..registry.robot_registry import robot_registry
from .base_multirotor import BaseMultirotor

robot_registry.register("base_quadrotor", BaseMultirotor, BaseQuadCfg)
robot_registry.register("base_octarotor", BaseMultirotor, BaseOctarotorCfg)
robot_registry.register("base_quadrotor_with_camera", BaseMultirotor, BaseQuadWithCameraCfg)
robot_registry.register("base_quadrotor_with_camera_64x48", BaseMultirotor, BaseQuadWithCamera64x48Cfg)
robot_registry.register("base_quadrotor_with_lidar", BaseMultirotor, BaseQuadWithLidarCfg)
robot_registry.register("base_octarotor_with_lidar_32x512", BaseMultirotor, BaseOctarotorWithLidar32x512Cfg)
robot_registry.register("base_quadrotor_with_faceid_normal_camera", BaseMultirotor, BaseQuadWithFaceIDNormalCameraCfg)
robot_registry.register("base_quadrotor_with_stereo_camera", BaseMultirotor, BaseQuadWithStereoCameraCfg)
robot_registry.register("magpie", BaseMultirotor, MagpieCfg)
robot_registry.register("base_quadrotor_with_imu", BaseMultirotor, BaseQuadWithImuCfg)
robot_registry.register("base_quadrotor_with_camera_imu", BaseMultirotor, BaseQuadWithCameraImuCfg)
robot_registry.register("lmf2", BaseMultirotor, LMF2Cfg)
robot_registry.register("lmf2_with_camera_64x48", BaseMultirotor, LMF2With64x48CameraCfg)
robot_registry.register("base_quad_root_link_control", BaseMultirotor, BaseQuadRootLinkControlCfg)
