"""Robot configs: base_quadrotor and base_octarotor
(aerial_gym/config/robot_config/base_quad_config.py, base_octarotor_config.py) plus the URDF
constants of resources/robots/{quad,octarotor}/*.urdf as data (``robot_model``)."""
import numpy as np

from .sensor_config import (
    BaseDepthCameraConfig,
    BaseImuConfig,
    BaseLidarConfig,
    BaseNormalFaceIDCameraConfig,
    DepthCamera64x48Config,
    Lidar32x512Config,
    RSLidar_Airy_Config,
    StereoCameraConfig,
)

_PI = float(np.pi)


def _state_ratio(lo_xyz, hi_xyz, yaw):
    lo = list(lo_xyz) + [0, 0, -yaw, 1.0] + [0] * 6
    hi = list(hi_xyz) + [0, 0, yaw, 1.0] + [0] * 6
    return lo, hi


class _CommonAsset:
    base_link_name = "base_link"
    disable_gravity = False
    collapse_fixed_joints = False
    fix_base_link = False
    collision_mask = 0
    density = 0.000001
    max_angular_velocity = 100.0
    max_linear_velocity = 100.0
    armature = 0.001
    semantic_id = 0
    per_link_semantic = False
    min_state_ratio, max_state_ratio = _state_ratio([0.1, 0.1, 0.1], [0.3, 0.9, 0.9], _PI)
    max_force_and_torque_disturbance = [0.1, 0.1, 0.1, 0.05, 0.05, 0.05]
    keep_in_env = True
    min_euler_angles, max_euler_angles = [-_PI] * 3, [_PI] * 3


class BaseQuadCfg:
    class init_config:  # [ratio xyz, roll, pitch, yaw, 1.0, v xyz, w xyz]
        min_init_state = [0.1, 0.15, 0.15, 0, 0, -_PI / 6, 1.0, -0.2, -0.2, -0.2, -0.2, -0.2, -0.2]
        max_init_state = [0.2, 0.85, 0.85, 0, 0, _PI / 6, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2]

    class sensor_config:
        enable_camera = False
        camera_config = BaseDepthCameraConfig
        enable_lidar = False
        lidar_config = BaseLidarConfig
        enable_imu = False
        imu_config = None

    class disturbance:
        enable_disturbance = False
        prob_apply_disturbance = 0.02
        max_force_and_torque_disturbance = [0.75, 0.75, 0.75, 0.004, 0.004, 0.004]

    class damping:  # body-frame drag coefficients, all zero for the base quad
        linvel_linear_damping_coefficient = [0.0, 0.0, 0.0]
        linvel_quadratic_damping_coefficient = [0.0, 0.0, 0.0]
        angular_linear_damping_coefficient = [0.0, 0.0, 0.0]
        angular_quadratic_damping_coefficient = [0.0, 0.0, 0.0]

    class robot_asset(_CommonAsset):
        file = "quad.urdf"
        name = "base_quadrotor"
        angular_damping = 0.01
        linear_damping = 0.01

    class robot_model:  # resources/robots/quad/quad.urdf
        base_mass = 0.225
        base_inertia = [[4.225e-4, 0, 0], [0, 4.225e-4, 0], [0, 0, 8.45e-4]]
        collision_sphere_radius = 0.18384776310850237
        motor_mass = 0.00625
        motor_xyz = [[0.13, -0.13, 0.0], [-0.13, -0.13, 0.0], [-0.13, 0.13, 0.0], [0.13, 0.13, 0.0]]
        motor_rpy = [[0.0, 0.0, 0.0]] * 4

    class control_allocator_config:
        num_motors = 4
        force_application_level = "motor_link"
        application_mask = [1 + 4 + i for i in range(4)]
        motor_directions = [1, -1, 1, -1]
        allocation_matrix = [
            [0.0, 0.0, 0.0, 0.0],
            [0.0, 0.0, 0.0, 0.0],
            [1.0, 1.0, 1.0, 1.0],
            [-0.13, -0.13, 0.13, 0.13],
            [-0.13, 0.13, 0.13, -0.13],
            [-0.01, 0.01, -0.01, 0.01],
        ]

        class motor_model_config:
            use_rps = True
            motor_thrust_constant_min, motor_thrust_constant_max = 0.00000926312, 0.00001826312
            motor_time_constant_increasing_min = motor_time_constant_increasing_max = 0.04
            motor_time_constant_decreasing_min = motor_time_constant_decreasing_max = 0.04
            max_thrust, min_thrust = 2, 0
            max_thrust_rate = 100000.0
            thrust_to_torque_ratio = 0.01
            use_discrete_approximation = True


class BaseQuadWithCameraCfg(BaseQuadCfg):
    class sensor_config(BaseQuadCfg.sensor_config):
        enable_camera = True
        camera_config = BaseDepthCameraConfig


class BaseQuadWithCamera64x48Cfg(BaseQuadCfg):
    class sensor_config(BaseQuadCfg.sensor_config):
        enable_camera = True
        camera_config = DepthCamera64x48Config


class BaseQuadWithImuCfg(BaseQuadCfg):  # base_quad_config.py:196-199
    class sensor_config(BaseQuadCfg.sensor_config):
        enable_imu = True
        imu_config = BaseImuConfig


class BaseQuadWithCameraImuCfg(BaseQuadCfg):  # base_quad_config.py:207-213
    class sensor_config(BaseQuadCfg.sensor_config):
        enable_camera = True
        camera_config = BaseDepthCameraConfig
        enable_imu = True
        imu_config = BaseImuConfig


class BaseQuadWithLidarCfg(BaseQuadCfg):
    class sensor_config(BaseQuadCfg.sensor_config):
        enable_lidar = True
        lidar_config = BaseLidarConfig


class BaseQuadWithFaceIDNormalCameraCfg(BaseQuadCfg):  # base_quad_config.py:220-223
    class sensor_config(BaseQuadCfg.sensor_config):
        enable_camera = True
        camera_config = BaseNormalFaceIDCameraConfig


class BaseQuadWithStereoCameraCfg(BaseQuadCfg):  # base_quad_config.py:225-228
    class sensor_config(BaseQuadCfg.sensor_config):
        enable_camera = True
        camera_config = StereoCameraConfig


_S = 0.17320508075688776
_OCTA_RPY = [
    [0.5369911762672153, 0.8339952782025826, -2.726309414768267],
    [-2.191861365711223, -0.12515405937833668, -0.17271936472604107],
    [2.191861365711223, 0.1251540593783369, -0.17271936472604107],
    [-0.5369911762672153, -0.8339952782025827, -2.726309414768267],
]


class BaseOctarotorCfg:
    class init_config:
        min_init_state = [0.0, 0.0, 0.0, 0, 0, -_PI, 1.0, -0.2, -0.2, -0.2, -0.2, -0.2, -0.2]
        max_init_state = [1.0, 1.0, 1.0, 0, 0, _PI, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2]

    class sensor_config(BaseQuadCfg.sensor_config):
        pass

    class disturbance:
        enable_disturbance = True
        prob_apply_disturbance = 0.05
        max_force_and_torque_disturbance = [1.5, 1.5, 1.5, 0.25, 0.25, 0.25]

    class damping(BaseQuadCfg.damping):
        pass

    class robot_asset(_CommonAsset):
        file = "octarotor.urdf"
        name = "base_octarotor"
        angular_damping = 0.0000001
        linear_damping = 0.0000001

    class robot_model:  # resources/robots/octarotor/octarotor.urdf
        base_mass = 0.3
        base_inertia = [[0.048000000000000015, 0, 0], [0, 0.048000000000000015, 0], [0, 0, 0.048000000000000015]]
        collision_sphere_radius = 0.30000000000000004
        motor_mass = 0.1
        motor_xyz = [[sx * _S, sy * _S, sz * _S] for sz in (1, -1) for sy in (1, -1) for sx in (1, -1)]
        motor_rpy = _OCTA_RPY + _OCTA_RPY[::-1]

    class control_allocator_config:
        num_motors = 8
        force_application_level = "motor_link"
        application_mask = [1 + 8 + i for i in range(8)]
        motor_directions = [1, -1, 1, -1, 1, -1, 1, -1]
        _a, _b, _c = 0.78867513, 0.21132487, 0.57735027
        _d, _e, _f, _g = 0.14226497, 0.21547005, 0.25773503, 0.01547005
        _h, _i = 0.11547005, 0.23094011
        allocation_matrix = [
            [-_a, _b, -_b, _a, _a, -_b, _b, -_a],
            [_b, _a, -_a, -_b, -_b, -_a, _a, _b],
            [_c, -_c, -_c, _c, _c, -_c, -_c, _c],
            [_d, -_e, _f, _g, -_g, -_f, _e, -_d],
            [-_f, _g, _d, _e, -_e, -_d, -_g, _f],
            [_h, -_i, -_h, _i, -_i, _h, _i, -_h],
        ]

        class motor_model_config:
            use_rps = False
            motor_thrust_constant_min, motor_thrust_constant_max = 0.00000926312, 0.00001826312
            motor_time_constant_increasing_min, motor_time_constant_increasing_max = 0.01, 0.03
            motor_time_constant_decreasing_min = motor_time_constant_decreasing_max = 0.005
            max_thrust, min_thrust = 6.25, -6.25
            max_thrust_rate = 100000.0
            thrust_to_torque_ratio = 0.01
            use_discrete_approximation = True


class BaseOctarotorWithLidar32x512Cfg(BaseOctarotorCfg):
    class sensor_config(BaseOctarotorCfg.sensor_config):
        enable_lidar = True
        lidar_config = Lidar32x512Config


class MagpieCfg:  # magpie_config.py:15-176, resources/robots/magpie/model.urdf
    class init_config:
        min_init_state = [0.1, 0.15, 0.15, 0, 0, -_PI, 1.0, -0.2, -0.2, -0.2, -0.2, -0.2, -0.2]
        max_init_state = [0.2, 0.85, 0.85, 0, 0, _PI, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2]

    class sensor_config:
        enable_camera = False
        camera_config = BaseDepthCameraConfig
        enable_lidar = True
        lidar_config = RSLidar_Airy_Config
        enable_imu = False
        imu_config = None

    class disturbance:
        enable_disturbance = True
        prob_apply_disturbance = 0.05
        max_force_and_torque_disturbance = [4.75, 4.75, 4.75, 0.03, 0.03, 0.03]

    class damping:
        linvel_linear_damping_coefficient = [0.0, 0.0, 0.0]
        linvel_quadratic_damping_coefficient = [0.0, 0.0, 0.0]
        angular_linear_damping_coefficient = [0.0, 0.0, 0.0]
        angular_quadratic_damping_coefficient = [0.0, 0.0, 0.0]

    class robot_asset(_CommonAsset):
        file = "model.urdf"
        name = "base_quadrotor"
        collapse_fixed_joints = True  # the prop joints carry dont_collapse="true": 5 bodies remain
        angular_damping = 0.01
        linear_damping = 0.01
        min_state_ratio, max_state_ratio = _state_ratio([0.1, 0.1, 0.1], [0.9, 0.9, 0.9], _PI)

    class robot_model:  # model.urdf: base 1.2 kg + 4 props of 10 g at (+-0.1, +-0.1, 0)
        base_mass = 1.2
        base_inertia = [[0.013, 0, 0], [0, 0.014, 0], [0, 0, 0.013]]
        # the URDF collision shape is a 0.7 x 0.7 x 0.5 box; the collision model here is a sphere (DESIGN.md
        # "Collision"): radius = half the box width
        collision_sphere_radius = 0.35
        motor_mass = 0.01
        motor_inertia = 0.000001
        motor_xyz = [[0.1, 0.1, 0.0], [0.1, -0.1, 0.0], [-0.1, -0.1, 0.0], [-0.1, 0.1, 0.0]]
        motor_rpy = [[0.0, 0.0, 0.0]] * 4

    class control_allocator_config:
        num_motors = 4
        force_application_level = "base_link"  # not "motor_link": the combined wrench A u acts on the root link
        application_mask = [1 + 4 + i for i in range(4)]
        motor_directions = [1, -1, 1, -1]
        allocation_matrix = [
            [0.0, 0.0, 0.0, 0.0],
            [0.0, 0.0, 0.0, 0.0],
            [1.0, 1.0, 1.0, 1.0],
            [-0.13, -0.13, 0.13, 0.13],
            [-0.13, 0.13, 0.13, -0.13],
            [-0.02, 0.02, -0.02, 0.02],
        ]

        class motor_model_config:
            use_rps = True
            motor_thrust_constant_min, motor_thrust_constant_max = 0.00000926312, 0.00001826312
            motor_time_constant_increasing_min, motor_time_constant_increasing_max = 0.01, 0.02
            motor_time_constant_decreasing_min, motor_time_constant_decreasing_max = 0.005, 0.015
            max_thrust, min_thrust = 12.0, 0.1
            max_thrust_rate = 1000000.0
            thrust_to_torque_ratio = 0.02
            use_discrete_approximation = True


class LMF2Cfg(MagpieCfg):  # lmf2_config.py:18-180, resources/robots/lmf2/model.urdf -- the robot of the reference's default navigation recipe
    """The reference's `navigation_task` default (navigation_task_config.py:9-10: lmf2 + lmf2_velocity_control).  Same airframe
    layout as magpie in the URDF (base 1.2 kg + four 10 g props at (+-0.1, +-0.1, 0), fixed joints kept: 5 bodies); the
    allocator applies the combined wrench at the root body (force_application_level "base_link": anything but "motor_link",
    control_allocation.py:53-65).  Pinned by tests/golden/robot_lmf2.npz (composite of the reference's URDF)."""

    class init_config:
        min_init_state = [0.1, 0.15, 0.15, 0, 0, -_PI / 6, 1.0, -0.2, -0.2, -0.2, -0.2, -0.2, -0.2]
        max_init_state = [0.2, 0.85, 0.85, 0, 0, _PI / 6, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2]

    class sensor_config:
        enable_camera = True
        camera_config = BaseDepthCameraConfig  # 135 x 240 depth + segmentation
        enable_lidar = False
        lidar_config = BaseLidarConfig
        enable_imu = False
        imu_config = None

    class robot_model(MagpieCfg.robot_model):
        # the URDF collision shape is a 0.5 m cube; the collision model here is a sphere of half its width (DESIGN.md "Collision")
        collision_sphere_radius = 0.25

    class control_allocator_config:
        num_motors = 4
        force_application_level = "base_link"
        application_mask = [1 + 4 + i for i in range(4)]
        motor_directions = [1, -1, 1, -1]
        allocation_matrix = [
            [0.0, 0.0, 0.0, 0.0],
            [0.0, 0.0, 0.0, 0.0],
            [1.0, 1.0, 1.0, 1.0],
            [-0.13, -0.13, 0.13, 0.13],
            [-0.13, 0.13, 0.13, -0.13],
            [-0.07, 0.07, -0.07, 0.07],
        ]

        class motor_model_config:
            use_rps = True
            motor_thrust_constant_min, motor_thrust_constant_max = 0.00000926312, 0.00001826312
            motor_time_constant_increasing_min, motor_time_constant_increasing_max = 0.05, 0.08
            motor_time_constant_decreasing_min, motor_time_constant_decreasing_max = 0.005, 0.005
            max_thrust, min_thrust = 10.0, 0.1
            max_thrust_rate = 100000.0
            thrust_to_torque_ratio = 0.07
            use_discrete_approximation = True


class LMF2With64x48CameraCfg(LMF2Cfg):
    """lmf2 with the 64 x 48 camera of BASELINE configs[2] instead of its 135 x 240 one"""

    class sensor_config(LMF2Cfg.sensor_config):
        camera_config = DepthCamera64x48Config


class BaseQuadRootLinkControlCfg(BaseQuadCfg):  # base_quad_root_link_control_config.py:18-53
    """base_quadrotor with the allocator's wrench applied at the root link.  (The reference points this config at
    resources/robots/quad/model.urdf -- the lmf2-style 1.2 kg body; the rigid-body constants here are that URDF's.)"""

    class robot_asset(BaseQuadCfg.robot_asset):
        file = "model.urdf"
        collapse_fixed_joints = False

    class robot_model(MagpieCfg.robot_model):
        collision_sphere_radius = 0.2  # the URDF collision shape is a 0.4 m cube

    class control_allocator_config(BaseQuadCfg.control_allocator_config):
        force_application_level = "root_link"

        class motor_model_config(BaseQuadCfg.control_allocator_config.motor_model_config):
            motor_thrust_constant_min, motor_thrust_constant_max = 0.00001826312, 0.00001826312
            motor_time_constant_increasing_min, motor_time_constant_increasing_max = 0.01, 0.03
            motor_time_constant_decreasing_min, motor_time_constant_decreasing_max = 0.005, 0.005
            max_thrust, min_thrust = 10, 0
