"""Ray-cast sensor configs (aerial_gym/config/sensor_config/{camera,lidar}_config/*.py)."""


class BaseSensorConfig:
    num_sensors = 1
    randomize_placement = False
    min_translation, max_translation = [0.07, -0.06, 0.01], [0.12, 0.03, 0.04]
    min_euler_rotation_deg, max_euler_rotation_deg = [-5.0, -5.0, -5.0], [5.0, 5.0, 5.0]


def _oor(max_range, normalize_range):
    # base_depth_camera_config.py:46-52: [-1] U [0, 1] after normalisation
    return (max_range, -max_range) if normalize_range else (-1.0, -1.0)


class BaseDepthCameraConfig(BaseSensorConfig):  # base_depth_camera_config.py:5-71
    sensor_type = "camera"
    height, width = 135, 240
    horizontal_fov_deg = 87.0
    max_range, min_range = 10.0, 0.2
    calculate_depth = True
    return_pointcloud = False
    pointcloud_in_world_frame = False
    segmentation_camera = True
    euler_frame_rot_deg = [-90.0, 0, -90.0]
    normalize_range = True
    far_out_of_range_value, near_out_of_range_value = _oor(max_range, normalize_range)
    randomize_placement = True
    nominal_position = [0.10, 0.0, 0.03]
    nominal_orientation_euler_deg = [0.0, 0.0, 0.0]
    use_collision_geometry = False

    class sensor_noise:
        enable_sensor_noise = False
        pixel_dropout_prob = 0.01
        pixel_std_dev_multiplier = 0.01
        std_a = std_b = std_c = 0.0
        mean_offset = 0.0


class DepthCamera64x48Config(BaseDepthCameraConfig):
    """BASELINE config 3: 64 x 48 depth + segmentation camera."""

    height, width = 48, 64


class StereoCameraConfig(BaseDepthCameraConfig):  # stereo_camera_config.py:4-9
    sensor_type = "stereo_camera"
    height, width = 270, 480
    baseline = -0.095  # distance from the left to the right camera in metres, +y is positive


class BaseNormalFaceIDCameraConfig(BaseDepthCameraConfig):  # base_normal_faceID_camera_config.py:7-45
    sensor_type = "normal_faceID_camera"
    height, width = 270, 480
    return_pointcloud = True  # normal information comes in the form of a point cloud
    normal_in_world_frame = True
    randomize_placement = False


class BaseLidarConfig(BaseSensorConfig):  # base_lidar_config.py:5-76
    sensor_type = "lidar"
    height, width = 128, 512
    horizontal_fov_deg_min, horizontal_fov_deg_max = -180, 180
    vertical_fov_deg_min, vertical_fov_deg_max = -45, 45
    max_range, min_range = 10.0, 0.2
    return_pointcloud = False
    pointcloud_in_world_frame = False
    segmentation_camera = True
    euler_frame_rot_deg = [0.0, 0.0, 0.0]
    normalize_range = True
    far_out_of_range_value, near_out_of_range_value = _oor(max_range, normalize_range)
    randomize_placement = True
    nominal_position = [0.10, 0.0, 0.03]
    nominal_orientation_euler_deg = [0.0, 0.0, 0.0]

    class sensor_noise:
        enable_sensor_noise = True
        std_a = std_b = std_c = 0.00001
        mean_offset = -0.05
        pixel_dropout_prob = 0.0


class Lidar32x512Config(BaseLidarConfig):
    """BASELINE config 4: 32 beams x 512 points."""

    height, width = 32, 512

    class sensor_noise(BaseLidarConfig.sensor_noise):
        enable_sensor_noise = False


class RSLidar_Airy_Config(BaseLidarConfig):  # rslidar_airy_config.py:4-35 (dome LiDAR of the LiDAR-navigation task)
    height, width = 48, 120
    horizontal_fov_deg_min, horizontal_fov_deg_max = -180, 180
    vertical_fov_deg_min, vertical_fov_deg_max = 0, 90
    max_range, min_range = 10.0, 0.2
    return_pointcloud = True
    segmentation_camera = False
    normalize_range = False
    pointcloud_in_world_frame = True
    far_out_of_range_value, near_out_of_range_value = _oor(max_range, normalize_range)
    randomize_placement = True
    min_translation = max_translation = [-0.05, 0.0, 0.0]
    min_euler_rotation_deg = max_euler_rotation_deg = [0.0, -90.0, 0.0]  # front-mounted dome, looking up -> forward

    class sensor_noise:
        enable_sensor_noise = False
        std_a, std_b, std_c = 0.00038089, -0.00343351, 0.01553284
        mean_offset = -0.025
        pixel_dropout_prob = 0.0


class BaseNormalFaceIDLidarConfig(BaseLidarConfig):
    """sensor_type the reference's WarpSensor accepts (warp_sensor.py:63-70) without shipping a config."""

    sensor_type = "normal_faceID_lidar"
    return_pointcloud = True
    normal_in_world_frame = True

    class sensor_noise(BaseLidarConfig.sensor_noise):
        enable_sensor_noise = False


class BaseImuConfig(BaseSensorConfig):  # imu_config/base_imu_config.py:4-62 (a VN100-like IMU)
    sensor_type = "imu"
    world_frame = False
    enable_noise = True
    enable_bias = True
    bias_std = [9.782812831313576e-07] * 3 + [2.6541629581345176e-05] * 3
    imu_noise_std = [0.001688956233495657] * 3 + [0.0010679343003532472] * 3
    max_measurement_value = [100.0, 100.0, 100.0, 10.0, 10.0, 10.0]
    max_bias_init_value = [1.0e-03] * 6
    gravity_compensation = False
    randomize_placement = False
    min_euler_rotation_deg, max_euler_rotation_deg = [-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]
