"""aerial_gym/robots/robot_manager.py (RobotManagerIGE) without Isaac Gym: creates the robot
through the registry, allocates the action tensors and the sensor image tensors
(robot_manager.py:108-272), dispatches sensor capture (robot_manager.py:497-501)."""
import torch

from ..registry.robot_registry import robot_registry
from ..tensors import aos_view, soa
from ..utils.logging import CustomLogger

logger = CustomLogger("robot_manager")


class RobotManagerHIP:
    def __init__(self, global_tensor_dict, env_config, robot_name, controller_name, device):
        self.env_config, self.device = env_config, device
        self.num_envs = env_config.env.num_envs
        self.use_warp = env_config.env.use_warp
        self.robot, self.cfg = robot_registry.make_robot(robot_name, controller_name, env_config, device)
        sc = self.cfg.sensor_config
        if sc.enable_camera and sc.enable_lidar:
            raise ValueError("Both camera and lidar are enabled; they share the same image tensors (robot_manager.py:66-90)")
        if (sc.enable_camera or sc.enable_lidar) and not self.use_warp:
            raise ValueError("ray-cast sensors need use_warp=True (the rasteriser camera of Isaac Gym is out of scope)")
        self.warp_sensor = None
        self.imu_sensor = None
        self.has_IGE_sensors = False

    def prepare_for_sim(self, global_tensor_dict, scene):
        g, N, dev = global_tensor_dict, self.num_envs, self.device
        self.global_tensor_dict = g
        A = self.robot.num_actions
        self.actions_soa, self.prev_actions_soa = soa(A, N, dev), soa(A, N, dev)
        g["robot_actions_soa"], g["robot_prev_actions_soa"] = self.actions_soa, self.prev_actions_soa
        g["robot_actions"], g["robot_prev_actions"] = aos_view(self.actions_soa), aos_view(self.prev_actions_soa)
        self.actions, self.prev_actions = g["robot_actions"], g["robot_prev_actions"]
        g["dof_control_mode"] = "none"
        # per-body force / torque tensors (IGE_env_manager.py:293-358; written by robot.step(), each body's wrench in its own
        # frame).  The fused step never touches them; they carry the robot plug-in path (EnvManager: a robot class that
        # overrides step()).  Bodies: the root link, then whatever lies below the highest index of the application mask.
        ca = self.cfg.control_allocator_config
        self.num_robot_bodies = int(getattr(self.cfg.robot_asset, "num_bodies", 0) or (max(int(b) for b in ca.application_mask) + 1))
        g["robot_force_tensor"] = torch.zeros(N, self.num_robot_bodies, 3, device=dev)
        g["robot_torque_tensor"] = torch.zeros(N, self.num_robot_bodies, 3, device=dev)
        g["num_robot_bodies"] = self.num_robot_bodies
        self.robot.init_tensors(g)
        pd = self.robot.params_dict
        # (allocated by the robot before its controller's init_tensors, which reads them like the reference's controllers do)
        g["robot_mass"][:] = pd["mass"]
        g["robot_inertia"] = g["robot_inertia"] if "robot_inertia" in g else torch.tensor(pd["inertia"], device=dev).view(1, 3, 3).expand(N, 3, 3)
        self.robot_masses, self.robot_inertias = g["robot_mass"], g["robot_inertia"]
        sc = self.cfg.sensor_config
        if self.use_warp and (sc.enable_camera or sc.enable_lidar):
            from ..sensors.hip_sensor import HipSensor

            cfg = sc.camera_config if sc.enable_camera else sc.lidar_config
            shape = (N, cfg.num_sensors, cfg.height, cfg.width)
            if cfg.return_pointcloud:
                shape = shape + (3,)
            g["depth_range_pixels"] = torch.zeros(shape, device=dev)
            g["segmentation_pixels"] = (
                torch.zeros((N, cfg.num_sensors, cfg.height, cfg.width), dtype=torch.int32, device=dev)
                if cfg.segmentation_camera else None
            )
            self.warp_sensor = HipSensor(cfg, N, scene, dev)
            self.warp_sensor.init_tensors(g)
        if getattr(sc, "enable_imu", False):  # robot_manager.py:256-268
            from ..sensors.imu_sensor import IMUSensor

            self.imu_sensor = IMUSensor(sc.imu_config, N, dev)
            self.imu_sensor.init_tensors(g)

    def draw_sensor_reset_randoms(self, env_ids):
        if self.warp_sensor is not None:
            self.warp_sensor.draw_reset_randoms(env_ids)
        if self.imu_sensor is not None:
            self.imu_sensor.draw_reset_randoms(env_ids)

    def reset_sensors_masked(self):
        if self.warp_sensor is not None:
            self.warp_sensor.reset_masked()
        if self.imu_sensor is not None:
            self.imu_sensor.reset_masked()

    def post_physics_step(self, k_substeps):
        """the reference updates the IMU after every physics sub-step (robot_manager.py:491-495)"""
        if self.imu_sensor is not None:
            self.imu_sensor.update(k_substeps)

    def reset(self):
        self.robot.reset()

    def reset_idx(self, env_ids):
        self.robot.reset_idx(env_ids)

    def capture_sensors(self):
        if self.warp_sensor is not None:
            self.warp_sensor.update()


RobotManagerIGE = RobotManagerHIP  # reference class name
