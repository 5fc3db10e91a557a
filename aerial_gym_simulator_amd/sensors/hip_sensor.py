"""Ray-cast sensor front-end: WarpSensor + WarpCam / WarpStereoCam / WarpLidar /
WarpNormalFaceID{Cam,Lidar} of the reference (sensors/warp/warp_sensor.py:26-249, warp_cam.py:31-182,
warp_stereo_cam.py, warp_lidar.py:40-191, warp_normal_faceID_{cam,lidar}.py) on the HIP kernels.
Capture = agx_sensor_pose -> agx_raycast_{camera,stereo_camera,lidar} -> agx_sensor_postprocess[_points]."""
import ctypes as C
import math

import torch

from .. import _lib
from ..utils.math import quat_from_euler_xyz, quat_from_euler_xyz_tensor

RAY_MODE = {"range": 0, "depth": 1, "pointcloud": 2, "pointcloud_world": 3, "normal": 4, "normal_world": 5}
SENSOR_TYPES = ("camera", "stereo_camera", "lidar", "normal_faceID_camera", "normal_faceID_lidar")  # warp_sensor.py:36-81


def pinhole_kinv(width, height, horizontal_fov_deg):
    """((K_inv[0][0], K_inv[0][2], K_inv[1][1], K_inv[1][2]), c_x, c_y) of warp_cam.py:31-64 -- the four entries of K_inv that
    wp.transform_vector(K_inv, (x, y, 1)) touches, with the arithmetic the reference gets them by: `wp.mat44(...)` rounds
    alpha_u, alpha_v, u_0, v_0 to float32, and `wp.inverse` (warp/native/mat.h, adapted from USD's GfMatrix4f::Inverse) inverts
    THAT matrix by cofactors: for K = [[a, 0, u, 0], [0, b, v, 0], [0, 0, 1, 0], [0, 0, 0, 1]] the determinant is the float32
    product a b, its reciprocal is taken in double, and the cofactors b, -b u, a, -a v (float32 roundings of double products)
    are multiplied by it in double and rounded once.  1 / alpha_u evaluated in double from the unrounded alpha_u (rounds 1-2)
    differs from this in the last bit for most resolutions (tests/golden/warp_kernels_camera.npz holds the reference's values)."""
    import numpy as np

    W, H = width, height
    u0, v0 = W / 2, H / 2
    hfov = math.radians(horizontal_fov_deg)
    f = W / 2 * 1 / math.tan(hfov / 2)
    vfov = 2 * math.atan(H / (2 * f))
    alpha_u, alpha_v = u0 / math.tan(hfov / 2), v0 / math.tan(vfov / 2)
    f32, f64 = np.float32, np.float64
    a, b, u, v = f32(alpha_u), f32(alpha_v), f32(u0), f32(v0)
    rcp = f64(1.0) / f64(a * b)  # a * b: one float32 product
    entry = lambda cof: float(f32(f64(f32(cof)) * rcp))  # noqa: E731
    k = (entry(f64(b)), entry(-(f64(b) * f64(u))), entry(f64(a)), entry(-(f64(a) * f64(v))))
    return k, int(u0), int(v0)


class HipSensor:
    def __init__(self, sensor_config, num_envs, scene, device):
        self.cfg, self.num_envs, self.scene, self.device = sensor_config, num_envs, scene, device
        self.num_sensors = sensor_config.num_sensors
        cfg = sensor_config
        if cfg.sensor_type not in SENSOR_TYPES:
            raise NotImplementedError(f"sensor_type {cfg.sensor_type}")  # like warp_sensor.py:80-81
        self.is_lidar = cfg.sensor_type in ("lidar", "normal_faceID_lidar")
        self.is_normal = cfg.sensor_type.startswith("normal_faceID")
        self.is_stereo = cfg.sensor_type == "stereo_camera"
        if self.is_normal:
            # normals come as a "point cloud"; the segmentation image carries face indices
            if not cfg.return_pointcloud:
                raise ValueError("normal_faceID sensors need return_pointcloud = True (pixels are vec3)")
            self.mode = RAY_MODE["normal_world" if cfg.normal_in_world_frame else "normal"]
        elif cfg.return_pointcloud:
            self.mode = RAY_MODE["pointcloud_world" if cfg.pointcloud_in_world_frame else "pointcloud"]
        elif self.is_lidar:
            self.mode = RAY_MODE["range"]
        else:
            self.mode = RAY_MODE["depth" if cfg.calculate_depth else "range"]
        if self.is_lidar:
            self._init_ray_table()
        else:
            self._init_intrinsics()

    # warp_cam.py:31-64
    def _init_intrinsics(self):
        k, self.c_x, self.c_y = pinhole_kinv(self.cfg.width, self.cfg.height, self.cfg.horizontal_fov_deg)
        self.kinv = (C.c_float * 4)(*k)

    # warp_lidar.py:40-64
    def _init_ray_table(self):
        cfg = self.cfg
        H, W = cfg.height, cfg.width
        hmin, hmax = math.radians(cfg.horizontal_fov_deg_min), math.radians(cfg.horizontal_fov_deg_max)
        vmin, vmax = math.radians(cfg.vertical_fov_deg_min), math.radians(cfg.vertical_fov_deg_max)
        if hmax - hmin > 2 * math.pi:
            raise ValueError("Horizontal FOV must be less than 2pi")
        if vmax - vmin > math.pi:
            raise ValueError("Vertical FOV must be less than pi")
        az = torch.tensor([hmax - (hmax - hmin) * (j / (W - 1)) for j in range(W)], dtype=torch.float64)
        el = torch.tensor([vmax - (vmax - vmin) * (i / (H - 1)) for i in range(H)], dtype=torch.float64)
        rv = torch.stack(
            [torch.cos(az)[None, :] * torch.cos(el)[:, None], torch.sin(az)[None, :] * torch.cos(el)[:, None],
             torch.sin(el)[:, None].expand(H, W)], dim=2,
        ).to(torch.float32)
        rv = rv / torch.norm(rv, dim=2, keepdim=True)
        self.ray_vectors = rv.contiguous().to(self.device)

    def init_tensors(self, global_tensor_dict):
        g, N, S, dev, cfg = global_tensor_dict, self.num_envs, self.num_sensors, self.device, self.cfg
        self.g = g
        self.pixels = g["depth_range_pixels"]
        self.segmentation_pixels = g["segmentation_pixels"] if cfg.segmentation_camera else None
        self.min_translation = torch.tensor(cfg.min_translation, device=dev)
        self.max_translation = torch.tensor(cfg.max_translation, device=dev)
        self.min_rotation = torch.deg2rad(torch.tensor(cfg.min_euler_rotation_deg, device=dev))
        self.max_rotation = torch.deg2rad(torch.tensor(cfg.max_euler_rotation_deg, device=dev))
        frame = quat_from_euler_xyz_tensor(torch.deg2rad(torch.tensor(cfg.euler_frame_rot_deg, dtype=torch.float32)))
        self.frame_quat = (C.c_float * 4)(*[float(x) for x in frame])
        self.sensor_local_position = torch.zeros(N, S, 3, device=dev)
        self.sensor_local_orientation = torch.zeros(N, S, 4, device=dev)
        mean_rot = (self.min_rotation + self.max_rotation) / 2.0
        self.sensor_local_orientation[:] = quat_from_euler_xyz(mean_rot[0], mean_rot[1], mean_rot[2])
        self.sensor_position = torch.zeros(N, S, 3, device=dev)
        self.sensor_orientation = torch.zeros(N, S, 4, device=dev)
        self.sensor_orientation[..., 3] = 1.0
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
        self._min_t, self._max_t = f3(cfg.min_translation), f3(cfg.max_translation)
        self._min_r = f3([math.radians(x) for x in cfg.min_euler_rotation_deg])
        self._max_r = f3([math.radians(x) for x in cfg.max_euler_rotation_deg])
        self._u_pos = torch.zeros(N, S, 3, device=dev)
        self._u_rot = torch.zeros(N, S, 3, device=dev)
        self._noise_z = self._noise_u = None
        if cfg.sensor_noise.enable_sensor_noise:
            self._noise_z = torch.zeros_like(self.pixels)
            self._noise_u = torch.zeros_like(self.pixels)
        g["sensor_position"], g["sensor_orientation"] = self.sensor_position, self.sensor_orientation

    # warp_sensor.py:153-172
    def draw_reset_randoms(self, env_ids):
        """strict_rng: rand_like over the reset envs only, translation then rotation."""
        if not self.cfg.randomize_placement:
            return
        rs, n = self.g["random_source"], len(env_ids)
        self._u_pos[env_ids] = rs.rand(n, self.num_sensors, 3, tag="sensor_pos")
        self._u_rot[env_ids] = rs.rand(n, self.num_sensors, 3, tag="sensor_rot")

    def reset_masked(self):
        if not self.cfg.randomize_placement:
            return
        g = self.g
        env = g["env_manager"]
        p = _lib.dptr
        strict = bool(g.get("strict_rng", True))
        _lib.check(
            env._lib.agx_sensor_mount_reset(env._buffers, self.num_envs, self.num_sensors, self._min_t, self._max_t, self._min_r, self._max_r,
                                            p(self._u_pos) if strict else None, p(self._u_rot) if strict else None,
                                            p(self.sensor_local_position), p(self.sensor_local_orientation), env._stream()),
            "agx_sensor_mount_reset",
        )

    def update(self):
        """WarpSensor.update (warp_sensor.py:177-200): pose -> ray-cast -> noise / range limits / normalise."""
        self.compose_pose()
        fused = self.limits_fusable()
        self.raycast(fuse_limits=fused)
        if not fused:
            self.postprocess()

    def limits_fusable(self):
        """Scalar images without sensor noise: the range limits and the normalisation are applied by the ray-cast kernel
        to the pixel it is about to store (AgxRangeLimits), the image is not read and written a second time."""
        cfg = self.cfg
        return not cfg.sensor_noise.enable_sensor_noise and not self.is_normal and not cfg.return_pointcloud

    def _limits(self):
        cfg = self.cfg
        return _lib.AgxRangeLimits(float(cfg.min_range), float(cfg.max_range), float(cfg.far_out_of_range_value),
                                   float(cfg.near_out_of_range_value), int(bool(cfg.normalize_range)))

    def compose_pose(self):
        env = self.g["env_manager"]
        if env._sensor_pose_fresh:  # this step's poses were written by the fused robot-side launch (agx_nav_robot_side)
            env._sensor_pose_fresh = False
            return
        p = _lib.dptr
        _lib.check(
            env._lib.agx_sensor_pose(env._buffers, self.num_envs, self.num_sensors, p(self.sensor_local_position),
                                     p(self.sensor_local_orientation), self.frame_quat, p(self.sensor_position),
                                     p(self.sensor_orientation), env._stream()),
            "agx_sensor_pose",
        )

    def raycast(self, stream=None, fuse_limits=False):
        """fuse_limits=False leaves the raw distances of the reference's warp kernels in `pixels` (postprocess() follows)."""
        env = self.g["env_manager"]
        lim = C.byref(self._limits()) if fuse_limits else None
        lib, p, cfg, sc = env._lib, _lib.dptr, self.cfg, self.scene
        stream = stream if stream is not None else env._stream()
        N, S = self.num_envs, self.num_sensors
        seg = p(self.segmentation_pixels) if self.segmentation_pixels is not None else None
        if self.is_lidar:
            return _lib.check(
                lib.agx_raycast_lidar(N, S, cfg.width, cfg.height, p(self.ray_vectors), float(cfg.max_range), self.mode,
                                      p(self.sensor_position), p(self.sensor_orientation), p(sc.tri_world), p(sc.tri_seg),
                                      p(sc.bvh_nodes), sc.num_tris, p(self.pixels), seg, lim, stream),
                "agx_raycast_lidar",
            )
        if self.is_stereo:
            return _lib.check(
                lib.agx_raycast_stereo_camera(N, S, cfg.width, cfg.height, self.kinv, float(cfg.max_range), float(cfg.baseline),
                                              self.c_x, self.c_y, self.mode, p(self.sensor_position), p(self.sensor_orientation),
                                              p(sc.tri_world), p(sc.tri_seg), p(sc.bvh_nodes), sc.num_tris, p(self.pixels), seg,
                                              lim, stream),
                "agx_raycast_stereo_camera",
            )
        return _lib.check(
            lib.agx_raycast_camera(N, S, cfg.width, cfg.height, self.kinv, float(cfg.max_range), self.c_x, self.c_y, self.mode,
                                   p(self.sensor_position), p(self.sensor_orientation), p(sc.tri_world), p(sc.tri_seg),
                                   p(sc.bvh_nodes), sc.num_tris, p(self.pixels), seg, lim, stream),
            "agx_raycast_camera",
        )

    def postprocess(self):
        """apply_noise for every sensor type; apply_range_limits + normalize_observation only for
        camera / lidar / stereo_camera (warp_sensor.py:197-200)."""
        env = self.g["env_manager"]
        lib, p, cfg = env._lib, _lib.dptr, self.cfg
        zn = ud = None
        sn = cfg.sensor_noise
        if sn.enable_sensor_noise:
            rs = self.g["random_source"]
            zn = p(rs.normal_into(self._noise_z, tag="sensor_noise_z"))
            ud = p(rs.rand_into(self._noise_u, tag="sensor_noise_u"))
        elif self.is_normal or (cfg.return_pointcloud and cfg.pointcloud_in_world_frame):
            return  # nothing to do: no noise, and these images are neither range-limited nor normalised
        noise = (float(getattr(sn, "std_a", 0.0)), float(getattr(sn, "std_b", 0.0)), float(getattr(sn, "std_c", 0.0)),
                 float(getattr(sn, "mean_offset", 0.0)), float(sn.pixel_dropout_prob))
        limits = (float(cfg.min_range), float(cfg.max_range), float(cfg.far_out_of_range_value), float(cfg.near_out_of_range_value))
        if cfg.return_pointcloud:
            use_limits = not self.is_normal and not cfg.pointcloud_in_world_frame
            _lib.check(
                lib.agx_sensor_postprocess_points(self.pixels.numel() // 3, p(self.pixels), zn, ud, *noise, *limits,
                                                  int(use_limits), int(bool(cfg.normalize_range)), env._stream()),
                "agx_sensor_postprocess_points",
            )
            return
        _lib.check(
            lib.agx_sensor_postprocess(self.pixels.numel(), p(self.pixels), zn, ud, *noise, *limits,
                                       int(bool(cfg.normalize_range)), env._stream()),
            "agx_sensor_postprocess",
        )

    def get_observation(self):
        return self.pixels, self.segmentation_pixels


WarpSensor = HipSensor  # reference class name
