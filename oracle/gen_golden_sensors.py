"""Golden vectors for the torch / python half of the reference's Warp sensors (rows a23-a25): camera
intrinsics (warp_cam.py:31-64), LiDAR ray table (warp_lidar.py:40-64) and the post-processing chain
apply_noise -> apply_range_limits -> normalize_observation (warp_sensor.py:196-247), produced by RUNNING the
reference's own classes.  warp-lang is not installed here: a stand-in `warp` module lets the reference modules
import (decorators and type annotations evaluate to inert objects; mat44 / inverse / from_torch carry plain
numpy / torch values).  The ray-cast kernels themselves are never executed -- they need Warp.

    python oracle/gen_golden_sensors.py        (in the build container: needs /root/reference)
"""
import os
import sys
import types

import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402  FIRST: it switches TorchScript off before torch is imported (cr_torch.py)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shells  # noqa: E402

OUT = gg.OUT


class _Inert:
    """Anything the reference touches on `warp` at import time: attribute -> inert, call -> inert (a decorator
    call hands back the function)."""

    def __getattr__(self, name):
        return _Inert()

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not isinstance(a[0], _Inert) and not k:
            return a[0]
        return _Inert()

    def __getitem__(self, k):
        return _Inert()


def install_warp_stub():
    if "warp" in sys.modules:
        return sys.modules["warp"]
    wp = types.ModuleType("warp")
    inert = _Inert()
    wp.__getattr__ = lambda name: getattr(inert, name)  # PEP 562
    wp.mat44 = lambda *a: np.array(a, dtype=np.float64).reshape(4, 4)
    wp.inverse = lambda m: np.linalg.inv(m)
    wp.from_torch = lambda t, dtype=None: t
    wp.constant = lambda x: x
    sys.modules["warp"] = wp
    return wp


def lidar_tables():
    install_warp_stub()
    WarpLidar = ref_shells.ref("sensors.warp.warp_lidar").WarpLidar
    out = {}
    for mod, cls in (("base_lidar_config", "BaseLidarConfig"), ("osdome_64_config", "OSDome_64_Config"),
                     ("rslidar_airy_config", "RSLidar_Airy_Config")):
        try:
            cfg = getattr(ref_shells.ref("config.sensor_config.lidar_config." + mod), cls)
        except (ImportError, AttributeError) as e:
            print("skip", mod, e)
            continue
        lid = WarpLidar(num_envs=1, config=cfg, mesh_ids_array=None, device="cpu")
        out["lidar_%s_rays" % mod] = lid.ray_vectors.numpy()
        out["lidar_%s_params" % mod] = np.array([cfg.height, cfg.width, cfg.horizontal_fov_deg_min, cfg.horizontal_fov_deg_max,
                                                  cfg.vertical_fov_deg_min, cfg.vertical_fov_deg_max], np.float64)
    return out


def camera_matrices():
    install_warp_stub()
    WarpCam = ref_shells.ref("sensors.warp.warp_cam").WarpCam
    out = {}
    for mod, cls in (("base_depth_camera_config", "BaseDepthCameraConfig"), ("stereo_camera_config", "StereoCameraConfig"),
                     ("d455_depth_config", "RsD455Config")):
        try:
            cfg = getattr(ref_shells.ref("config.sensor_config.camera_config." + mod), cls)
        except (ImportError, AttributeError) as e:
            print("skip", mod, e)
            continue
        cam = WarpCam(num_envs=1, config=cfg, mesh_ids_array=None, device="cpu")
        out["camera_%s_K" % mod] = np.asarray(cam.K, np.float64)
        out["camera_%s_params" % mod] = np.array([cfg.width, cfg.height, cfg.horizontal_fov_deg, cam.c_x, cam.c_y], np.float64)
    return out


def postprocess_cases():
    """WarpSensor.apply_noise / apply_range_limits / normalize_observation called as plain functions on a stand-in
    `self` (they only touch self.pixels and self.cfg).  torch.normal / torch.bernoulli are replaced by versions that
    draw through torch.randn / torch.rand and RECORD the draws (normal = randn * std + mean, the two roundings
    torch's own kernel makes; bernoulli = rand < p)."""
    install_warp_stub()
    WS = ref_shells.ref("sensors.warp.warp_sensor").WarpSensor
    base = ref_shells.ref("config.sensor_config.camera_config.base_depth_camera_config").BaseDepthCameraConfig
    out = {}
    g = torch.Generator().manual_seed(77)

    def run(tag, pixels, **over):
        class cfg(base):
            pass

        class noise(base.sensor_noise):
            pass

        for k, v in over.items():
            if k.startswith("noise_"):
                setattr(noise, k[6:], v)
            else:
                setattr(cfg, k, v)
        cfg.sensor_noise = noise
        me = types.SimpleNamespace(pixels=pixels.clone(), cfg=cfg)
        rec = {}
        real_normal, real_bern = torch.normal, torch.bernoulli

        def normal(mean, std):
            z = torch.randn(mean.shape, generator=g)
            rec["z"] = z
            return z * std + mean

        def bernoulli(p):
            u = torch.rand(p.shape, generator=g)
            rec["u"] = u
            return (u < p).to(p.dtype)

        torch.normal, torch.bernoulli = normal, bernoulli
        try:
            WS.apply_noise(me)
            if cfg.sensor_type in ["camera", "lidar", "stereo_camera"]:
                WS.apply_range_limits(me)
                WS.normalize_observation(me)
        finally:
            torch.normal, torch.bernoulli = real_normal, real_bern
        out["pp_%s_in" % tag] = pixels.numpy()
        out["pp_%s_out" % tag] = me.pixels.numpy()
        out["pp_%s_cfg" % tag] = np.array([cfg.min_range, cfg.max_range, cfg.far_out_of_range_value, cfg.near_out_of_range_value,
                                           float(cfg.normalize_range), float(cfg.return_pointcloud), float(cfg.pointcloud_in_world_frame),
                                           float(noise.enable_sensor_noise), getattr(noise, "std_a", 0.0), getattr(noise, "std_b", 0.0),
                                           getattr(noise, "std_c", 0.0), getattr(noise, "mean_offset", 0.0),
                                           noise.pixel_dropout_prob], np.float64)  # the camera configs define no std_* at all
        if "z" in rec:
            out["pp_%s_z" % tag], out["pp_%s_u" % tag] = rec["z"].numpy(), rec["u"].numpy()

    img = torch.rand(3, 1, 48, 64, generator=g) * 14.0  # beyond max_range (10) and below min_range (0.2) included
    img[0, 0, 0, :8] = torch.tensor([0.0, 0.1, 0.2, 0.2000001, 9.9999, 10.0, 10.000001, 1000.0])
    run("depth_plain", img)
    run("depth_unnormalised", img, normalize_range=False)
    run("depth_noise", img, noise_enable_sensor_noise=True, noise_pixel_dropout_prob=0.05,
        noise_std_a=0.002, noise_std_b=0.01, noise_std_c=0.003, noise_mean_offset=0.02)
    pts = (torch.rand(2, 1, 16, 24, 3, generator=g) - 0.5) * 16.0
    pts[0, 0, 0, 0] = torch.tensor([1000.0, 1000.0, 1000.0])
    pts[0, 0, 0, 1] = torch.tensor([0.05, 0.05, 0.05])
    run("points_sensor_frame", pts, return_pointcloud=True, pointcloud_in_world_frame=False)
    run("points_world_frame", pts, return_pointcloud=True, pointcloud_in_world_frame=True)
    run("points_noise", pts, return_pointcloud=True, pointcloud_in_world_frame=False, noise_enable_sensor_noise=True,
        noise_pixel_dropout_prob=0.1, noise_std_a=0.001, noise_std_b=0.02, noise_std_c=0.004, noise_mean_offset=0.0)
    return out


def pose_cases():
    """A REAL WarpSensor object (its Warp calls land on the inert stand-in): init_tensors -> reset() (mount
    randomisation, warp_sensor.py:139-161, draws replayed from the seeded global generator) -> update() (pose composition,
    :177-187).  Rows a20 (sensor mount) and a22."""
    install_warp_stub()
    ws_mod = ref_shells.ref("sensors.warp.warp_sensor")
    WS = ws_mod.WarpSensor
    real_lidar = ref_shells.ref("sensors.warp.warp_lidar").WarpLidar
    ws_mod.WarpLidar = lambda **kw: real_lidar(device="cpu", **kw)  # WarpSensor relies on the "cuda:0" default
    out = {}
    g = torch.Generator().manual_seed(123)
    for tag, mod, cls in (("camera", "camera_config.base_depth_camera_config", "BaseDepthCameraConfig"),
                          ("lidar", "lidar_config.osdome_64_config", "OSDome_64_Config")):
        cfg = getattr(ref_shells.ref("config.sensor_config." + mod), cls)
        n = 48
        q = torch.randn(n, 4, generator=g)
        q = q / q.norm(dim=1, keepdim=True)
        gtd = {"robot_position": (torch.rand(n, 3, generator=g) - 0.5) * 12.0, "robot_orientation": q,
               "gravity": torch.tensor([0.0, 0.0, -9.81]), "dt": 0.01, "robot_mass": torch.ones(n),
               "depth_range_pixels": torch.zeros(n, cfg.num_sensors, cfg.height, cfg.width),
               "segmentation_pixels": torch.zeros(n, cfg.num_sensors, cfg.height, cfg.width, dtype=torch.int32)}
        # torch_rand_float_tensor is TorchScript (aten::rand_like on the global generator, not patchable from
        # python): seed, run, then replay the same two draws (translation, rotation: [n, ns, 3] each)
        sensor = WS(sensor_config=cfg, num_envs=n, mesh_id_list=[0] * n, device="cpu")
        torch.manual_seed(4242)
        sensor.init_tensors(gtd)  # ends with reset(): all envs
        sensor.update()
        draws = []
        if cfg.randomize_placement:
            torch.manual_seed(4242)
            draws = [torch.rand(n, cfg.num_sensors, 3), torch.rand(n, cfg.num_sensors, 3)]
        out["pose_%s_robot_position" % tag] = gtd["robot_position"].numpy()
        out["pose_%s_robot_orientation" % tag] = gtd["robot_orientation"].numpy()
        out["pose_%s_cfg" % tag] = np.array(list(cfg.min_translation) + list(cfg.max_translation) + list(cfg.min_euler_rotation_deg)
                                            + list(cfg.max_euler_rotation_deg) + list(cfg.euler_frame_rot_deg)
                                            + [float(cfg.randomize_placement), cfg.num_sensors], np.float64)
        for i, u in enumerate(draws):
            out["pose_%s_u%d" % (tag, i)] = u.numpy()
        out["pose_%s_local_position" % tag] = sensor.sensor_local_position.numpy().copy()
        out["pose_%s_local_orientation" % tag] = sensor.sensor_local_orientation.numpy().copy()
        out["pose_%s_frame_quat" % tag] = sensor.sensor_data_frame_quat[0, 0].numpy().copy()
        out["pose_%s_sensor_position" % tag] = sensor.sensor_position.numpy().copy()
        out["pose_%s_sensor_orientation" % tag] = sensor.sensor_orientation.numpy().copy()
        print("pose", tag, "draws", len(draws), "randomize", cfg.randomize_placement)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    out = {}
    out.update(lidar_tables())
    out.update(camera_matrices())
    out.update(postprocess_cases())
    out.update(pose_cases())
    np.savez_compressed(os.path.join(OUT, "sensor_frontend.npz"), **out)
    print("sensor_frontend: ok ", len(out), "arrays:", ", ".join(sorted(k for k in out if k.endswith("_params") or k.endswith("_cfg"))))


if __name__ == "__main__":
    main()
