"""Golden vectors for the reference's ray-cast kernels, produced by EXECUTING THE REFERENCE'S OWN
SOURCE: the unmodified kernel bodies of

    sensors/warp/warp_kernels/warp_camera_kernels.py:13-282          DepthCameraWarpKernels     (5 kernels)
    sensors/warp/warp_kernels/warp_lidar_kernels.py:13-194           LidarWarpKernels           (5 kernels)
    sensors/warp/warp_kernels/warp_stereo_camera_kernels.py:13-299   StereoCameraWarpKernels    (4 kernels)

launched by the reference's own classes (WarpSensor -> WarpCam / WarpLidar / WarpStereoCam / WarpNormalFaceIDCam /
WarpNormalFaceIDLidar: intrinsics, ray table, graph capture, pose composition, noise / range limits / normalisation) under
the `warp` emulation of oracle/wp_emul.py (read its header for what is emulated and how: builtins restated from Warp's
headers as single binary32 operations, mesh_query_ray answered by that module's own brute-force loop -- the C oracle takes no part
in producing these fixtures, it is checked against them).  The only edits
to the reference at run time: its sensor classes are constructed with device="cpu" (their default is "cuda:0").

    python oracle/gen_golden_warp_kernels.py        (in the build container: needs /root/reference)

writes tests/golden/warp_kernels_{camera,lidar,stereo}.npz.  Scenes: 3 envs of indexed box meshes (8 vertices / 12 faces
per box, as trimesh boxes are: the segmentation lookup mesh.velocities[mesh.indices[3 f]][0] is exercised for real),
world vertices = the reference's tf_apply of random asset poses (what WarpEnv.reset_idx does, warp_env_manager.py:40-54).
  env 0: the robot inside a large room (every ray hits; the room's corners lie beyond the far plane) with boxes in it;
  env 1: the sensor INSIDE an obstacle (hits from the inside below min_range) next to others;
  env 2: open space, few boxes, one of them beyond the far plane: misses, silhouettes, beyond-far.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
import wp_emul  # noqa: E402

wp_emul.install()  # `import warp` now finds the emulation -- before any reference module is imported

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_shells  # noqa: E402

OUT = os.path.normpath(os.path.join(_HERE, "..", "tests", "golden"))

# (mesh_query_ray is answered by wp_emul's own brute-force loop: nothing of oracle/*.c takes part in producing these fixtures)

# a box as a closed indexed mesh: 8 corners, 12 outward-facing triangles
BOX_V = np.array([[x, y, z] for x in (-0.5, 0.5) for y in (-0.5, 0.5) for z in (-0.5, 0.5)], np.float32)
BOX_F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int32)


def build_scene(seed=20260926):
    """-> per env: (local vertices [V, 3], faces [F, 3], vertex -> asset index, asset poses [K, 7], per-asset seg id)"""
    rng = np.random.default_rng(seed)
    envs = []
    seg_counter = 100  # env_manager.py:147: the reference's segmentation counter starts here
    K = 12
    for e in range(3):
        size, pos, quat = [], [], []
        for k in range(K):
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            if e == 0 and k == 0:    # the room: 16 x 16 x 9 m around the origin, axis aligned
                size.append([16.0, 16.0, 9.0]); pos.append([0.0, 0.0, 2.0]); quat.append([0.0, 0.0, 0.0, 1.0])
            elif e == 1 and k == 0:  # the obstacle the sensor of env 1 sits in
                size.append([0.5, 0.45, 0.4]); pos.append([1.0, -0.5, 1.2]); quat.append(list(q))
            elif e == 2 and k >= 5:  # env 2 is sparse: the rest of its boxes lie beyond the far plane
                size.append(list(rng.uniform(0.5, 2.0, 3))); pos.append([14.0 + 2.0 * k, rng.uniform(-3, 3), rng.uniform(0, 3)]); quat.append(list(q))
            else:
                size.append(list(rng.uniform(0.3, 2.2, 3)))
                pos.append([rng.uniform(-3.5, 3.5), rng.uniform(-3.5, 3.5), rng.uniform(0.2, 3.0)])
                quat.append(list(q))
        size, pos, quat = np.array(size, np.float32), np.array(pos, np.float32), np.array(quat, np.float32)
        verts = np.concatenate([BOX_V * size[k] for k in range(K)]).astype(np.float32)
        faces = np.concatenate([BOX_F + 8 * k for k in range(K)]).astype(np.int32)
        v2a = np.repeat(np.arange(K), 8)
        seg = seg_counter + np.arange(K)
        seg[K - 1] = 70001 + e  # a large id: survives the float32 `velocities` detour and the int32 cast
        seg_counter += K
        envs.append(dict(verts=verts, faces=faces, v2a=v2a, pose=np.concatenate([pos, quat], axis=1), seg=seg.astype(np.int32)))
    return envs


def make_meshes(envs):
    """what WarpEnv.prepare_for_simulation (warp_env_manager.py:131-166) and reset_idx (:40-54) do, per env"""
    import warp as wp

    tf_apply = ref_shells.ref("utils.math").tf_apply
    meshes, keep = [], []
    for env in envs:
        pose = torch.tensor(env["pose"])
        v2a = torch.tensor(env["v2a"], dtype=torch.long)
        updated = tf_apply(pose[v2a, 3:7], pose[v2a, 0:3], torch.tensor(env["verts"])).contiguous()
        faces = torch.tensor(env["faces"], dtype=torch.int32)
        vel = torch.zeros(len(env["verts"]), 3)
        vel[:, 0] = torch.tensor(env["seg"][env["v2a"]], dtype=torch.float32)  # "we hijack this field and use it for segmentation"
        m = wp.Mesh(points=wp.from_torch(updated, dtype=wp.vec3), indices=wp.from_torch(faces.flatten(), dtype=wp.int32),
                    velocities=wp.from_torch(vel, dtype=wp.vec3))
        meshes.append(m)
        keep.append((updated, faces, vel))
    tri_world = np.stack([u.numpy()[f.numpy().astype(np.int64)].reshape(-1, 9) for u, f, _ in keep]).astype(np.float32)
    # the product's per-triangle segmentation table: int(velocities[indices[3 f]][0])  (camera kernels :60-62)
    tri_seg = np.stack([v.numpy()[f.numpy()[:, 0].astype(np.int64), 0].astype(np.int32) for _, f, v in keep])
    return meshes, keep, tri_world, tri_seg


def robot_states(seed=7):
    """tilted (up to ~15 deg of roll / pitch), yawed robots among the obstacles"""
    quat_from_euler_xyz = ref_shells.ref("utils.math").quat_from_euler_xyz
    e = torch.tensor([[0.21, -0.13, 0.6], [-0.17, 0.26, -2.1], [0.08, 0.19, 2.7]])
    q = quat_from_euler_xyz(e[:, 0], e[:, 1], e[:, 2])
    pos = torch.tensor([[0.3, -0.2, 1.5], [1.0, -0.5, 1.2], [-1.0, 0.5, 1.0]])
    return pos, q


def subclass(base, **over):
    noise_over = {k[6:]: v for k, v in over.items() if k.startswith("noise_")}
    body = {k: v for k, v in over.items() if not k.startswith("noise_")}

    class noise(base.sensor_noise):
        pass

    for k, v in noise_over.items():
        setattr(noise, k, v)
    body["sensor_noise"] = noise
    cfg = type("Cfg", (base,), body)
    # the config classes derive these three from the fields above at class-creation time (base_depth_camera_config.py:23-33)
    if "normalize_range" not in over:
        cfg.normalize_range = False if (cfg.return_pointcloud and cfg.pointcloud_in_world_frame) else base.normalize_range
    cfg.far_out_of_range_value = cfg.max_range if cfg.normalize_range else -1.0
    cfg.near_out_of_range_value = -cfg.max_range if cfg.normalize_range else -1.0
    return cfg


def run_case(out, tag, cfg, meshes, seed, states=None, keep_sensor=None):
    """A REAL WarpSensor of the reference over the emulated meshes: init_tensors (-> reset: mount randomisation) -> update()."""
    import warp as wp

    ws_mod = ref_shells.ref("sensors.warp.warp_sensor")
    for name in ("WarpCam", "WarpStereoCam", "WarpLidar", "WarpNormalFaceIDCam", "WarpNormalFaceIDLidar"):
        real = getattr(ws_mod, "_real_" + name, None) or getattr(ws_mod, name)
        setattr(ws_mod, "_real_" + name, real)
        setattr(ws_mod, name, (lambda r: (lambda **kw: r(device="cpu", **kw)))(real))  # WarpSensor relies on the "cuda:0" default
    n, S, H, W = len(meshes), cfg.num_sensors, cfg.height, cfg.width
    vec = cfg.return_pointcloud
    pos, q = states if states is not None else robot_states()
    gtd = {"robot_position": pos.clone(), "robot_orientation": q.clone(), "gravity": torch.tensor([0.0, 0.0, -9.81]), "dt": 0.01,
           "robot_mass": torch.ones(n),
           "depth_range_pixels": torch.full((n, S, H, W, 3) if vec else (n, S, H, W), -7.0),
           "segmentation_pixels": torch.full((n, S, H, W), -7, dtype=torch.int32)}
    before = (len(wp_emul.LAUNCH_LOG), len(wp_emul.UNDEFINED_READS))
    sensor = ws_mod.WarpSensor(sensor_config=cfg, num_envs=n, mesh_id_list=[m.id for m in meshes], device="cpu")
    torch.manual_seed(seed)
    sensor.init_tensors(gtd)
    sensor.update()  # pose composition, capture (graph build + replay), apply_noise, apply_range_limits, normalize_observation
    final = sensor.pixels.numpy().copy()
    raw = sensor.sensor.capture().numpy().copy()  # the same frame again, without the post-processing
    seg = gtd["segmentation_pixels"].numpy().copy()
    launches = wp_emul.LAUNCH_LOG[before[0]:]
    undefined = wp_emul.UNDEFINED_READS[before[1]:]
    kern = sorted({k for k, _ in launches})
    assert len(kern) == 1 and len(launches) == 2, launches
    has_seg = bool(cfg.segmentation_camera) or cfg.sensor_type.startswith("normal_faceID")
    p = "%s_" % tag
    out[p + "kernel"] = np.array(kern[0])
    out[p + "sensor_position"] = sensor.sensor_position.numpy().copy()
    out[p + "sensor_orientation"] = sensor.sensor_orientation.numpy().copy()
    out[p + "raw"] = raw
    out[p + "final"] = final
    if has_seg:
        out[p + "seg"] = seg
        # threads that stored a variable their control path never assigned (warp_lidar_kernels.py:49-86 on a miss):
        # the value Warp stores there is whatever its generated code left in the register -- masked out of every comparison
        undef = np.zeros((n, S, H, W), bool)
        for _, t, _ in undefined:
            undef[t[0], t[1], t[2], t[3]] = True  # lidar tid = (env, sensor, scan line, point)
        out[p + "seg_undefined"] = undef
    inner = sensor.sensor
    if keep_sensor is not None:
        keep_sensor.append(sensor)
    if hasattr(inner, "K_inv"):
        Ki = inner.K_inv.m
        assert all(Ki[r, c] == 0 for r, c in ((0, 1), (0, 3), (1, 0), (1, 3), (2, 0), (2, 1), (2, 3))) and Ki[2, 2] == 1.0, Ki
        out[p + "kinv"] = np.array([Ki[0, 0], Ki[0, 2], Ki[1, 1], Ki[1, 2]], np.float32)
        out[p + "K"] = np.asarray(inner.K.m, np.float32)
        out[p + "cxy"] = np.array([inner.c_x, inner.c_y], np.int32)
    if hasattr(inner, "ray_vectors"):
        out[p + "ray_vectors"] = inner.ray_vectors.np.copy()
    num = lambda v: float(v) if isinstance(v, (bool, int, float)) else v  # noqa: E731
    out[p + "cfg"] = np.array([W, H, S, num(cfg.max_range), num(cfg.min_range), num(cfg.far_out_of_range_value), num(cfg.near_out_of_range_value),
                               num(cfg.normalize_range), num(cfg.return_pointcloud), num(cfg.pointcloud_in_world_frame),
                               num(getattr(cfg, "calculate_depth", False)), num(bool(getattr(cfg, "segmentation_camera", False))),
                               num(getattr(cfg, "normal_in_world_frame", False)), num(getattr(cfg, "baseline", 0.0)),
                               num(getattr(cfg, "horizontal_fov_deg", 0.0))], np.float64)
    hits = (seg >= 0).mean() if has_seg else float("nan")
    print("%-34s %-58s rewrites=%d hit-fraction=%.2f undefined-reads=%d" % (tag, kern[0], inner_rewrites(kern[0]), hits, len(undefined)))


def inner_rewrites(qualname):
    for mod, cls in (("warp_camera_kernels", "DepthCameraWarpKernels"), ("warp_lidar_kernels", "LidarWarpKernels"),
                     ("warp_stereo_camera_kernels", "StereoCameraWarpKernels")):
        c = getattr(ref_shells.ref("sensors.warp.warp_kernels." + mod), cls)
        if qualname.startswith(cls + "."):
            return getattr(c, qualname.split(".", 1)[1]).rewrites
    return -1


CFG_LAYOUT = ("width", "height", "num_sensors", "max_range", "min_range", "far_oor", "near_oor", "normalize", "return_pointcloud", "world_frame",
              "calculate_depth", "segmentation", "normal_world", "baseline", "hfov_deg")


def scene_arrays(envs, tri_world, tri_seg):
    return {"tri_world": tri_world, "tri_seg": tri_seg, "asset_pose": np.stack([e["pose"] for e in envs]),
            "asset_seg": np.stack([e["seg"] for e in envs]), "faces": envs[0]["faces"], "cfg_layout": np.array(CFG_LAYOUT),
            "robot_position": robot_states()[0].numpy(), "robot_orientation": robot_states()[1].numpy()}


# ---------------------------------------------------------------------------------------------------------------------
# warp_kernels_boxes.npz: obstacles as the reference's loader really hands them to Warp
# ---------------------------------------------------------------------------------------------------------------------
# Every env asset of the reference is a URDF <box> (resources/models/environment_assets/*), loaded by urdfpy 0.0.22 whose
# Box.meshes is trimesh.creation.box(extents) (assets/warp_asset.py:19-24 -> visual_trimesh_fk): corners ({0,1}^3 - 0.5) * extents
# with index 4 x + 2 y + z, and THIS face list (trimesh/creation.py box(): `faces = [1,3,0, 4,1,0, 0,3,2, 2,4,0, 1,7,3, 5,1,4,
# 5,7,1, 3,7,2, 6,4,2, 2,7,6, 6,5,4, 7,5,6]`).  The product's BVH builder recognises boxes of exactly this topology and ends the
# tree at an "object node" (csrc/agx_scene.hip box_frame / box_triangle_ok, csrc/agx_raycast.hip box_face_candidates): the
# three fixtures above use another face order on purpose, so this one is what pins that traversal to the reference's kernels.
#   env 0: the robot in a room, rotated boxes, two thin wall slabs at an angle;
#   env 1: boxes that share face planes (side by side, stacked, back to back), the principal ray along a common edge;
#   env 2: the sensor origin ON a face (in its plane: the watertight test's t comes out as +-0 for every ray, accepted or not by the
#          sign of a zero) -- rays it rejects go on through the inside of that box or away from it;
#   env 3: the sensor strictly inside a rotated box;
#   env 4: a box corner / the midpoint of a box edge placed on chosen rays of the zero-mount camera and LiDAR;
#   env 5: generic rotated boxes, some beyond the far plane;
#   env 6: the sensor 0.5 mm above a box's top face: the principal row runs parallel to a face at a distance < 1e-3.
# In envs 1, 2, 4, 6 the random filler boxes stand behind the robot (the LiDAR sees them, the designed sight lines stay free).
TM_BOX_F = np.array([[1, 3, 0], [4, 1, 0], [0, 3, 2], [2, 4, 0], [1, 7, 3], [5, 1, 4], [5, 7, 1], [3, 7, 2], [6, 4, 2], [2, 7, 6], [6, 5, 4], [7, 5, 6]], np.int32)
ZERO_MOUNT = dict(min_translation=[0.0, 0.0, 0.0], max_translation=[0.0, 0.0, 0.0], min_euler_rotation_deg=[0.0, 0.0, 0.0],
                  max_euler_rotation_deg=[0.0, 0.0, 0.0])
BOX_ENVS = 7
BOX_K = 12


def box_robot_states():
    quat_from_euler_xyz = ref_shells.ref("utils.math").quat_from_euler_xyz
    e = torch.tensor([[0.21, -0.13, 0.6], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [-0.17, 0.26, -2.1], [0.0, 0.0, 0.0], [0.08, 0.19, 2.7], [0.0, 0.0, 0.0]])
    q = quat_from_euler_xyz(e[:, 0], e[:, 1], e[:, 2])
    for i in (1, 2, 4, 6):
        q[i] = torch.tensor([0.0, 0.0, 0.0, 1.0])  # exactly the identity (quat_from_euler_xyz(0, 0, 0) is, too; stated)
    pos = torch.tensor([[0.3, -0.2, 1.5], [0.0, 0.0, 1.0], [0.0, 0.0, 1.0], [1.0, -0.5, 1.2], [0.0, 0.0, 1.25], [-1.0, 0.5, 1.0], [0.0, 0.0, 1.0]])
    return pos, q


def _rot(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def aimed_rays():
    """World-frame rays (origin, direction) of chosen pixels of the zero-mount sensors of env 4, as the reference's kernels compute
    them: the arithmetic of warp_camera_kernels.py:197-210 / warp_lidar_kernels.py:30-37 with the emulated built-ins, on the pose
    a real WarpSensor composes.  Geometry is then placed ON these rays (a box corner, the midpoint of a box edge)."""
    import warp as wp

    cam = ref_shells.ref("config.sensor_config.camera_config.base_depth_camera_config").BaseDepthCameraConfig
    lidar = ref_shells.ref("config.sensor_config.lidar_config.base_lidar_config").BaseLidarConfig
    placeholder = [dict(verts=BOX_V.copy(), faces=TM_BOX_F.copy(), v2a=np.zeros(8, np.int64), pose=np.array([[50.0, 0, 0, 0, 0, 0, 1]], np.float32),
                        seg=np.array([1], np.int32)) for _ in range(BOX_ENVS)]
    meshes, _, _, _ = make_meshes(placeholder)
    rays = {}
    for kind, cfg, pix in (("camera", subclass(cam, calculate_depth=True, segmentation_camera=True, width=16, height=12, **ZERO_MOUNT), [(8, 6), (3, 2), (12, 9)]),
                           ("lidar", subclass(lidar, segmentation_camera=True, width=33, height=9, noise_enable_sensor_noise=False, **ZERO_MOUNT),
                            [(16, 4), (20, 3), (12, 6)])):
        held, scratch = [], {}
        run_case(scratch, "probe", cfg, meshes, 1, states=box_robot_states(), keep_sensor=held)
        sensor = held[0]
        ro = wp.vec3(*[float(v) for v in sensor.sensor_position[4, 0]])
        q = wp.quat(*[float(v) for v in sensor.sensor_orientation[4, 0]])
        for x, y in pix:
            if kind == "camera":
                rd = wp.normalize(wp.quat_rotate(q, wp.transform_vector(sensor.sensor.K_inv, wp.vec3(float(x), float(y), 1.0))))
            else:
                rd = wp.normalize(wp.quat_rotate(q, wp.vec3(*[float(v) for v in scratch["probe_ray_vectors"][y, x]])))
            rays[(kind, x, y)] = (np.array([float(c) for c in ro.c]), np.array([float(c) for c in rd.c]))
    return rays


def build_box_scene(seed=20260930):
    """7 envs x 12 trimesh-order boxes; what each env is there for: the comment block above."""
    rng = np.random.default_rng(seed)
    rays = aimed_rays()
    pos_r, _ = box_robot_states()
    pos_r = pos_r.numpy().astype(np.float64)
    ident = [0.0, 0.0, 0.0, 1.0]

    def rq():
        q = rng.normal(size=4)
        return list(q / np.linalg.norm(q))

    def yaw(a):
        return [0.0, 0.0, float(np.sin(a / 2)), float(np.cos(a / 2))]

    def random_box(lo=0.3, hi=2.2, far=False, behind=False):
        if far:
            c = [14.0 + rng.uniform(0, 20), rng.uniform(-3, 3), rng.uniform(0, 3)]
        elif behind:
            c = [rng.uniform(-6.0, -2.5), rng.uniform(-3.5, 3.5), rng.uniform(0.2, 3.0)]
        else:
            c = [rng.uniform(-3.5, 3.5), rng.uniform(-3.5, 3.5), rng.uniform(0.2, 3.0)]
        return list(rng.uniform(lo, hi, 3)), c, rq()

    def anchored(point, q, size, corner):
        """centre of the box with orientation q and extents `size` whose local point corner * size / 2 lies at `point`"""
        return list(np.asarray(point) - _rot(q) @ (np.asarray(corner, np.float64) * np.asarray(size) * 0.5))

    envs, seg_counter = [], 100
    for e in range(BOX_ENVS):
        B = []  # (size, centre, quat)
        if e == 0:
            B.append(([16.0, 16.0, 9.0], [0.0, 0.0, 2.0], ident))
            B.append(([4.0, 0.05, 3.0], [2.5, 1.0, 1.5], yaw(0.4)))
            B.append(([0.04, 5.0, 2.5], [-2.0, -1.0, 1.25], yaw(-0.3)))
        elif e == 1:  # A | B side by side (coincident inner faces at y = 0, coplanar fronts at x = 2.5: the principal ray runs along their
            # common edge), C on top of both (its -z face in the plane of their +z faces), D behind A with its front face IN A's back face
            B.append(([1.0, 1.0, 2.0], [3.0, -0.5, 1.0], ident))
            B.append(([1.0, 1.0, 2.0], [3.0, 0.5, 1.0], ident))
            B.append(([1.0, 2.0, 1.0], [3.0, 0.0, 2.5], ident))
            B.append(([1.0, 1.0, 2.0], [4.0, -0.5, 1.0], ident))
            B.append(([0.05, 6.0, 4.0], [6.0, 0.0, 2.0], ident))
        elif e == 2:  # box D's -y face lies in the plane y = 0 through the origin (rays with d_y > 0 start on the surface and run inside,
            # d_y < 0 leave it, the principal column runs IN the face's plane)
            B.append(([3.5, 1.0, 1.5], [1.25, 0.5, 1.0], ident))
            B.append(([1.0, 1.0, 1.0], [2.5, -1.5, 1.0], yaw(0.5)))
            B.append(([0.05, 8.0, 4.0], [7.0, 0.0, 2.0], yaw(0.2)))
        elif e == 6:  # box E's top face is 0.5 mm below the origin's height; F's side face 0.5 mm beside the principal column's plane
            B.append(([3.0, 1.75, 0.9995], [2.5, -0.5, 0.49975], ident))
            B.append(([2.0, 1.0, 1.0], [3.0, 0.5005, 1.5], ident))
            B.append(([0.05, 8.0, 4.0], [7.0, 0.0, 2.0], yaw(0.2)))
        elif e == 3:
            B.append(([0.6, 0.5, 0.45], list(pos_r[3] + np.array([0.03, -0.02, 0.01])), rq()))
            B.append(([5.0, 0.06, 3.0], [2.0, 1.5, 1.5], yaw(2.0)))
        elif e == 4:  # (to within the rounding of the placement: the edge functions of the watertight test come out as exact zeros or
            # as last-bit values)
            corners = [(1, 1, 1), (-1, 1, -1), (1, -1, 1), (-1, -1, 1), (1, 1, -1), (-1, 1, 1)]
            edges = [(1, 1, 0), (0, -1, 1), (-1, 0, -1), (1, 0, 1), (0, 1, -1), (-1, -1, 0)]
            for i, key in enumerate(sorted(rays)):
                ro, rd = rays[key]
                q = ident if i % 3 == 0 else rq()
                size = list(rng.uniform(0.4, 1.2, 3))
                local = corners[i] if i % 2 == 0 else edges[i]
                B.append((size, anchored(ro + (2.0 + 0.5 * i) * rd, q, size, local), q))
            B.append(([0.05, 9.0, 5.0], [8.0, 0.0, 2.0], ident))
        while len(B) < BOX_K:
            B.append(random_box(far=(e == 5 and len(B) >= 7), behind=e in (1, 2, 4, 6)))
        size = np.array([b[0] for b in B], np.float32)
        pos = np.array([b[1] for b in B], np.float32)
        quat = np.array([b[2] for b in B], np.float32)
        verts = np.concatenate([BOX_V * size[k] for k in range(BOX_K)]).astype(np.float32)
        faces = np.concatenate([TM_BOX_F + 8 * k for k in range(BOX_K)]).astype(np.int32)
        seg = seg_counter + np.arange(BOX_K)
        seg[BOX_K - 1] = 70001 + e
        seg_counter += BOX_K
        envs.append(dict(verts=verts, faces=faces, v2a=np.repeat(np.arange(BOX_K), 8), pose=np.concatenate([pos, quat], axis=1), seg=seg.astype(np.int32),
                         size=size))
    return envs


def main_boxes():
    os.makedirs(OUT, exist_ok=True)
    envs = build_box_scene()
    meshes, keep, tri_world, tri_seg = make_meshes(envs)
    cam = ref_shells.ref("config.sensor_config.camera_config.base_depth_camera_config").BaseDepthCameraConfig
    stereo = ref_shells.ref("config.sensor_config.camera_config.stereo_camera_config").StereoCameraConfig
    ncam = ref_shells.ref("config.sensor_config.camera_config.base_normal_faceID_camera_config").BaseNormalFaceIDCameraConfig
    lidar = ref_shells.ref("config.sensor_config.lidar_config.base_lidar_config").BaseLidarConfig
    nlidar = type("NormalFaceIDLidarCfg", (lidar,), dict(sensor_type="normal_faceID_lidar", return_pointcloud=True, normal_in_world_frame=True,
                                                         segmentation_camera=True))
    st = box_robot_states()
    out = scene_arrays(envs, tri_world, tri_seg)
    out["robot_position"], out["robot_orientation"] = st[0].numpy(), st[1].numpy()
    out["asset_size"] = np.stack([e["size"] for e in envs])
    kinds = {}

    def case(kind, tag, cfg, seed):
        run_case(out, tag, cfg, meshes, seed, states=st)
        kinds[tag] = kind

    c16, l33 = dict(width=16, height=12), dict(width=33, height=9, noise_enable_sensor_noise=False)
    case("camera", "cam_depth_seg_zero_mount", subclass(cam, calculate_depth=True, segmentation_camera=True, **c16, **ZERO_MOUNT), 51)
    case("camera", "cam_range_seg_32x24", subclass(cam, calculate_depth=False, segmentation_camera=True, width=32, height=24), 52)
    case("camera", "cam_points_world_seg_zero_mount", subclass(cam, return_pointcloud=True, pointcloud_in_world_frame=True, segmentation_camera=True, **c16, **ZERO_MOUNT), 53)
    case("camera", "cam_normal_world_zero_mount", subclass(ncam, normal_in_world_frame=True, **c16, **ZERO_MOUNT), 54)
    case("camera", "cam_depth_2sensors", subclass(cam, calculate_depth=True, segmentation_camera=False, num_sensors=2, **c16), 55)
    case("lidar", "lidar_range_seg_zero_mount", subclass(lidar, segmentation_camera=True, **l33, **ZERO_MOUNT), 61)
    case("lidar", "lidar_points_seg", subclass(lidar, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=True, width=32, height=8,
                                               noise_enable_sensor_noise=False), 62)
    case("lidar", "lidar_normal_world_zero_mount", subclass(nlidar, normal_in_world_frame=True, **l33, **ZERO_MOUNT), 63)
    case("stereo", "stereo_depth_seg_zero_mount", subclass(stereo, calculate_depth=True, segmentation_camera=True, **c16, **ZERO_MOUNT), 71)
    case("stereo", "stereo_range_wide_baseline", subclass(stereo, calculate_depth=False, segmentation_camera=False, baseline=0.6, **c16), 72)
    case("stereo", "stereo_points_seg_zero_mount", subclass(stereo, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=True, **c16, **ZERO_MOUNT), 73)
    for tag, kind in kinds.items():
        out[tag + "_kind"] = np.array(kind)
    np.savez_compressed(os.path.join(OUT, "warp_kernels_boxes.npz"), **out)
    print("warp_kernels_boxes.npz written to", OUT)


def main():
    os.makedirs(OUT, exist_ok=True)
    envs = build_scene()
    meshes, keep, tri_world, tri_seg = make_meshes(envs)
    cam = ref_shells.ref("config.sensor_config.camera_config.base_depth_camera_config").BaseDepthCameraConfig
    stereo = ref_shells.ref("config.sensor_config.camera_config.stereo_camera_config").StereoCameraConfig
    ncam = ref_shells.ref("config.sensor_config.camera_config.base_normal_faceID_camera_config").BaseNormalFaceIDCameraConfig
    lidar = ref_shells.ref("config.sensor_config.lidar_config.base_lidar_config").BaseLidarConfig
    # the reference ships no config class for sensor_type "normal_faceID_lidar" (warp_sensor.py:63-70 accepts it): the base
    # LiDAR config with the three fields WarpNormalFaceIDLidar reads on top
    nlidar = type("NormalFaceIDLidarCfg", (lidar,), dict(sensor_type="normal_faceID_lidar", return_pointcloud=True, normal_in_world_frame=True,
                                                         segmentation_camera=True))

    small = dict(width=16, height=12)
    out = scene_arrays(envs, tri_world, tri_seg)
    run_case(out, "depth_seg", subclass(cam, calculate_depth=True, segmentation_camera=True, **small), meshes, 11)
    run_case(out, "range_seg", subclass(cam, calculate_depth=False, segmentation_camera=True, **small), meshes, 12)
    run_case(out, "depth", subclass(cam, calculate_depth=True, segmentation_camera=False, **small), meshes, 13)
    run_case(out, "range_2sensors", subclass(cam, calculate_depth=False, segmentation_camera=False, num_sensors=2, **small), meshes, 14)
    run_case(out, "depth_seg_unnormalised_64x48", subclass(cam, calculate_depth=True, segmentation_camera=True, normalize_range=False, width=64, height=48), meshes, 15)
    run_case(out, "points_seg", subclass(cam, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=True, **small), meshes, 16)
    run_case(out, "points_world_seg", subclass(cam, return_pointcloud=True, pointcloud_in_world_frame=True, segmentation_camera=True, **small), meshes, 17)
    run_case(out, "points", subclass(cam, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=False, **small), meshes, 18)
    run_case(out, "points_world", subclass(cam, return_pointcloud=True, pointcloud_in_world_frame=True, segmentation_camera=False, **small), meshes, 19)
    run_case(out, "normal_world", subclass(ncam, normal_in_world_frame=True, **small), meshes, 20)
    run_case(out, "normal_camera_frame", subclass(ncam, normal_in_world_frame=False, **small), meshes, 21)
    np.savez_compressed(os.path.join(OUT, "warp_kernels_camera.npz"), **out)

    small = dict(width=32, height=8, noise_enable_sensor_noise=False)  # (the noise model is pinned by sensor_frontend.npz)
    out = scene_arrays(envs, tri_world, tri_seg)
    run_case(out, "range_seg", subclass(lidar, segmentation_camera=True, **small), meshes, 31)
    run_case(out, "range", subclass(lidar, segmentation_camera=False, **small), meshes, 32)
    run_case(out, "points_seg", subclass(lidar, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=True, **small), meshes, 33)
    run_case(out, "points_world_seg", subclass(lidar, return_pointcloud=True, pointcloud_in_world_frame=True, segmentation_camera=True, **small), meshes, 34)
    run_case(out, "points", subclass(lidar, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=False, **small), meshes, 35)
    run_case(out, "points_world", subclass(lidar, return_pointcloud=True, pointcloud_in_world_frame=True, segmentation_camera=False, **small), meshes, 36)
    run_case(out, "range_seg_dome_2sensors", subclass(lidar, segmentation_camera=True, num_sensors=2, vertical_fov_deg_min=0, vertical_fov_deg_max=90,
                                                     euler_frame_rot_deg=[0.0, 0.0, 0.0], width=24, height=6, noise_enable_sensor_noise=False), meshes, 37)
    run_case(out, "normal_world", subclass(nlidar, normal_in_world_frame=True, **small), meshes, 38)
    run_case(out, "normal_sensor_frame", subclass(nlidar, normal_in_world_frame=False, **small), meshes, 39)
    np.savez_compressed(os.path.join(OUT, "warp_kernels_lidar.npz"), **out)

    small = dict(width=16, height=12)
    out = scene_arrays(envs, tri_world, tri_seg)
    run_case(out, "depth_seg", subclass(stereo, calculate_depth=True, segmentation_camera=True, **small), meshes, 41)
    run_case(out, "range_seg", subclass(stereo, calculate_depth=False, segmentation_camera=True, **small), meshes, 42)
    run_case(out, "depth", subclass(stereo, calculate_depth=True, segmentation_camera=False, **small), meshes, 43)
    run_case(out, "range_wide_baseline", subclass(stereo, calculate_depth=False, segmentation_camera=False, baseline=0.6, **small), meshes, 44)
    run_case(out, "points_seg", subclass(stereo, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=True, **small), meshes, 45)
    run_case(out, "points_world_seg", subclass(stereo, return_pointcloud=True, pointcloud_in_world_frame=True, segmentation_camera=True, **small), meshes, 46)
    run_case(out, "points", subclass(stereo, return_pointcloud=True, pointcloud_in_world_frame=False, segmentation_camera=False, **small), meshes, 47)
    run_case(out, "points_world_wide_baseline", subclass(stereo, return_pointcloud=True, pointcloud_in_world_frame=True, segmentation_camera=False,
                                                        baseline=0.6, **small), meshes, 48)
    np.savez_compressed(os.path.join(OUT, "warp_kernels_stereo.npz"), **out)
    print("warp_kernels_{camera,lidar,stereo}.npz written to", OUT)


if __name__ == "__main__":
    if "--boxes-only" not in sys.argv:
        main()
    main_boxes()
