"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy front-end of the CPU oracle (oracle/*.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does (a test enforces
that no file under ``aerial_gym_simulator_amd/`` mentions ``oracle``).

All arrays are numpy float32/int32/uint8, C-contiguous, in the REFERENCE's AoS
layout ([N, C] rows, quaternions xyzw).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
MAX_MOTORS = 8

CTRL = {
    "no_control": 0,
    "position": 1,
    "velocity": 2,
    "attitude": 3,
    "rates": 4,
    "acceleration": 5,
    "velocity_steering": 6,
    "fully_actuated": 7,
}


class OrcRobotParams(C.Structure):
    _fields_ = [
        ("num_motors", C.c_int32),
        ("num_actions", C.c_int32),
        ("controller", C.c_int32),
        ("root_link_mode", C.c_int32),
        ("dt", C.c_float),
        ("dt_over_6", C.c_float),
        ("gravity", C.c_float * 3),
        ("mass", C.c_float),
        ("inertia", C.c_float * 9),
        ("inertia_inv", C.c_float * 9),
        ("alloc", C.c_float * (6 * MAX_MOTORS)),
        ("alloc_pinv", C.c_float * (MAX_MOTORS * 6)),
        ("wrench_map", C.c_float * (6 * MAX_MOTORS)),
        ("motor_dir", C.c_float * MAX_MOTORS),
        ("cq", C.c_float),
        ("use_rps", C.c_int32),
        ("use_discrete_approximation", C.c_int32),
        ("integration_rk4", C.c_int32),
        ("min_thrust", C.c_float),
        ("max_thrust", C.c_float),
        ("max_rate", C.c_float),
        ("max_yaw_rate", C.c_float),
        ("lin_drag_linear", C.c_float * 3),
        ("lin_drag_quadratic", C.c_float * 3),
        ("ang_drag_linear", C.c_float * 3),
        ("ang_drag_quadratic", C.c_float * 3),
        ("linear_damping", C.c_float),
        ("angular_damping", C.c_float),
        ("max_linear_velocity", C.c_float),
        ("max_angular_velocity", C.c_float),
        ("collision_radius", C.c_float),
    ]


def build(force=False):
    """Compile oracle/*.c with gcc (oracle/Makefile).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle_dynamics.c", "oracle_raycast.c", "oracle_types.h", "oracle_math.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def make_params(d):
    """Build an OrcRobotParams from a plain dict (see robot_model.params_dict in the tests)."""
    P = OrcRobotParams()
    M = int(d["num_motors"])
    P.num_motors = M
    P.num_actions = int(d["num_actions"])
    ctrl = d["controller"]
    P.controller = CTRL[ctrl] if isinstance(ctrl, str) else int(ctrl)
    P.root_link_mode = int(d.get("root_link_mode", 0))
    P.dt = float(d["dt"])
    P.dt_over_6 = float(d["dt"]) / 6.0  # python double arithmetic, rounded once by ctypes (motor_model.py:198)
    P.gravity[:] = [float(x) for x in d["gravity"]]
    P.mass = float(d["mass"])
    J = np.asarray(d["inertia"], dtype=np.float32).reshape(9)
    P.inertia[:] = J.tolist()
    Ji = np.asarray(d.get("inertia_inv", np.linalg.inv(J.reshape(3, 3).astype(np.float64))), dtype=np.float32).reshape(9)
    P.inertia_inv[:] = Ji.tolist()
    A = np.asarray(d["alloc"], dtype=np.float32).reshape(6, M)
    Ap = np.asarray(d["alloc_pinv"], dtype=np.float32).reshape(M, 6)
    W = np.asarray(d.get("wrench_map", A), dtype=np.float32).reshape(6, M)
    for name, mat in (("alloc", A), ("alloc_pinv", Ap), ("wrench_map", W)):
        buf = getattr(P, name)
        flat = mat.reshape(-1)
        for i in range(flat.size):
            buf[i] = float(flat[i])
    md = np.asarray(d["motor_dir"], dtype=np.float32).reshape(M)
    for i in range(M):
        P.motor_dir[i] = float(md[i])
    P.cq = float(d["cq"])
    P.use_rps = int(d["use_rps"])
    P.use_discrete_approximation = int(d["use_discrete_approximation"])
    P.integration_rk4 = int(d.get("integration_rk4", 1))
    P.min_thrust = float(d["min_thrust"])
    P.max_thrust = float(d["max_thrust"])
    P.max_rate = float(d["max_rate"])
    P.max_yaw_rate = float(d.get("max_yaw_rate", np.pi / 3.0))
    for name in ("lin_drag_linear", "lin_drag_quadratic", "ang_drag_linear", "ang_drag_quadratic"):
        getattr(P, name)[:] = [float(x) for x in d.get(name, [0.0, 0.0, 0.0])]
    P.linear_damping = float(d.get("linear_damping", 0.0))
    P.angular_damping = float(d.get("angular_damping", 0.0))
    P.max_linear_velocity = float(d.get("max_linear_velocity", 100.0))
    P.max_angular_velocity = float(d.get("max_angular_velocity", 100.0))
    P.collision_radius = float(d.get("collision_radius", 0.0))
    return P


class SubstepOut:
    __slots__ = ("euler", "qveh", "vveh", "vbody", "wbody", "wrench_cmd", "body_wrench", "action_clipped")


def substep(P, state, action, thrust, kT, tau_inc, tau_dec, Kp, Kv, KR, Kw, disturb=None,
            disturb_max=None, integrate=True):
    """One physics sub-step in place on ``state`` [N,13] and ``thrust`` [N,M]."""
    n = state.shape[0]
    M, A = P.num_motors, P.num_actions
    assert state.dtype == np.float32 and state.shape == (n, 13)
    assert thrust.dtype == np.float32 and thrust.shape == (n, M)
    out = SubstepOut()
    out.euler = np.zeros((n, 3), np.float32)
    out.qveh = np.zeros((n, 4), np.float32)
    out.vveh = np.zeros((n, 3), np.float32)
    out.vbody = np.zeros((n, 3), np.float32)
    out.wbody = np.zeros((n, 3), np.float32)
    out.wrench_cmd = np.zeros((n, 6), np.float32)
    out.body_wrench = np.zeros((n, 6), np.float32)
    out.action_clipped = np.zeros((n, A), np.float32)
    action = _f(action).reshape(n, A)
    dm = _f(disturb_max) if disturb_max is not None else None
    ds = _f(disturb) if disturb is not None else None
    lib().orc_substep(
        C.byref(P), n, _p(state), _p(action), _p(out.action_clipped), _p(thrust), _p(_f(kT)),
        _p(_f(tau_inc)), _p(_f(tau_dec)), _p(_f(Kp)), _p(_f(Kv)), _p(_f(KR)), _p(_f(Kw)),
        _p(ds), _p(dm), _p(out.euler), _p(out.qveh), _p(out.vveh), _p(out.vbody), _p(out.wbody),
        _p(out.wrench_cmd), _p(out.body_wrench), int(bool(integrate)),
    )
    return out


def robot_step(P, state, action, thrust, kT, tau_inc, tau_dec, Kp, Kv, KR, Kw, num_bodies, body_of_motor, disturb=None, disturb_max=None):
    """BaseMultirotor.step(action) with its per-body outputs (orc_robot_step): returns (SubstepOut, force [N,B,3], torque [N,B,3]);
    ``thrust`` [N,M] is updated in place, ``state`` is not integrated."""
    n = state.shape[0]
    M, A = P.num_motors, P.num_actions
    out = SubstepOut()
    out.euler, out.qveh = np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32)
    out.vveh, out.vbody, out.wbody = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    out.wrench_cmd, out.body_wrench = np.zeros((n, 6), np.float32), None
    out.action_clipped = np.zeros((n, A), np.float32)
    force, torque = np.zeros((n, num_bodies, 3), np.float32), np.zeros((n, num_bodies, 3), np.float32)
    mask = np.ascontiguousarray(np.asarray(list(body_of_motor) + [0] * 8, np.int32)[:8])
    dm = _f(disturb_max) if disturb_max is not None else None
    ds = _f(disturb) if disturb is not None else None
    lib().orc_robot_step(
        C.byref(P), n, _p(_f(state)), _p(_f(action).reshape(n, A)), _p(out.action_clipped), _p(thrust), _p(_f(kT)), _p(_f(tau_inc)),
        _p(_f(tau_dec)), _p(_f(Kp)), _p(_f(Kv)), _p(_f(KR)), _p(_f(Kw)), _p(ds), _p(dm), _p(out.euler), _p(out.qveh), _p(out.vveh),
        _p(out.vbody), _p(out.wbody), _p(out.wrench_cmd), int(num_bodies), _p(mask), _p(force), _p(torque))
    return out, force, torque


def net_body_wrench(rot, pos, force, torque):
    """[N,6] net body-frame wrench about the COM of per-body wrenches given in each body's frame (orc_net_body_wrench)"""
    n, nb = force.shape[0], force.shape[1]
    out = np.zeros((n, 6), np.float32)
    lib().orc_net_body_wrench(n, nb, _p(_f(rot).reshape(nb, 9)), _p(_f(pos).reshape(nb, 3)), _p(_f(force)), _p(_f(torque)), _p(out))
    return out


def update_states(state):
    n = state.shape[0]
    euler = np.zeros((n, 3), np.float32)
    qveh = np.zeros((n, 4), np.float32)
    vveh = np.zeros((n, 3), np.float32)
    vbody = np.zeros((n, 3), np.float32)
    wbody = np.zeros((n, 3), np.float32)
    lib().orc_update_states(n, _p(_f(state)), _p(euler), _p(qveh), _p(vveh), _p(vbody), _p(wbody))
    return euler, qveh, vveh, vbody, wbody


def rotmat_to_quat(m):
    """pytorch3d.transforms.matrix_to_quaternion as restated in the oracle; m [n,3,3] -> xyzw [n,4]."""
    m = _f(np.asarray(m).reshape(-1, 9))
    q = np.zeros((m.shape[0], 4), np.float32)
    lib().orc_rotmat_to_quat(m.shape[0], _p(m), _p(q))
    return q


def integrate(P, state, body_wrench):
    lib().orc_integrate(C.byref(P), state.shape[0], _p(state), _p(_f(body_wrench)))


def collide_sphere_boxes(radius, state, boxes, crashes):
    n, nb = boxes.shape[0], boxes.shape[1]
    lib().orc_collide_sphere_boxes(n, nb, C.c_float(radius), _p(_f(state)), _p(_f(boxes)), _p(crashes))


def reward_position(state, qveh, wbody, target, crashes):
    n = state.shape[0]
    reward = np.zeros(n, np.float32)
    lib().orc_reward_position(n, _p(_f(state)), _p(_f(qveh)), _p(_f(wbody)), _p(_f(target)), _p(crashes), _p(reward))
    return reward


def obs_position(state, vbody, wbody, target):
    n = state.shape[0]
    obs = np.zeros((n, 13), np.float32)
    lib().orc_obs_position(n, _p(_f(state)), _p(_f(vbody)), _p(_f(wbody)), _p(_f(target)), _p(obs))
    return obs


def reward_navigation(state, qveh, target, action, prev_action, curriculum_progress, rp, pos_err,
                      prev_pos_err, crashes):
    n = state.shape[0]
    reward = np.zeros(n, np.float32)
    action = _f(action)
    lib().orc_reward_navigation(
        n, _p(_f(state)), _p(_f(qveh)), _p(_f(target)), _p(action), _p(_f(prev_action)),
        action.shape[1], C.c_float(curriculum_progress), _p(_f(rp)), _p(pos_err), _p(prev_pos_err),
        _p(crashes), _p(reward),
    )
    return reward


def reset_robot_state(mask, u01, min_state, max_state, bounds_min, bounds_max, state):
    lib().orc_reset_robot_state(
        state.shape[0], _p(mask), _p(_f(u01)), _p(_f(min_state)), _p(_f(max_state)),
        _p(_f(bounds_min)), _p(_f(bounds_max)), _p(state),
    )


def quat_from_euler(e):
    e = _f(e)
    q = np.zeros((e.shape[0], 4), np.float32)
    lib().orc_quat_from_euler(e.shape[0], _p(e), _p(q))
    return q


def quat_mul(a, b):
    a, b = _f(a), _f(b)
    o = np.zeros_like(a)
    lib().orc_quat_mul(a.shape[0], _p(a), _p(b), _p(o))
    return o


def tf_apply(q, t, v):
    q, t, v = _f(q), _f(t), _f(v)
    o = np.zeros_like(v)
    lib().orc_tf_apply(v.shape[0], _p(q), _p(t), _p(v), _p(o))
    return o


# ----------------------------------------------------------------------------- ray-cast
def scene_transform(tri_local, tri_asset, asset_state):
    n, nt = tri_local.shape[0], tri_local.shape[1]
    na = asset_state.shape[1]
    out = np.zeros((n, nt, 9), np.float32)
    ta = np.ascontiguousarray(tri_asset, dtype=np.int32)
    lib().orc_scene_transform(n, nt, na, _p(_f(tri_local)), _p(ta), _p(_f(asset_state)), _p(out))
    return out


def prims_from_assets(prim_asset, asset_state, local_pos, local_quat):
    n, na = asset_state.shape[0], asset_state.shape[1]
    pa = np.ascontiguousarray(prim_asset, dtype=np.int32)  # [N, P]
    out = np.zeros((n, pa.shape[1], 13), np.float32)
    lib().orc_prims_from_assets(n, pa.shape[1], na, _p(pa), _p(_f(asset_state)), _p(_f(local_pos)), _p(_f(local_quat)), _p(out))
    return out


def sensor_pose(state, local_pos, local_quat, frame_quat):
    n, ns = local_pos.shape[0], local_pos.shape[1]
    pos = np.zeros((n, ns, 3), np.float32)
    quat = np.zeros((n, ns, 4), np.float32)
    lib().orc_sensor_pose(n, ns, _p(_f(state)), _p(_f(local_pos)), _p(_f(local_quat)), _p(_f(frame_quat)), _p(pos), _p(quat))
    return pos, quat


def mesh_query_ray(o, d, max_t, tris):
    """(hit, t, face) of ONE ray against [T, 9] float32 triangles: brute-force closest hit (oracle_raycast.c closest_hit)"""
    t, f = C.c_float(0.0), C.c_int(-1)
    o3, d3 = (C.c_float * 3)(*[float(x) for x in o]), (C.c_float * 3)(*[float(x) for x in d])
    fn = lib().orc_mesh_query_ray
    fn.restype = C.c_int
    hit = fn(o3, d3, C.c_float(float(max_t)), _p(tris), int(tris.shape[0]), C.byref(t), C.byref(f))
    return bool(hit), np.float32(t.value), int(f.value)


def camera_kinv(width, height, hfov_deg):
    """{K_inv[0][0], K_inv[0][2], K_inv[1][1], K_inv[1][2]} of warp_cam.py:31-64: K is a float32 matrix (wp.mat44) and
    wp.inverse inverts it by cofactors with the determinant's reciprocal in double (warp mat.h / USD GfMatrix4f::Inverse);
    pinned by tests/golden/warp_kernels_camera.npz (the reference's WarpCam under oracle/wp_emul.py)."""
    import math

    W, H = width, height
    u0, v0 = W / 2, H / 2
    hfov = math.radians(hfov_deg)
    f = W / 2 * 1 / math.tan(hfov / 2)
    vfov = 2 * math.atan(H / (2 * f))
    K = np.eye(4, dtype=np.float32)
    K[0, 0], K[0, 2], K[1, 1], K[1, 2] = u0 / math.tan(hfov / 2), u0, v0 / math.tan(vfov / 2), v0
    det = np.float64(K[0, 0] * K[1, 1])  # float32 product: the only non-zero term of the cofactor expansion
    adj = np.array([np.float32(np.float64(K[1, 1])), np.float32(-(np.float64(K[1, 1]) * np.float64(K[0, 2]))),
                    np.float32(np.float64(K[0, 0])), np.float32(-(np.float64(K[0, 0]) * np.float64(K[1, 2])))], np.float64)
    return (adj * (np.float64(1.0) / det)).astype(np.float32), int(u0), int(v0)


MODE = {"range": 0, "depth": 1, "pointcloud": 2, "pointcloud_world": 3, "normal": 4, "normal_world": 5}


def raycast_camera(width, height, kinv, far_plane, c_x, c_y, mode, cam_pos, cam_quat, tris, tri_seg,
                   want_seg=True, use_bvh=False):
    n, ns = cam_pos.shape[0], cam_pos.shape[1]
    nt = tris.shape[1]
    m = MODE[mode] if isinstance(mode, str) else mode
    shape = (n, ns, height, width) if m <= 1 else (n, ns, height, width, 3)
    pixels = np.zeros(shape, np.float32)
    seg = np.zeros((n, ns, height, width), np.int32) if want_seg else None
    ts = np.ascontiguousarray(tri_seg, dtype=np.int32)
    lib().orc_raycast_camera(
        n, ns, width, height, _p(_f(kinv)), C.c_float(far_plane), c_x, c_y, m, _p(_f(cam_pos)),
        _p(_f(cam_quat)), _p(_f(tris)), _p(ts), nt, int(use_bvh), _p(pixels), _p(seg),
    )
    return pixels, seg


def raycast_stereo_camera(width, height, kinv, far_plane, baseline, c_x, c_y, mode, cam_pos, cam_quat, tris, tri_seg,
                          want_seg=True, use_bvh=False):
    """warp_stereo_camera_kernels.py: occlusion-checked depth / range / point cloud (mode 0..3)."""
    n, ns = cam_pos.shape[0], cam_pos.shape[1]
    nt = tris.shape[1]
    m = MODE[mode] if isinstance(mode, str) else mode
    shape = (n, ns, height, width) if m <= 1 else (n, ns, height, width, 3)
    pixels = np.zeros(shape, np.float32)
    seg = np.zeros((n, ns, height, width), np.int32) if want_seg else None
    ts = np.ascontiguousarray(tri_seg, dtype=np.int32)
    lib().orc_raycast_stereo_camera(
        n, ns, width, height, _p(_f(kinv)), C.c_float(far_plane), C.c_float(baseline), c_x, c_y, m, _p(_f(cam_pos)),
        _p(_f(cam_quat)), _p(_f(tris)), _p(ts), nt, int(use_bvh), _p(pixels), _p(seg),
    )
    return pixels, seg


def lidar_ray_table(height, width, hfov_min_deg, hfov_max_deg, vfov_min_deg, vfov_max_deg):
    """warp_lidar.py:40-64: az from +max to min over W, el from +max to min over H, normalised."""
    import math

    hmin, hmax = math.radians(hfov_min_deg), math.radians(hfov_max_deg)
    vmin, vmax = math.radians(vfov_min_deg), math.radians(vfov_max_deg)
    rv = np.zeros((height, width, 3), np.float32)
    for i in range(height):
        for j in range(width):
            az = hmax - (hmax - hmin) * (j / (width - 1))
            el = vmax - (vmax - vmin) * (i / (height - 1))
            rv[i, j, 0] = math.cos(az) * math.cos(el)
            rv[i, j, 1] = math.sin(az) * math.cos(el)
            rv[i, j, 2] = math.sin(el)
    nrm = np.sqrt((rv * rv).sum(axis=2, keepdims=True, dtype=np.float32)).astype(np.float32)
    return (rv / nrm).astype(np.float32)


def raycast_lidar(ray_vectors, far_plane, mode, pos, quat, tris, tri_seg, want_seg=True, use_bvh=False):
    n, ns = pos.shape[0], pos.shape[1]
    height, width = ray_vectors.shape[0], ray_vectors.shape[1]
    nt = tris.shape[1]
    m = MODE[mode] if isinstance(mode, str) else mode
    shape = (n, ns, height, width) if m == 0 else (n, ns, height, width, 3)
    pixels = np.zeros(shape, np.float32)
    seg = np.zeros((n, ns, height, width), np.int32) if want_seg else None
    ts = np.ascontiguousarray(tri_seg, dtype=np.int32)
    lib().orc_raycast_lidar(
        n, ns, width, height, _p(_f(ray_vectors)), C.c_float(far_plane), m, _p(_f(pos)), _p(_f(quat)),
        _p(_f(tris)), _p(ts), nt, int(use_bvh), _p(pixels), _p(seg),
    )
    return pixels, seg


def sensor_postprocess(pixels, min_range, max_range, far_oor, near_oor, normalize, z_normal=None,
                       u_dropout=None, std_a=0.0, std_b=0.0, std_c=0.0, mean_offset=0.0,
                       dropout_prob=0.0):
    assert pixels.dtype == np.float32 and pixels.flags["C_CONTIGUOUS"]
    zn = _f(z_normal) if z_normal is not None else None
    ud = _f(u_dropout) if u_dropout is not None else None
    lib().orc_sensor_postprocess(
        C.c_size_t(pixels.size), _p(pixels), _p(zn), _p(ud), C.c_float(std_a), C.c_float(std_b),
        C.c_float(std_c), C.c_float(mean_offset), C.c_float(dropout_prob), C.c_float(min_range),
        C.c_float(max_range), C.c_float(far_oor), C.c_float(near_oor), int(bool(normalize)),
    )
    return pixels


def sensor_postprocess_points(pixels, min_range, max_range, far_oor, near_oor, limits, normalize, z_normal=None,
                              u_dropout=None, std_a=0.0, std_b=0.0, std_c=0.0, mean_offset=0.0, dropout_prob=0.0):
    assert pixels.dtype == np.float32 and pixels.flags["C_CONTIGUOUS"] and pixels.shape[-1] == 3
    zn = _f(z_normal) if z_normal is not None else None
    ud = _f(u_dropout) if u_dropout is not None else None
    lib().orc_sensor_postprocess_points(
        C.c_size_t(pixels.size // 3), _p(pixels), _p(zn), _p(ud), C.c_float(std_a), C.c_float(std_b),
        C.c_float(std_c), C.c_float(mean_offset), C.c_float(dropout_prob), C.c_float(min_range),
        C.c_float(max_range), C.c_float(far_oor), C.c_float(near_oor), int(bool(limits)), int(bool(normalize)),
    )
    return pixels


# ----------------------------------------------------------------------------- f2: LiDAR navigation task
def reward_lidar_navigation(pos_err, vveh, wbody, yaw_error, crashes, action, prev_action, ttc, curriculum_progress, rp):
    n = pos_err.shape[0]
    reward = np.zeros(n, np.float32)
    cr = np.ascontiguousarray(crashes, dtype=np.uint8)
    lib().orc_reward_lidar_navigation(n, _p(_f(pos_err)), _p(_f(vveh)), _p(_f(wbody)), _p(_f(yaw_error)), _p(cr), _p(_f(action)),
                                      _p(_f(prev_action)), _p(_f(ttc)), C.c_float(curriculum_progress), _p(_f(rp)), _p(reward))
    return reward


def lidar_image_obs(pointcloud, robot_pos, robot_linvel, ph=3, pw=6, low_row0=10, noise_mask=None, noise_val=None, max_mask=None,
                    low_mask=None, low_val=None):
    """pointcloud [N,H,W,3] -> (time_to_collision [N], inverse min-pooled range [N, (H/ph)*(W/pw)]).
    low_mask / low_val: [N, H/ph, W/pw] (only rows >= low_row0 are read)."""
    n, H, W = pointcloud.shape[0], pointcloud.shape[1], pointcloud.shape[2]
    ttc = np.zeros(n, np.float32)
    ds = np.zeros((n, (H // ph) * (W // pw)), np.float32)
    opt = [(_f(a) if a is not None else None) for a in (noise_mask, noise_val, max_mask, low_mask, low_val)]
    lib().orc_lidar_image_obs(n, H, W, ph, pw, low_row0, _p(_f(pointcloud)), _p(_f(robot_pos)), _p(_f(robot_linvel)),
                              *[_p(a) for a in opt], _p(ttc), _p(ds))
    return ttc, ds


def obs_lidar_navigation(state, euler, qveh, vbody, wbody, actions, target, target_yaw, u_vec, u_euler, downsampled):
    n, cells = state.shape[0], downsampled.shape[1]
    obs = np.zeros((n, 17 + cells), np.float32)
    a = _f(actions)
    lib().orc_obs_lidar_navigation(n, _p(_f(state)), _p(_f(euler)), _p(_f(qveh)), _p(_f(vbody)), _p(_f(wbody)), _p(a), a.shape[1],
                                   _p(_f(target)), _p(_f(target_yaw)), _p(_f(u_vec)), _p(_f(u_euler)), _p(_f(downsampled)), cells,
                                   _p(obs))
    return obs


# ----------------------------------------------------------------------------- f4: kinematic obstacles
def assets_integrate(asset_state, twist, dt, k):
    """asset_state [N,K,13] updated in place from twist [N,K,6]"""
    assert asset_state.dtype == np.float32 and asset_state.flags["C_CONTIGUOUS"]
    lib().orc_assets_integrate(asset_state.shape[0] * asset_state.shape[1], _p(asset_state), _p(_f(twist)), C.c_float(dt), int(k))
    return asset_state


# ----------------------------------------------------------------------------- f4: IMU
def imu_update(mass, g_world, sqrt_dt, world_frame, enable_noise, enable_bias, bias_std, noise_std, max_value, force, quat, wbody,
               sensor_quat, z_noise, z_bias, bias):
    """one physics sub-step of IMUSensor.update; `bias` [N,6] is updated in place; returns imu_meas [N,6]"""
    n = force.shape[0]
    meas = np.zeros((n, 6), np.float32)
    assert bias.dtype == np.float32 and bias.flags["C_CONTIGUOUS"]
    lib().orc_imu_update(n, C.c_float(mass), _p(_f(g_world)), C.c_float(sqrt_dt), int(world_frame), int(enable_noise), int(enable_bias),
                         _p(_f(bias_std)), _p(_f(noise_std)), _p(_f(max_value)), _p(_f(force)), _p(_f(quat)), _p(_f(wbody)),
                         _p(_f(sensor_quat)), _p(_f(z_noise)), _p(_f(z_bias)), _p(bias), _p(meas))
    return meas


def imu_reset(mask, u_bias, u_rot, max_bias_init, min_rot, max_rot, bias, sensor_quat):
    n = bias.shape[0]
    mk = np.ascontiguousarray(mask, dtype=np.uint8)
    lib().orc_imu_reset(n, _p(mk), _p(_f(u_bias)), _p(_f(u_rot)), _p(_f(max_bias_init)), _p(_f(min_rot)), _p(_f(max_rot)), _p(bias),
                        _p(sensor_quat))


# ----------------------------------------------------------------------------- device-RNG restatement
RNG_BOUNDS, RNG_STATE, RNG_GAINS, RNG_MOTOR, RNG_ASSET_SEL, RNG_ASSETS = 0, 1, 2, 3, 4, 16
RNG_LIDAR_NOISE, RNG_OBS_NOISE, RNG_IMU_RESET, RNG_IMU, RNG_TARGET, RNG_SENSOR_MOUNT = 5, 6, 7, 8, 9, 1 << 16


def rng_fill(seed, episodes, stream, count):
    ep = np.ascontiguousarray(episodes, dtype=np.int32)
    out = np.zeros((ep.shape[0], count), np.float32)
    lib().orc_rng_fill(C.c_uint64(seed), ep.shape[0], _p(ep), int(stream), int(count), _p(out))
    return out


def reset_assets(mask, u, sel, min_ratio, max_ratio, bounds_min, bounds_max, num_obstacles, num_keep, asset_state):
    n, K = asset_state.shape[0], asset_state.shape[1]
    lib().orc_reset_assets(n, K, _p(np.ascontiguousarray(mask, np.uint8)), _p(_f(u)), _p(np.ascontiguousarray(sel, np.uint8)),
                           _p(_f(min_ratio)), _p(_f(max_ratio)), _p(_f(bounds_min)), _p(_f(bounds_max)), int(num_obstacles),
                           int(num_keep), _p(asset_state))


def obs_navigation(state, euler, qveh, vbody, wbody, actions, target, u_vec, u_euler, pixels, obs_dim, gh=8, gw=8):
    n = state.shape[0]
    obs = np.zeros((n, obs_dim), np.float32)
    actions = _f(actions)
    if pixels is not None:
        pixels = _f(pixels)
        ns, H, W = pixels.shape[1], pixels.shape[2], pixels.shape[3]
    else:
        ns = H = W = 0
    lib().orc_obs_navigation(n, _p(_f(state)), _p(_f(euler)), _p(_f(qveh)), _p(_f(vbody)), _p(_f(wbody)), _p(actions),
                             actions.shape[1], _p(_f(target)), _p(_f(u_vec)), _p(_f(u_euler)), _p(pixels), ns, H, W, gh, gw,
                             obs_dim, _p(obs))
    return obs


def math_eval(which, x, y=None):
    """oracle_math.h kernels: which in ("sin", "cos", "atan2", "asin", "exp"); atan2(x, y) takes (y-arg, x-arg)."""
    idx = ("sin", "cos", "atan2", "asin", "exp").index(which)
    x = _f(x).ravel()
    yy = _f(y).ravel() if y is not None else x
    out = np.zeros_like(x)
    lib().orc_math_eval(idx, x.shape[0], _p(x), _p(yy), _p(out))
    return out
