"""Rigid-composite model of a multirotor from its link table.

The reference obtains mass / inertia from Isaac Gym's per-link rigid-body properties and
composes them with the parallel-axis theorem (robots/robot_manager.py:295-435); forces
are applied in each motor LINK's frame (IGE_env_manager.py:444-449, LOCAL_SPACE), so the
URDF joint origins (xyz, rpy) define the thrust axes.  Here the same numbers come from
the ``robot_model`` table of the robot config.
"""
import math

import numpy as np
import torch

from .._lib import AgxRobotParams, CTRL_IDS


def rpy_to_matrix(roll, pitch, yaw):
    """URDF fixed-axis rpy -> rotation matrix R = Rz(yaw) Ry(pitch) Rx(roll)."""
    sr, cr = math.sin(roll), math.cos(roll)
    sp, cp = math.sin(pitch), math.cos(pitch)
    sy, cy = math.sin(yaw), math.cos(yaw)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ]
    )


def composite_body(model):
    """(mass, com[3], inertia[3,3]) of base link + point-mass motors, float64."""
    xyz = np.asarray(model.motor_xyz, dtype=np.float64)
    m_motor = float(model.motor_mass)
    mass = float(model.base_mass) + m_motor * len(xyz)
    com = (m_motor * xyz.sum(axis=0)) / mass
    inertia = np.asarray(model.base_inertia, dtype=np.float64).copy()
    # the motor links' own (diagonal) inertia, when the URDF gives one: rotated frames do not matter for k * I
    inertia += len(xyz) * float(getattr(model, "motor_inertia", 0.0)) * np.eye(3)
    # base link sits at the origin; shift it and every motor to the common COM
    for m, r in [(float(model.base_mass), np.zeros(3))] + [(m_motor, p) for p in xyz]:
        d = r - com
        inertia += m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return mass, com, inertia


def motor_wrench_map(model, motor_directions, cq, com):
    """6 x M: body-frame wrench (about the COM) produced by unit thrust of each motor.

    Motor i pushes along its link's z axis with reaction torque -cq * dir_i about the same
    axis (control_allocation.py:103-114)."""
    M = len(model.motor_xyz)
    W = np.zeros((6, M))
    for i in range(M):
        axis = rpy_to_matrix(*model.motor_rpy[i])[:, 2]
        arm = np.asarray(model.motor_xyz[i], dtype=np.float64) - com
        W[0:3, i] = axis
        W[3:6, i] = np.cross(arm, axis) - cq * motor_directions[i] * axis
    return W


def link_frames(robot_cfg, num_bodies):
    """(AgxLinkFrames, known[num_bodies]): pose of every rigid body of the robot in the root-link frame, about the composite's
    centre of mass -- what turns the per-body tensors robot_force_tensor / robot_torque_tensor (each body's wrench in ITS
    frame, IGE_env_manager.py:444-449 LOCAL_SPACE) into the net wrench on the rigid composite.  Known from the config's
    robot_model table: the root link (body 0) and the motor links (control_allocator_config.application_mask -> motor_xyz /
    motor_rpy); optional `robot_model.link_xyz` / `link_rpy` dicts {body index: pose} name further links.  A body without
    a known pose (base_quadrotor's four massless arm links) must stay at zero wrench: `known` says which."""
    from .._lib import MAX_BODIES, AgxLinkFrames

    if not 1 <= num_bodies <= MAX_BODIES:
        raise ValueError(f"robot with {num_bodies} bodies: the link-frame table holds {MAX_BODIES}")
    model, ca = robot_cfg.robot_model, robot_cfg.control_allocator_config
    _, com, _ = composite_body(model)
    L = AgxLinkFrames()
    L.num_bodies = num_bodies
    known = [False] * num_bodies
    poses = {0: ([0.0, 0.0, 0.0], [0.0, 0.0, 0.0])}
    for j, b in enumerate(ca.application_mask):
        if j < len(model.motor_xyz):
            poses[int(b)] = (model.motor_xyz[j], model.motor_rpy[j])
    for b, xyz in dict(getattr(model, "link_xyz", {}) or {}).items():
        poses[int(b)] = (xyz, dict(getattr(model, "link_rpy", {}) or {}).get(b, [0.0, 0.0, 0.0]))
    for b in range(num_bodies):
        xyz, rpy = poses.get(b, ([0.0, 0.0, 0.0], [0.0, 0.0, 0.0]))
        R = rpy_to_matrix(*rpy).astype(np.float32).reshape(-1)
        r = (np.asarray(xyz, np.float64) - com).astype(np.float32)
        for c in range(9):
            L.rot[b][c] = float(R[c])
        for c in range(3):
            L.pos[b][c] = float(r[c])
        known[b] = b in poses
    return L, known


def _fill(dst, values):
    values = np.asarray(values, dtype=np.float32).reshape(-1)
    for i, v in enumerate(values):
        dst[i] = float(v)


def allocation_pinv(A32):
    """Moore-Penrose inverse of the fp32 allocation matrix (control_allocation.py:27: torch.linalg.pinv), evaluated in
    float64 and rounded once.  The reference's fp32 SVD leaves machine-dependent noise in the result (entries that are
    exactly 0 come out as +-2e-4, the 1 / (4 * 0.13) torque gains differ by 7e-7 between two hosts), which would make
    every thrust depend on the BLAS of the box; from float64 the fp32 result is the correctly rounded one, and entries
    below 1e-12 of the largest are the zeros they stand for."""
    P = np.linalg.pinv(np.asarray(A32, dtype=np.float32).astype(np.float64))
    P[np.abs(P) < 1e-12 * np.abs(P).max()] = 0.0
    return P.astype(np.float32)


def robot_params_dict(robot_cfg, controller_cfg, controller_kind, sim_cfg):
    """Plain-number description of the robot (the parity tests consume the same dict)."""
    ca = robot_cfg.control_allocator_config
    mm = ca.motor_model_config
    model = robot_cfg.robot_model
    mass, com, inertia = composite_body(model)
    if np.abs(com).max() > 1e-9:
        raise NotImplementedError("robots whose COM is not at the base-link origin are not supported yet")
    M = int(ca.num_motors)
    A = np.asarray(ca.allocation_matrix, dtype=np.float32)
    if A.shape != (6, M):
        raise ValueError("Allocation matrix must have 6 rows and num_motors columns.")
    A_pinv = allocation_pinv(A)
    J32 = inertia.astype(np.float32)
    W = motor_wrench_map(model, ca.motor_directions, float(mm.thrust_to_torque_ratio), com)
    scheme = getattr(mm, "integration_scheme", "rk4")
    num_actions = M if controller_kind == "none" else int(controller_cfg.num_actions)
    return dict(
        num_motors=M,
        num_actions=num_actions,
        controller=controller_kind,
        root_link_mode=int(ca.force_application_level != "motor_link"),  # control_allocation.py:53-65: anything else = root wrench
        dt=float(sim_cfg.sim.dt),
        gravity=[float(g) for g in sim_cfg.sim.gravity],
        mass=float(np.float32(mass)),
        inertia=J32.reshape(-1).tolist(),
        inertia_inv=np.linalg.inv(J32.astype(np.float64)).astype(np.float32).reshape(-1).tolist(),
        alloc=A.reshape(-1).tolist(),
        alloc_pinv=A_pinv.reshape(-1).tolist(),
        wrench_map=W.astype(np.float32).reshape(-1).tolist(),
        motor_dir=[float(d) for d in ca.motor_directions],
        cq=float(mm.thrust_to_torque_ratio),
        use_rps=int(bool(mm.use_rps)),
        use_discrete_approximation=int(bool(mm.use_discrete_approximation)),
        integration_rk4=int(scheme != "euler"),
        min_thrust=float(mm.min_thrust),
        max_thrust=float(mm.max_thrust),
        max_rate=float(mm.max_thrust_rate),
        max_yaw_rate=float(getattr(controller_cfg, "max_yaw_rate", math.pi / 3.0)),
        lin_drag_linear=list(robot_cfg.damping.linvel_linear_damping_coefficient),
        lin_drag_quadratic=list(robot_cfg.damping.linvel_quadratic_damping_coefficient),
        ang_drag_linear=list(robot_cfg.damping.angular_linear_damping_coefficient),
        ang_drag_quadratic=list(robot_cfg.damping.angular_quadratic_damping_coefficient),
        linear_damping=float(robot_cfg.robot_asset.linear_damping),
        angular_damping=float(robot_cfg.robot_asset.angular_damping),
        max_linear_velocity=float(robot_cfg.robot_asset.max_linear_velocity),
        max_angular_velocity=float(robot_cfg.robot_asset.max_angular_velocity),
        collision_radius=float(model.collision_sphere_radius),
    )


def pack_robot_params(d):
    """dict (robot_params_dict) -> the C struct of include/aerial_gym_hip.h."""
    P = AgxRobotParams()
    M = P.num_motors = int(d["num_motors"])
    P.num_actions = int(d["num_actions"])
    P.controller = CTRL_IDS[d["controller"]]
    P.root_link_mode = int(d["root_link_mode"])
    P.dt = d["dt"]
    P.dt_over_6 = float(d["dt"]) / 6.0  # double arithmetic like the reference's python scalars, rounded once (motor_model.py:198)
    _fill(P.gravity, d["gravity"])
    P.mass = d["mass"]
    _fill(P.inertia, d["inertia"])
    _fill(P.inertia_inv, d["inertia_inv"])
    _fill(P.alloc, d["alloc"])
    _fill(P.alloc_pinv, d["alloc_pinv"])
    _fill(P.wrench_map, d["wrench_map"])
    _fill(P.motor_dir, d["motor_dir"][:M])
    P.cq = d["cq"]
    P.use_rps, P.use_discrete_approximation, P.integration_rk4 = d["use_rps"], d["use_discrete_approximation"], d["integration_rk4"]
    P.min_thrust, P.max_thrust, P.max_rate = d["min_thrust"], d["max_thrust"], d["max_rate"]
    P.max_yaw_rate = d["max_yaw_rate"]
    for name in ("lin_drag_linear", "lin_drag_quadratic", "ang_drag_linear", "ang_drag_quadratic"):
        _fill(getattr(P, name), d[name])
    P.linear_damping, P.angular_damping = d["linear_damping"], d["angular_damping"]
    P.max_linear_velocity, P.max_angular_velocity = d["max_linear_velocity"], d["max_angular_velocity"]
    P.collision_radius = d["collision_radius"]
    _fill(P.gains_uniform, d.get("gains_uniform", [0.0] * 12))
    P.tau_inc_uniform = d.get("tau_inc_uniform", 0.0)
    P.tau_dec_uniform = d.get("tau_dec_uniform", 0.0)
    return P
