"""TEST INFRASTRUCTURE -- generates tests/golden/*.npz by running the REFERENCE's own code.

Run in the build container only (needs /root/reference, read-only):

    python oracle/gen_golden.py            # writes tests/golden/*.npz

What is pinned by these vectors (everything that exists as source in the reference):
  math_utils.npz           aerial_gym/utils/math.py helpers                     (rows a1, a22)
  robot_<name>.npz         composite mass / inertia / motor wrench map derived from the URDFs
                           with the algorithm of robots/robot_manager.py:295-435 (row a26)
  step_<robot>_<ctrl>.npz  BaseMultirotor.step = update_states + controller + allocation +
                           motor model + drag + disturbance, K chained sub-steps (rows a1-a14)
  reward_position.npz      position_setpoint_task.compute_reward (row a17)
  reward_navigation.npz    navigation_task.compute_reward (row a18)
  trace_position_64.npz    BASELINE config 1: 64 envs, empty_env, reference control + reference
                           reward + reference reset, with the ORACLE's rigid-body integrator
                           in the loop (PhysX is a closed binary: that one piece is unpinned)

The states between chained sub-steps are advanced with the oracle's integrator, which only
serves to produce plausible input sequences -- each recorded (input -> output) pair of the
reference code is valid regardless of how the next input was made.
"""
import json
import math
import os
import sys
import xml.etree.ElementTree as ET

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cr_torch  # noqa: E402

CR = cr_torch.requested()  # `--cr` / AGX_GOLDEN_CR=1: the reference with correctly rounded elementary functions -> tests/golden/cr/
# TorchScript off (the switchable patch of cr_torch must reach the scripted functions); AGX_GOLDEN_JIT=1 keeps it on: the
# `*_cr` arrays then cannot be made, everything else comes out bit-identical (a test checks exactly that)
JIT = os.environ.get("AGX_GOLDEN_JIT") == "1" and not CR
if not JIT:
    cr_torch.prepare_environment()

import numpy as np  # noqa: E402
import torch  # noqa: E402

if not JIT:
    cr_torch.install()
    cr_torch.enable(CR)
import oracle as orc  # noqa: E402
import ref_shells  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", *(["cr"] if CR else []))
torch.set_num_threads(1)


# --------------------------------------------------------------------------------------
# URDF -> link table -> composite rigid body (robots/robot_manager.py:295-435)
# --------------------------------------------------------------------------------------
def rpy_to_mat(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def parse_urdf(path):
    root = ET.parse(path).getroot()
    links = {}
    for ln in root.findall("link"):
        ine = ln.find("inertial")
        m = float(ine.find("mass").get("value"))
        i = ine.find("inertia")
        g = lambda k: float(i.get(k, "0"))  # noqa: E731
        I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        col = ln.find("collision")
        rad = None
        if col is not None and col.find("geometry").find("sphere") is not None:
            rad = float(col.find("geometry").find("sphere").get("radius"))
        links[ln.get("name")] = dict(mass=m, inertia=I, xyz=np.zeros(3), R=np.eye(3), rpy=[0, 0, 0], radius=rad)
    for jn in root.findall("joint"):
        child = jn.find("child").get("link")
        o = jn.find("origin")
        xyz = [float(v) for v in o.get("xyz").split()]
        rpy = [float(v) for v in o.get("rpy", "0 0 0").split()]
        links[child]["xyz"] = np.array(xyz)
        links[child]["rpy"] = rpy
        links[child]["R"] = rpy_to_mat(*rpy)
    return links


def composite(links):
    mass = sum(l["mass"] for l in links.values())
    com = sum(l["mass"] * l["xyz"] for l in links.values()) / mass
    J = np.zeros((3, 3))
    for l in links.values():
        Ib = l["R"] @ l["inertia"] @ l["R"].T
        d = -(l["xyz"] - com)
        m = l["mass"]
        Ib = Ib + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        J += Ib
    return mass, com, J


def robot_constants(name, cfg):
    urdf = os.path.join(ref_shells.REFERENCE_ROOT, "resources", "robots", name, name + ".urdf")
    links = parse_urdf(urdf)
    mass, com, J = composite(links)
    ca = cfg.control_allocator_config
    M = ca.num_motors
    cq = ca.motor_model_config.thrust_to_torque_ratio
    W = np.zeros((6, M))
    pos = np.zeros((M, 3))
    Rm = np.zeros((M, 3, 3))
    for i in range(M):
        l = links[f"motor_{i}"]
        ez = l["R"][:, 2]
        r = l["xyz"] - com
        W[0:3, i] = ez
        W[3:6, i] = np.cross(r, ez) - cq * ca.motor_directions[i] * ez
        pos[i] = l["xyz"]
        Rm[i] = l["R"]
    return dict(
        mass=np.float64(mass), com=com, inertia=J, wrench_map=W, motor_pos=pos, motor_rot=Rm,
        alloc=np.array(ca.allocation_matrix, dtype=np.float64),
        collision_radius=np.float64(links["base_link"]["radius"]),
        motor_rpy=np.array([links[f"motor_{i}"]["rpy"] for i in range(M)]),
        base_mass=np.float64(links["base_link"]["mass"]), base_inertia=links["base_link"]["inertia"],
        motor_mass=np.float64(links["motor_0"]["mass"]),
    )


def params_dict(robot_cfg, ctrl_cfg, controller, consts, dt=0.01, gravity=(0.0, 0.0, -9.81)):
    ca = robot_cfg.control_allocator_config
    mm = ca.motor_model_config
    A = np.array(ca.allocation_matrix, dtype=np.float32)
    Apinv = torch.linalg.pinv(torch.tensor(ca.allocation_matrix, dtype=torch.float32)).numpy()
    J32 = consts["inertia"].astype(np.float32)
    return dict(
        num_motors=ca.num_motors, num_actions=ctrl_cfg.num_actions if controller != "no_control" else ca.num_motors,
        controller=controller, root_link_mode=int(ca.force_application_level == "root_link"),
        dt=dt, gravity=list(gravity), mass=float(np.float32(consts["mass"])),
        inertia=J32.reshape(-1).tolist(),
        inertia_inv=np.linalg.inv(J32.astype(np.float64)).astype(np.float32).reshape(-1).tolist(),
        alloc=A.reshape(-1).tolist(), alloc_pinv=Apinv.reshape(-1).tolist(),
        wrench_map=consts["wrench_map"].astype(np.float32).reshape(-1).tolist(),
        motor_dir=[float(x) for x in ca.motor_directions], cq=float(mm.thrust_to_torque_ratio),
        use_rps=int(mm.use_rps), use_discrete_approximation=int(mm.use_discrete_approximation),
        integration_rk4=int(getattr(mm, "integration_scheme", "rk4") != "euler"),
        min_thrust=float(mm.min_thrust), max_thrust=float(mm.max_thrust), max_rate=float(mm.max_thrust_rate),
        max_yaw_rate=float(getattr(ctrl_cfg, "max_yaw_rate", math.pi / 3)),
        lin_drag_linear=list(robot_cfg.damping.linvel_linear_damping_coefficient),
        lin_drag_quadratic=list(robot_cfg.damping.linvel_quadratic_damping_coefficient),
        ang_drag_linear=list(robot_cfg.damping.angular_linear_damping_coefficient),
        ang_drag_quadratic=list(robot_cfg.damping.angular_quadratic_damping_coefficient),
        linear_damping=float(robot_cfg.robot_asset.linear_damping),
        angular_damping=float(robot_cfg.robot_asset.angular_damping),
        max_linear_velocity=float(robot_cfg.robot_asset.max_linear_velocity),
        max_angular_velocity=float(robot_cfg.robot_asset.max_angular_velocity),
        collision_radius=float(consts["collision_radius"]),
    )


# --------------------------------------------------------------------------------------
def gen_math(rng):
    m = ref_shells.ref("utils.math")
    n = 256
    q = torch.nn.functional.normalize(torch.randn(n, 4, generator=rng), dim=1)
    q2 = torch.nn.functional.normalize(torch.randn(n, 4, generator=rng), dim=1)
    v = torch.randn(n, 3, generator=rng) * 3
    t = torch.randn(n, 3, generator=rng)
    e = (torch.rand(n, 3, generator=rng) - 0.5) * 2 * math.pi * torch.tensor([1.0, 0.49, 1.0])
    out = dict(
        q=q, q2=q2, v=v, t=t, e=e,
        quat_mul=m.quat_mul(q, q2), quat_rotate=m.quat_rotate(q, v),
        quat_rotate_inverse=m.quat_rotate_inverse(q, v), quat_apply=m.quat_apply(q, v),
        quat_apply_inverse=m.quat_apply_inverse(q, v), tf_apply=m.tf_apply(q, t, v),
        euler=m.get_euler_xyz_tensor(q), ssa_euler=m.ssa(m.get_euler_xyz_tensor(q)),
        vehicle_quat=m.vehicle_frame_quat_from_quat(q), quat_from_euler=m.quat_from_euler_xyz_tensor(e),
        rotmat=m.quat_to_rotation_matrix(q), quat_axis2=m.quat_axis(q, 2),
    )
    np.savez(os.path.join(OUT, "math_utils.npz"), **{k: x.numpy() for k, x in out.items()})


# --------------------------------------------------------------------------------------
def make_ref_robot(robot_cfg, controller_name, n, consts, seed):
    """Reference BaseMultirotor on CPU with a hand-made global_tensor_dict (no Isaac Gym)."""
    from aerial_gym.config.env_config.empty_env import EmptyEnvCfg

    bm = ref_shells.ref("robots.base_multirotor")
    ref_shells.ref("control")  # registers controllers
    EmptyEnvCfg.env.num_envs = n
    torch.manual_seed(seed)
    robot = bm.BaseMultirotor(robot_cfg, controller_name, EmptyEnvCfg, "cpu")
    M = robot_cfg.control_allocator_config.num_motors
    B = 1 + 2 * M
    state = torch.zeros(n, 13)
    state[:, 6] = 1.0
    gtd = {
        "dt": 0.01,
        "gravity": torch.tensor([0.0, 0.0, -9.81]).expand(n, -1),
        "robot_state_tensor": state,
        "robot_position": state[:, 0:3],
        "robot_orientation": state[:, 3:7],
        "robot_linvel": state[:, 7:10],
        "robot_angvel": state[:, 10:13],
        "robot_force_tensor": torch.zeros(n, B, 3),
        "robot_torque_tensor": torch.zeros(n, B, 3),
        "env_bounds_min": -torch.ones(n, 3),
        "env_bounds_max": torch.ones(n, 3),
        "robot_mass": torch.full((n,), float(np.float32(consts["mass"]))),
        "robot_inertia": torch.tensor(consts["inertia"], dtype=torch.float32).expand(n, 3, 3).contiguous(),
    }
    robot.init_tensors(gtd)
    return robot, gtd


def link_wrench(u, W):
    """net body wrench of the motor-link forces: sum_j W[:, j] u_j accumulated motor by motor from 0 in float32 -- the order
    of the oracle (and of the kernels); a BLAS `u @ W.T` rounds differently (blocked, fused multiply-adds)"""
    u, W = np.asarray(u, np.float32), np.asarray(W, np.float32)
    acc = np.zeros((u.shape[0], 6), np.float32)
    for j in range(u.shape[1]):
        acc = acc + W[None, :, j] * u[:, j:j + 1]
    return acc


def advance_state(gtd, mask, consts, P):
    """the state after the reference's force / torque tensors have acted for one sub-step: motor links' sum + the root link's
    entry, through the oracle's integrator (PhysX is a closed binary)"""
    u = gtd["robot_force_tensor"][:, mask, 2].numpy().astype(np.float32)
    bw = link_wrench(u, consts["wrench_map"].astype(np.float32))
    bw[:, 0:3] += gtd["robot_force_tensor"][:, 0, :].numpy()
    bw[:, 3:6] += gtd["robot_torque_tensor"][:, 0, :].numpy()
    st = np.ascontiguousarray(gtd["robot_state_tensor"].numpy().astype(np.float32))
    orc.integrate(P, st, np.ascontiguousarray(bw))
    return st


def random_state(n, rng, spread=1.0, tilt=0.6):
    m = ref_shells.ref("utils.math")
    s = torch.zeros(n, 13)
    s[:, 0:3] = (torch.rand(n, 3, generator=rng) - 0.5) * 2 * spread
    e = (torch.rand(n, 3, generator=rng) - 0.5) * 2 * torch.tensor([tilt, tilt, math.pi])
    s[:, 3:7] = m.quat_from_euler_xyz_tensor(e)
    s[:, 7:10] = (torch.rand(n, 3, generator=rng) - 0.5) * 2.0
    s[:, 10:13] = (torch.rand(n, 3, generator=rng) - 0.5) * 2.0
    return s


def motor_arrays(robot, n, M):
    mmod = robot.control_allocator.motor_model
    kT = mmod.motor_thrust_constant if robot.cfg.control_allocator_config.motor_model_config.use_rps else torch.ones(n, M)
    return (mmod.current_motor_thrust, kT, mmod.motor_time_constants_increasing, mmod.motor_time_constants_decreasing)


def gains(robot, n):
    c = robot.controller
    if hasattr(c, "K_pos_tensor_current"):
        return (c.K_pos_tensor_current, c.K_linvel_tensor_current, c.K_rot_tensor_current, c.K_angvel_tensor_current)
    z = torch.zeros(n, 3)
    return (z, z, z, z)


def gen_step(robot_name, robot_cfg, controller_name, ctrl_key, consts, n=64, K=6, seed=7, action_scale=1.0):
    rng = torch.Generator().manual_seed(seed)
    robot, gtd = make_ref_robot(robot_cfg, controller_name, n, consts, seed)
    ctrl_cfg = robot.controller_config
    if getattr(ctrl_cfg, "randomize_params", False):
        robot.controller.randomize_params(torch.arange(n))
    M = robot_cfg.control_allocator_config.num_motors
    A = robot.num_actions
    pd = params_dict(robot_cfg, ctrl_cfg, ctrl_key, consts)
    P = orc.make_params(pd)
    gtd["robot_state_tensor"][:] = random_state(n, rng)
    thrust, kT, tinc, tdec = motor_arrays(robot, n, M)
    Kp, Kv, KR, Kw = gains(robot, n)
    rec = {k: [] for k in ("state", "action", "thrust_in", "thrust_out", "euler", "qveh", "vveh", "vbody", "wbody",
                           "wrench_cmd", "force", "torque", "disturb", "action_after",
                           "thrust_out_cr", "wrench_cmd_cr", "wbody_cr", "state_next_cr")}
    mask = torch.tensor(robot_cfg.control_allocator_config.application_mask)
    dist_on = bool(robot_cfg.disturbance.enable_disturbance)
    dmax = torch.tensor(robot_cfg.disturbance.max_force_and_torque_disturbance)
    for k in range(K):
        if ctrl_key == "no_control":
            action = torch.rand(n, A, generator=rng) * 2.5 - 0.2
        elif ctrl_key == "fully_actuated":
            action = torch.cat([(torch.rand(n, 3, generator=rng) - 0.5) * 2, torch.randn(n, 4, generator=rng)], dim=1)
        else:
            action = (torch.rand(n, A, generator=rng) - 0.5) * 2 * action_scale
            if k == K - 1:
                action = action * 30.0  # exercise the +-10 clip and the yaw-rate clamp
        rec["state"].append(gtd["robot_state_tensor"].clone())
        rec["action"].append(action.clone())
        rec["thrust_in"].append(thrust.clone())
        sd = 1000 + 17 * k
        if not CR and not JIT:
            # the same code on the same inputs with correctly rounded elementary functions: the reference's own last-bit
            # freedom on this sample (cr_torch.py); the motor thrusts (the one piece of state step() mutates) are restored
            thrust_keep = thrust.clone()
            with cr_torch.correctly_rounded():
                torch.manual_seed(sd)
                robot.step(action.clone())
            rec["thrust_out_cr"].append(thrust.clone())
            rec["wrench_cmd_cr"].append(robot.controller.wrench_command.clone() if hasattr(robot.controller, "wrench_command")
                                        else torch.zeros(n, 6))
            rec["wbody_cr"].append(robot.robot_body_angvel.clone())
            rec["state_next_cr"].append(torch.from_numpy(advance_state(gtd, mask, consts, P)))
            thrust[:] = thrust_keep
        torch.manual_seed(sd)
        robot.step(action.clone())
        # replay the RNG draws made by apply_disturbance (base_multirotor.py:213-234)
        d = torch.zeros(n, 7)
        if dist_on:
            torch.manual_seed(sd)
            d[:, 0] = torch.bernoulli(robot_cfg.disturbance.prob_apply_disturbance * torch.ones(n))
            d[:, 1:4] = torch.rand_like(dmax[0:3].expand(n, -1))
            d[:, 4:7] = torch.rand_like(dmax[3:6].expand(n, -1))
        rec["disturb"].append(d)
        rec["thrust_out"].append(thrust.clone())
        rec["action_after"].append(robot.action_tensor.clone())
        rec["euler"].append(robot.robot_euler_angles.clone())
        rec["qveh"].append(robot.robot_vehicle_orientation.clone())
        rec["vveh"].append(robot.robot_vehicle_linvel.clone())
        rec["vbody"].append(robot.robot_body_linvel.clone())
        rec["wbody"].append(robot.robot_body_angvel.clone())
        wc = robot.controller.wrench_command.clone() if hasattr(robot.controller, "wrench_command") else torch.zeros(n, 6)
        rec["wrench_cmd"].append(wc)
        rec["force"].append(gtd["robot_force_tensor"].clone())
        rec["torque"].append(gtd["robot_torque_tensor"].clone())
        # advance the state with the oracle integrator (input generation only)
        gtd["robot_state_tensor"][:] = torch.from_numpy(advance_state(gtd, mask, consts, P))
    out = {k: torch.stack(v).numpy() for k, v in rec.items() if v}
    out.update(kT=kT.numpy(), tau_inc=tinc.numpy(), tau_dec=tdec.numpy(), Kp=Kp.numpy(), Kv=Kv.numpy(),
               KR=KR.numpy(), Kw=Kw.numpy(), disturb_max=dmax.numpy(), application_mask=mask.numpy(),
               params_json=np.array(json.dumps(pd)))
    np.savez(os.path.join(OUT, f"step_{robot_name}_{ctrl_key}.npz"), **out)
    print(f"step_{robot_name}_{ctrl_key}: ok  wrench[0]={out['wrench_cmd'][0, 0]}")


# --------------------------------------------------------------------------------------
def gen_rewards(rng):
    ref_shells.install_task_shells()
    pt = ref_shells.ref("task.position_setpoint_task.position_setpoint_task")
    nt = ref_shells.ref("task.navigation_task.navigation_task")
    m = ref_shells.ref("utils.math")
    n = 512
    state = random_state(n, rng, spread=6.0)
    state[: n // 8, 0:3] *= 3.0  # some beyond the 8 m crash radius
    target = torch.zeros(n, 3)
    qveh = m.vehicle_frame_quat_from_quat(state[:, 3:7])
    wbody = torch.randn(n, 3, generator=rng)
    vbody = torch.randn(n, 3, generator=rng)
    crashes_in = torch.rand(n, generator=rng) < 0.1
    crashes = crashes_in.clone()
    pos_err = m.quat_apply_inverse(qveh, target - state[:, 0:3])
    actions = torch.rand(n, 4, generator=rng)
    reward, crashes_out = pt.compute_reward(pos_err, state[:, 7:10], state[:, 3:7], wbody, crashes, 1.0,
                                            actions, actions, {"x": torch.zeros(1)})
    obs = torch.zeros(n, 13)
    obs[:, 0:3] = target - state[:, 0:3]
    obs[:, 3:7] = state[:, 3:7]
    obs[:, 7:10] = vbody
    obs[:, 10:13] = wbody
    np.savez(os.path.join(OUT, "reward_position.npz"), state=state.numpy(), target=target.numpy(), qveh=qveh.numpy(),
             wbody=wbody.numpy(), vbody=vbody.numpy(), crashes_in=crashes_in.numpy(), reward=reward.numpy(),
             crashes_out=crashes_out.numpy(), obs=obs.numpy())

    from aerial_gym.config.task_config.navigation_task_config import task_config as nav_cfg

    keys = ["pos_reward_magnitude", "pos_reward_exponent", "very_close_to_goal_reward_magnitude",
            "very_close_to_goal_reward_exponent", "getting_closer_reward_multiplier",
            "x_action_diff_penalty_magnitude", "x_action_diff_penalty_exponent",
            "z_action_diff_penalty_magnitude", "z_action_diff_penalty_exponent",
            "yawrate_action_diff_penalty_magnitude", "yawrate_action_diff_penalty_exponent",
            "x_absolute_action_penalty_magnitude", "x_absolute_action_penalty_exponent",
            "z_absolute_action_penalty_magnitude", "z_absolute_action_penalty_exponent",
            "yawrate_absolute_action_penalty_magnitude", "yawrate_absolute_action_penalty_exponent",
            "collision_penalty"]
    pdct = {k: torch.tensor(float(nav_cfg.reward_parameters[k])) for k in keys}
    rp = np.array([float(nav_cfg.reward_parameters[k]) for k in keys], dtype=np.float32)
    target = (torch.rand(n, 3, generator=rng) - 0.5) * 10
    prev_pe = torch.randn(n, 3, generator=rng) * 3
    pe = m.quat_rotate_inverse(qveh, target - state[:, 0:3])
    act = (torch.rand(n, 4, generator=rng) - 0.5) * 3
    pact = (torch.rand(n, 4, generator=rng) - 0.5) * 3
    cpf = 0.4
    r2, _ = nt.compute_reward(pe, prev_pe, crashes_in.clone(), act, pact, cpf, pdct)
    np.savez(os.path.join(OUT, "reward_navigation.npz"), state=state.numpy(), qveh=qveh.numpy(), target=target.numpy(),
             prev_pos_err=prev_pe.numpy(), pos_err=pe.numpy(), action=act.numpy(), prev_action=pact.numpy(),
             crashes=crashes_in.numpy(), curriculum_progress=np.float32(cpf), rp=rp, reward=r2.numpy(),
             action_transform_in=(torch.rand(64, 4, generator=rng) * 3 - 1.5).numpy())
    # action transformation (navigation_task_config.py:87-117) on the same inputs
    nav_cfg.device = "cpu"
    ati = torch.from_numpy(np.load(os.path.join(OUT, "reward_navigation.npz"))["action_transform_in"])
    ato = nav_cfg.action_transformation_function(ati.clone())
    d = dict(np.load(os.path.join(OUT, "reward_navigation.npz")))
    d["action_transform_out"] = ato.numpy()
    np.savez(os.path.join(OUT, "reward_navigation.npz"), **d)
    print("rewards: ok")


# --------------------------------------------------------------------------------------
def gen_trace(consts, controller_name="lee_position_control", ctrl_key="position", tag="position", n=64, T=260,
              episode_len=100, seed=1, zero_steps=40, sparse_every=0, hold=1):
    """sparse_every > 0 (the 1000-step trace of SURVEY 8d config 1): actions (held for `hold` steps) are stored as int16 multiples of 1/256,
    observations / states only every `sparse_every` steps and the reset draws only for steps with a reset."""
    """BASELINE config 1 assembled from reference pieces + oracle integrator.

    Follows EnvManager.step (env_manager.py:399-432) / PositionSetpointTask.step
    (position_setpoint_task.py:152-182) ordering exactly; see SURVEY appendix A.
    """
    from aerial_gym.config.robot_config.base_quad_config import BaseQuadCfg

    ref_shells.install_task_shells()
    pt = ref_shells.ref("task.position_setpoint_task.position_setpoint_task")
    m = ref_shells.ref("utils.math")
    robot, gtd = make_ref_robot(BaseQuadCfg, controller_name, n, consts, seed)
    pd = params_dict(BaseQuadCfg, robot.controller_config, ctrl_key, consts)
    P = orc.make_params(pd)
    M = 4
    mask = torch.tensor(BaseQuadCfg.control_allocator_config.application_mask)
    thrust, kT, tinc, tdec = motor_arrays(robot, n, M)
    W = consts["wrench_map"].astype(np.float32)
    act_rng = torch.Generator().manual_seed(1234)
    target = torch.zeros(n, 3)
    sim_steps = torch.zeros(n, dtype=torch.int32)
    rec = {k: [] for k in ("action", "reward", "crashes", "truncations", "obs", "state_after_step", "reset_mask",
                           "u_state", "u_tau_inc", "u_tau_dec", "u_thrust", "u_kT")}

    def replay_reset_draws(sd):
        torch.manual_seed(sd)
        us = torch.rand(n, 13)
        u1, u2, u3, u4 = torch.rand(n, M), torch.rand(n, M), torch.rand(n, M), torch.rand(n, M)
        return us, u1, u2, u3, u4

    # initial reset of all envs (EnvManager.reset -> reset_idx(all))
    sd0 = 555
    torch.manual_seed(sd0)
    robot.reset_idx(torch.arange(n))
    init = replay_reset_draws(sd0)
    init_state = gtd["robot_state_tensor"].clone()
    init_motor = [x.clone() for x in (thrust, kT, tinc, tdec)]
    prev_actions = torch.zeros(n, 4)
    actions = torch.zeros(n, 4)
    for t in range(T):
        if t < zero_steps:
            new_action = torch.zeros(n, 4)
        elif sparse_every:
            if (t - zero_steps) % hold == 0:  # a new U(-1, 1) set-point every `hold` steps, held in between
                held = torch.randint(-256, 257, (n, 4), generator=act_rng).to(torch.float32) / 256.0
            new_action = held.clone()
        else:
            new_action = torch.rand(n, 4, generator=act_rng) * 2 - 1
        prev_actions[:] = actions
        actions = new_action
        crashes = torch.zeros(n, dtype=torch.bool)
        # one physics sub-step (empty_env.py:12)
        robot.step(actions.clone())
        u = gtd["robot_force_tensor"][:, mask, 2].numpy().astype(np.float32)
        bw = link_wrench(u, W)
        bw[:, 0:3] += gtd["robot_force_tensor"][:, 0, :].numpy()
        bw[:, 3:6] += gtd["robot_torque_tensor"][:, 0, :].numpy()
        st = np.ascontiguousarray(gtd["robot_state_tensor"].numpy().astype(np.float32))
        orc.integrate(P, st, np.ascontiguousarray(bw))
        gtd["robot_state_tensor"][:] = torch.from_numpy(st)
        sim_steps += 1
        rec["state_after_step"].append(gtd["robot_state_tensor"].clone())
        # reward (stale derived tensors, appendix A #1)
        pe = m.quat_apply_inverse(robot.robot_vehicle_orientation, target - gtd["robot_position"])
        reward, crashes = pt.compute_reward(pe, gtd["robot_linvel"], gtd["robot_orientation"], robot.robot_body_angvel,
                                            crashes, 1.0, actions, prev_actions, {"x": torch.zeros(1)})
        trunc = sim_steps > episode_len
        reset_mask = crashes | trunc
        ids = reset_mask.nonzero(as_tuple=False).squeeze(-1)
        sd = 9000 + t
        draws = [torch.zeros(n, 13)] + [torch.zeros(n, M)] * 4
        if len(ids) > 0:
            torch.manual_seed(sd)
            robot.reset_idx(ids)
            draws = replay_reset_draws(sd)
            sim_steps[ids] = 0
        obs = torch.zeros(n, 13)
        obs[:, 0:3] = target - gtd["robot_position"]
        obs[:, 3:7] = gtd["robot_orientation"]
        obs[:, 7:10] = robot.robot_body_linvel
        obs[:, 10:13] = robot.robot_body_angvel
        for k, v in zip(("action", "reward", "crashes", "truncations", "obs", "reset_mask", "u_state", "u_tau_inc",
                         "u_tau_dec", "u_thrust", "u_kT"),
                        (actions, reward, crashes, trunc, obs, reset_mask, *draws)):
            rec[k].append(v.clone())
    out = {k: torch.stack(v).numpy() for k, v in rec.items()}
    if sparse_every:
        steps = np.nonzero(out["reset_mask"].any(axis=1))[0].astype(np.int32)
        for k in ("u_state", "u_tau_inc", "u_tau_dec", "u_thrust", "u_kT"):
            out[k] = out[k][steps]
        out["reset_steps"] = steps
        keep = np.arange(sparse_every - 1, T, sparse_every)
        out["kept_steps"] = keep.astype(np.int32)
        for k in ("obs", "state_after_step"):
            out[k] = out[k][keep]
        a16 = np.round(out["action"] * 256.0).astype(np.int16)
        assert np.array_equal(a16.astype(np.float32) / 256.0, out["action"])
        out["action_q8"] = a16
        del out["action"]
    out.update(init_state=init_state.numpy(), init_thrust=init_motor[0].numpy(), init_kT=init_motor[1].numpy(),
               init_tau_inc=init_motor[2].numpy(), init_tau_dec=init_motor[3].numpy(),
               init_u_state=init[0].numpy(), init_u_tau_inc=init[1].numpy(), init_u_tau_dec=init[2].numpy(),
               init_u_thrust=init[3].numpy(), init_u_kT=init[4].numpy(),
               Kp=gains(robot, n)[0].numpy(), Kv=gains(robot, n)[1].numpy(), KR=gains(robot, n)[2].numpy(),
               Kw=gains(robot, n)[3].numpy(), episode_len=np.int32(episode_len),
               min_init_state=np.array(BaseQuadCfg.init_config.min_init_state, dtype=np.float32),
               max_init_state=np.array(BaseQuadCfg.init_config.max_init_state, dtype=np.float32),
               params_json=np.array(json.dumps(pd)))
    np.savez_compressed(os.path.join(OUT, f"trace_{tag}_64.npz" if not sparse_every else f"trace_{tag}_64_long.npz"), **out)
    print(f"trace_{tag}: ok, mean reward first/last = {out['reward'][0].mean():.4f} / {out['reward'][-1].mean():.4f}, "
          f"resets = {int(out['reset_mask'].sum())}")


def register_unregistered_reference_controllers():
    """Two controller classes of the reference cannot be reached through its registry as shipped:

    * LeeVelocitySteeringAngleController (velocity_steeing_angle_controller.py:15-45) is imported by
      control/__init__.py:8-10 but never registered: registered here, unmodified, with lee_controller_config.
    * LeeRatesController.update (rates_control.py:16-30) raises for every batch size: line 25 subtracts the
      [N, 3] gravity tensor from the [N] thrust column.  The golden case runs the reference's class with that ONE
      line restated on the z component of gravity (what the attitude controller's thrust law implies); everything
      else -- reset_commands, compute_body_torque incl. the in-place yaw-rate clamp on the action view, allocation,
      motor model -- is the reference's code.  SURVEY appendix A lists this as a reference bug."""
    ctrl = ref_shells.ref("control")
    from aerial_gym.registry.controller_registry import controller_registry

    controller_registry.register_controller("lee_velocity_steering_angle_control", ctrl.LeeVelocitySteeringAngleController,
                                            ctrl.lee_controller_config)

    class LeeRatesControllerZ(ctrl.LeeRatesController):
        def update(self, command_actions):
            self.reset_commands()
            self.wrench_command[:, 2] = (command_actions[:, 0] - self.gravity[:, 2]) * self.mass[:, 0]
            self.wrench_command[:, 3:6] = self.compute_body_torque(self.robot_orientation, command_actions[:, 1:4])
            return self.wrench_command

    controller_registry.register_controller("lee_rates_control_zfix", LeeRatesControllerZ, ctrl.lee_controller_config)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = torch.Generator().manual_seed(2024)
    ref_shells.install()
    from aerial_gym.config.robot_config.base_octarotor_config import BaseOctarotorCfg
    from aerial_gym.config.robot_config.base_quad_config import BaseQuadCfg

    gen_math(rng)
    consts = {"quad": robot_constants("quad", BaseQuadCfg), "octarotor": robot_constants("octarotor", BaseOctarotorCfg)}
    for name, c in consts.items():
        if not CR:  # URDF constants: no elementary function involved
            np.savez(os.path.join(OUT, f"robot_{name}.npz"), **c)
        print(name, "mass", c["mass"], "J diag", np.diag(c["inertia"]), "com", c["com"],
              "| max |W - A| =", np.abs(c["wrench_map"] - c["alloc"]).max())
    for ctrl_name, key in (("lee_position_control", "position"), ("lee_velocity_control", "velocity"),
                           ("lee_attitude_control", "attitude"), ("lee_acceleration_control", "acceleration"),
                           ("no_control", "no_control")):
        gen_step("quad", BaseQuadCfg, ctrl_name, key, consts["quad"])
    for ctrl_name, key in (("octarotor_position_control", "position"), ("octarotor_velocity_control", "velocity"),
                           ("rov_fully_actuated_control", "fully_actuated")):
        gen_step("octarotor", BaseOctarotorCfg, ctrl_name, key, consts["octarotor"])
    register_unregistered_reference_controllers()
    gen_step("quad", BaseQuadCfg, "lee_velocity_steering_angle_control", "velocity_steering", consts["quad"])
    gen_step("quad", BaseQuadCfg, "lee_rates_control_zfix", "rates", consts["quad"], action_scale=2.0)
    gen_rewards(rng)
    gen_trace(consts["quad"], "lee_position_control", "position", "position")
    gen_trace(consts["quad"], "lee_attitude_control", "attitude", "attitude", T=160)
    # SURVEY 8d config 1 as written: 1000 steps, the task's real episode length (500), zero set-point then U(-1, 1)
    # (each drawn set-point is held for 25 steps: per-step white-noise set-points, as in the 260-step trace above, make
    # the vehicles tumble away from the origin within ~400 steps and turn the comparison into a chaos test)
    gen_trace(consts["quad"], "lee_position_control", "position", "position", T=1000, episode_len=500, zero_steps=500,
              sparse_every=10, hold=25)


if __name__ == "__main__":
    main()
