"""GPU: the robot plug-in surface of SURVEY 8(b) (robots/base_robot.py:10-63, robot_manager.py:486-489, base_multirotor.py:296-307).

A robot CLASS registered with robot_registry.register that overrides step(action) is called by the host once per physics
sub-step; what it leaves in robot_force_tensors / robot_torque_tensors (each body's wrench in that body's frame) is reduced to
the net wrench on the rigid composite and integrated (AGX_CTRL_WRENCH + AGX_LAUNCH_BODY_WRENCH).  BaseMultirotor.step itself --
what such a class reaches through super().step(action) -- is ONE launch (agx_robot_step).  Checked bit for bit against the
oracle's restatement of the same path (orc.robot_step -> orc.net_body_wrench -> orc.integrate), and against the fused step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EXTRA_F = np.array([0.3, -0.2, 0.5], np.float32)  # constant extra force on the root body, in its frame [N]
EXTRA_TZ = np.float32(0.01)                       # and a torque about the z axis of the third motor link [N m]


def npy(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def _register(name, cfg_name, with_extra):
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.registry.robot_registry import robot_registry
    from aerial_gym_simulator_amd.robots.base_multirotor import BaseMultirotor

    class ToyRobot(BaseMultirotor):
        """the reference's contract: step(action) leaves the per-body tensors Isaac Gym would apply"""

        calls = 0

        def step(self, action_tensor):
            super().step(action_tensor)  # update_states, controller, allocation, motors, drag, disturbance: one launch
            type(self).calls += 1
            if with_extra:
                # (a torque on the third motor link; root-link robots apply everything at body 0 but keep their motor links)
                third = int(self.cfg.control_allocator_config.application_mask[2])
                self.robot_force_tensors[:, 0, :] += torch.from_numpy(EXTRA_F).to(self.robot_force_tensors.device)
                self.robot_torque_tensors[:, third, 2] += float(EXTRA_TZ)

    robot_registry.register(name, ToyRobot, robot_registry.get_robot_config(cfg_name))
    return ToyRobot


@pytest.mark.parametrize("robot,controller,env_name,substeps", [
    ("base_quadrotor", "lee_position_control", "empty_env", 1),
    ("base_quadrotor", "lee_velocity_control", "env_with_random_boxes", 10),
    ("base_octarotor", "octarotor_velocity_control", "empty_env", 1),     # tilted motor links, non-zero drag
    ("base_quad_root_link_control", "lee_position_control", "empty_env", 1),  # the allocator's wrench at the root link (mask [0])
])
def test_toy_robot_subclass_is_stepped_like_the_reference_and_matches_the_oracle(orc, robot, controller, env_name, substeps):
    from aerial_gym_simulator_amd.robots.robot_model import link_frames
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    name = f"toy_{robot}_{controller}"
    cls = _register(name, robot, with_extra=True)
    n = 160
    env = SimBuilder().build_env(sim_name="base_sim", env_name=env_name, robot_name=name, controller_name=controller, device=DEV,
                                 args={"rng_seed": 5}, num_envs=n, headless=True, use_warp=False)
    rob = env.robot_manager.robot
    assert rob.external_robot and not rob.external_controller and isinstance(rob, cls)
    assert env.cfg.env.num_physics_steps_per_env_step_mean == substeps
    env.reset()
    g = env.get_obs()
    pd = rob.params_dict
    P = orc.make_params(pd)
    ctrl, mm = rob.controller, rob.control_allocator.motor_model
    gains = [np.tile(((np.array(ctrl.gains_max, np.float32) + np.array(ctrl.gains_min, np.float32)) / np.float32(2))[3 * k:3 * k + 3], (n, 1))
             for k in range(4)]
    NB = int(g["robot_force_tensor"].shape[1])
    links = [int(b) for b in rob.cfg.control_allocator_config.application_mask]  # the motor links' body indices
    mask = [0] if pd["root_link_mode"] else links                                # where the allocator's output goes (base_multirotor.py:152-159)
    assert NB == max(links) + 1 == {"base_quadrotor": 9, "base_octarotor": 17, "base_quad_root_link_control": 9}[robot]
    L, known = link_frames(rob.cfg, NB)
    rot = np.array([[L.rot[b][c] for c in range(9)] for b in range(NB)], np.float32)
    pos = np.array([[L.pos[b][c] for c in range(3)] for b in range(NB)], np.float32)
    assert [b for b, k in enumerate(known) if k] == [0] + links
    dcfg = rob.cfg.disturbance
    dmax = np.array(dcfg.max_force_and_torque_disturbance, np.float32)
    gen = torch.Generator(device=DEV).manual_seed(3)
    calls0 = cls.calls
    if getattr(ctrl, "_per_env_gains_bound", False):  # randomize_params: per-env gains, re-drawn at every reset (none happens below)
        gains = [npy(x) for x in (ctrl.K_pos_tensor_current, ctrl.K_linvel_tensor_current, ctrl.K_rot_tensor_current, ctrl.K_angvel_tensor_current)]
    for t in range(6):
        st, th = npy(g["robot_state_tensor"]), npy(mm.current_motor_thrust)
        kT, ti, td = npy(mm.motor_thrust_constant), npy(mm.motor_time_constants_increasing), npy(mm.motor_time_constants_decreasing)
        a = torch.rand(n, env.num_robot_actions, device=DEV, generator=gen) * 2 - 1
        env.step(a)
        for sub in range(substeps):
            dist = None
            if dcfg.enable_disturbance:  # apply_disturbance drawn in the kernel (device stream RNG_DISTURB + sub-step), as in the fused step
                dist = orc.rng_fill(env.rng_seed, np.full(n, env.step_counter - 1, np.int32), (1 << 20) + sub, 7)
                dist[:, 0] = (dist[:, 0] < np.float32(dcfg.prob_apply_disturbance)).astype(np.float32)
            o, F, T = orc.robot_step(P, st, npy(a), th, kT, ti, td, *gains, NB, mask, disturb=dist, disturb_max=dmax)
            F[:, 0, :] += EXTRA_F
            T[:, links[2], 2] += EXTRA_TZ
            net = orc.net_body_wrench(rot, pos, F, T)
            orc.integrate(P, st, net)
        # the per-body tensors as the LAST sub-step's step() left them, the motor thrusts, the state: bit for bit
        assert np.array_equal(npy(g["robot_force_tensor"]), F) and np.array_equal(npy(g["robot_torque_tensor"]), T), t
        assert np.array_equal(npy(mm.current_motor_thrust), th), t
        assert np.array_equal(npy(g["robot_state_tensor"]), st), t
        assert np.array_equal(npy(g["robot_actions"]), npy(a)) and int(env.sim_steps[0]) == t + 1
        assert np.array_equal(npy(g["robot_euler_angles"]), o.euler)  # update_states ran inside step(), pre-physics
    assert cls.calls - calls0 == 6 * substeps  # once per physics sub-step, like robot_manager.py:486-489
    # the extra force did something: +0.5 N upward on a 0.25 kg (quad) airframe
    assert np.isfinite(npy(g["robot_state_tensor"])).all()


def test_subclass_that_only_calls_super_flies_like_the_fused_step():
    """step() = super().step(): the split path (one launch per part, per-body tensors reduced with the link-frame table) against the
    ONE fused launch of the same robot -- same initial state, same actions; the motor thrusts are bit-identical (same controller,
    allocation and motor model on the same numbers) as long as the states are, the states agree to fp32 rounding of the wrench
    reduction (wrench_map folded ahead of time vs per-body sum)."""
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    _register("toy_passthrough_octarotor", "base_octarotor", with_extra=False)
    n = 128
    envs = [SimBuilder().build_env(sim_name="base_sim", env_name="env_with_random_boxes", robot_name=r, controller_name="octarotor_velocity_control",
                                   device=DEV, args={"rng_seed": 9}, num_envs=n, headless=True, use_warp=False)
            for r in ("base_octarotor", "toy_passthrough_octarotor")]
    assert envs[1].robot_manager.robot.external_robot and not envs[0].robot_manager.robot.external_robot
    for e in envs:
        e.reset()
    assert torch.equal(envs[0].global_tensor_dict["robot_state_tensor"], envs[1].global_tensor_dict["robot_state_tensor"])
    gen = torch.Generator(device=DEV).manual_seed(1)
    worst = 0.0
    for t in range(8):
        a = torch.rand(n, 4, device=DEV, generator=gen) * 2 - 1
        for e in envs:
            e.step(a)
        s0, s1 = (e.global_tensor_dict["robot_state_tensor"] for e in envs)
        worst = max(worst, float((s0 - s1).abs().max()))
        if t == 0:
            assert worst < 2e-5, worst  # 10 sub-steps from identical states
        assert torch.equal(envs[0].global_tensor_dict["crashes"], envs[1].global_tensor_dict["crashes"])
        assert torch.equal(envs[0].sim_steps, envs[1].sim_steps)
    assert worst < 2e-3, worst


def test_builtin_robot_step_on_its_own_and_through_a_task(orc):
    """BaseMultirotor.step(action) called by hand on the default robot writes the tensors (it raised until round 5); a
    position_setpoint_task over a plug-in robot takes the general path (no one-call fast plan), reward and flags included."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    from aerial_gym_simulator_amd.sim.sim_builder import SimBuilder

    env = SimBuilder().build_env(sim_name="base_sim", env_name="empty_env", robot_name="base_quadrotor", controller_name="lee_position_control",
                                 device=DEV, num_envs=32, headless=True, use_warp=False)
    env.reset()
    g = env.get_obs()
    rob = env.robot_manager.robot
    before = g["robot_state_tensor"].clone()
    rob.step(torch.zeros(32, 4, device=DEV))
    F, T = g["robot_force_tensor"], g["robot_torque_tensor"]
    u = rob.control_allocator.motor_model.current_motor_thrust
    assert torch.equal(F[:, 5:9, 2], u) and float(F[:, :5].abs().max()) == 0.0 and float(F[:, 5:9, :2].abs().max()) == 0.0
    assert torch.equal(T[:, 5:9, 2], (0.01 * u) * (-torch.tensor([1.0, -1.0, 1.0, -1.0], device=DEV)))
    assert torch.equal(g["robot_state_tensor"], before)  # nothing integrated
    with pytest.raises(ValueError, match="correct number of environments"):
        rob.step(torch.zeros(31, 4, device=DEV))
    _register("toy_task_quadrotor", "base_quadrotor", with_extra=True)
    old = (cfg.robot_name, cfg.device, cfg.controller_name, cfg.args)
    try:
        cfg.robot_name, cfg.device, cfg.controller_name, cfg.args = "toy_task_quadrotor", DEV, "lee_position_control", {}
        task = task_registry.make_task("position_setpoint_task", seed=3, num_envs=64, headless=True)
        assert task._plan is None and task.sim_env.robot_manager.robot.external_robot
        task.reset()
        for _ in range(5):
            obs, rew, term, trunc, _ = task.step(torch.zeros(64, 4, device=DEV))
        assert torch.isfinite(rew).all() and obs["observations"].shape == (64, 13) and int(task.sim_env.sim_steps[0]) == 5
    finally:
        cfg.robot_name, cfg.device, cfg.controller_name, cfg.args = old


@pytest.mark.parametrize("case", ["quad_position", "quad_velocity", "quad_attitude", "quad_acceleration", "quad_no_control", "octarotor_position",
                                  "octarotor_velocity", "octarotor_fully_actuated", "quad_rates", "quad_velocity_steering"])
def test_robot_step_per_body_tensors_vs_the_reference_arrays(orc, parity, case):
    """VERDICT r05 missing-5 / next-2: agx_robot_step's per-body force / torque ([N][bodies][3]) ENTRY BY ENTRY against
    robot_force_tensor / robot_torque_tensor as the reference's BaseMultirotor.step left them on the recorded inputs
    (base_multirotor.py:236-285; goldens `force` / `torque`, oracle/gen_golden.py:326-327): <= 1e-5 max(1, |x|); and bit for bit
    against orc.robot_step on the same inputs."""
    from conftest import golden_params, load_golden
    from gpu_harness import DynHarness

    g = load_golden("step_" + case)
    pd = golden_params(g)
    n, NB = g["state"].shape[1], g["force"].shape[2]
    mask = [int(b) for b in g["application_mask"]]
    P = orc.make_params(pd)
    H = DynHarness(pd, n)
    H.set(kT=g["kT"], tau_inc=g["tau_inc"], tau_dec=g["tau_dec"])
    H.set_gains(g["Kp"], g["Kv"], g["KR"], g["Kw"])
    for k in range(g["state"].shape[0]):
        H.set(state=g["state"][k], thrust=g["thrust_in"][k])
        dist = g["disturb"][k] if g["disturb"][k].any() else None
        if dist is not None:
            H.set_disturb(dist[None], g["disturb_max"])
        F, T = H.robot_step(g["action"][k], NB, mask)
        th = g["thrust_in"][k].copy()
        _, Fo, To = orc.robot_step(P, g["state"][k].copy(), g["action"][k], th, g["kT"], g["tau_inc"], g["tau_dec"], g["Kp"], g["Kv"], g["KR"], g["Kw"],
                                   NB, mask, disturb=dist, disturb_max=g["disturb_max"])
        assert np.array_equal(F, Fo) and np.array_equal(T, To) and np.array_equal(H.get("thrust"), th), (case, k)
        for name, got, ref in (("force", F, g["force"][k]), ("torque", T, g["torque"][k])):
            err = float((np.abs(got - ref) / np.maximum(1.0, np.abs(ref))).max())
            parity.check(f"robot_step_{name}_vs_reference[{case}]", err, 1e-5, "|err| / max(1, |x|)", k)
