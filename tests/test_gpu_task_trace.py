"""GPU: the product's Task API (task_registry.make_task -> reset/step) reproduces BASELINE
config 1's trace: 64 envs, empty_env, base_quadrotor -- reference control + reference reward +
reference reset code with the oracle's integrator in the loop (tests/golden/trace_*_64.npz),
when it is fed the same random draws (strict_rng + replayed stream)."""
import numpy as np
import pytest
from aerial_gym_simulator_amd import _lib as _agx_lib
import torch
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class ReplaySource:
    """Serves recorded U(0,1) tensors by tag, in order; anything unrecorded is an error."""

    def __init__(self, device):
        self.device, self.q = device, {}

    def push(self, tag, arr):
        self.q.setdefault(tag, []).append(torch.from_numpy(np.ascontiguousarray(arr, np.float32)).to(self.device))

    def _pop(self, tag):
        if not self.q.get(tag):
            raise AssertionError(f"no recorded draw for '{tag}'")
        return self.q[tag].pop(0)

    def rand(self, *shape, tag=""):
        t = self._pop(tag)
        assert tuple(t.shape) == tuple(shape), (tag, t.shape, shape)
        return t

    def rand_into(self, out, tag=""):
        out.copy_(self._pop(tag).view_as(out))
        return out

    def bernoulli(self, p, *shape, tag=""):
        return self._pop(tag)

    def normal_into(self, out, tag=""):
        return self.rand_into(out, tag)

    def gauss(self, mean, std):
        return mean


@pytest.mark.parametrize("tag,controller", [("position", "lee_position_control"), ("attitude", "lee_attitude_control"),
                                            ("position_long", "lee_position_control")])
def test_task_api_reproduces_config1_trace(tag, controller):
    """task_registry.make_task -> reset / step, free-running, vs the reference trace (tests/trace_util.py states the
    gates): flags bit-exact and reward <= 1e-4 absolute over the gated segment (500 steps for `position_long`, SURVEY
    8d), first-divergence step reported for the rest."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    from conftest import max_abs
    from trace_util import TRACES, run_trace_against

    name, gate_steps, tail_gate = TRACES[tag]
    g = load_golden(name)
    n = g["init_state"].shape[0]
    rs = ReplaySource(DEV)
    zeros3 = np.zeros((n, 3), np.float32)
    # construction: MotorModel.init_tensors draws (values are overwritten by the first reset)
    for t in ("motor_init_thrust", "motor_init_tau_inc", "motor_init_tau_dec", "motor_init_kT"):
        rs.push(t, np.zeros((n, 4), np.float32))

    def push_reset(us, ti, td, th, kt):
        rs.push("bounds_lo", zeros3)
        rs.push("bounds_hi", zeros3)
        rs.push("robot_state", us)
        rs.push("tau_inc", ti)
        rs.push("tau_dec", td)
        rs.push("thrust", th)
        rs.push("kT", kt)

    push_reset(g["init_u_state"], g["init_u_tau_inc"], g["init_u_tau_dec"], g["init_u_thrust"], g["init_u_kT"])
    cfg.device, cfg.controller_name = DEV, controller
    cfg.episode_len_steps = int(g["episode_len"])
    cfg.args = {"strict_rng": True, "random_source": rs}
    task = task_registry.make_task("position_setpoint_task", seed=1, num_envs=n, headless=True)
    try:
        task.reset()
        st = task.obs_dict["robot_state_tensor"].cpu().numpy()
        assert max_abs(st, g["init_state"]) < 1e-6

        def step(t, action, draws):
            if draws is not None:
                push_reset(*draws)
            obs, rew, term, trunc, info = task.step(torch.from_numpy(np.ascontiguousarray(action)).to(DEV))
            return obs["observations"].cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy(), trunc.cpu().numpy()

        first_div = run_trace_against(step, g, gate_steps, tail_gate, f"gpu_task_vs_reference_trace[{tag}]")
        if first_div is None or tail_gate is not None:
            assert all(len(v) == 0 for v in rs.q.values()), {k: len(v) for k, v in rs.q.items()}  # every draw consumed
    finally:
        cfg.args = {}
        cfg.episode_len_steps = 500


@pytest.mark.parametrize("tag,controller", [("position", "lee_position_control"), ("attitude", "lee_attitude_control"),
                                            ("position_long", "lee_position_control")])
def test_task_api_trace_is_bit_identical_to_the_oracle_loop(orc, tag, controller):
    """The same traces, GPU Task API vs the CPU oracle's env loop (tests/oracle_env.py) fed the same actions and
    reset draws: state, reward, observation and flags are BIT-IDENTICAL at every one of the 260 / 160 / 1000
    free-running steps (the kernels evaluate the oracle's IEEE operation sequence; DESIGN.md "numerics").  Whatever
    distance the trace keeps from the reference's torch arithmetic is therefore the oracle's, measured on the CPU
    (tests/test_oracle_vs_reference.py::test_trace_config1)."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    from conftest import TraceReader, golden_params
    from oracle_env import OraclePositionEnv
    from trace_util import TRACES

    g = load_golden(TRACES[tag][0])
    tr = TraceReader(g)
    n = g["init_state"].shape[0]
    rs = ReplaySource(DEV)
    zeros3 = np.zeros((n, 3), np.float32)
    for t in ("motor_init_thrust", "motor_init_tau_inc", "motor_init_tau_dec", "motor_init_kT"):
        rs.push(t, np.zeros((n, 4), np.float32))

    def push_reset(us, ti, td, th, kt):
        for name, arr in (("bounds_lo", zeros3), ("bounds_hi", zeros3), ("robot_state", us), ("tau_inc", ti), ("tau_dec", td),
                          ("thrust", th), ("kT", kt)):
            rs.push(name, arr)

    init = (g["init_u_state"], g["init_u_tau_inc"], g["init_u_tau_dec"], g["init_u_thrust"], g["init_u_kT"])
    push_reset(*init)
    cfg.device, cfg.controller_name = DEV, controller
    cfg.episode_len_steps = int(g["episode_len"])
    cfg.args = {"strict_rng": True, "random_source": rs}
    task = task_registry.make_task("position_setpoint_task", seed=1, num_envs=n, headless=True)
    ranges = dict(tau_inc=(0.04, 0.04), tau_dec=(0.04, 0.04), thrust=(0.0, 2.0), kT=(0.00000926312, 0.00001826312))
    # the oracle runs on the PRODUCT's constants (its allocation pseudo-inverse is evaluated in float64; the goldens
    # carry the reference's fp32 torch.linalg.pinv of the build container: same to 1e-6, not to the bit)
    pd = dict(task.sim_env.robot_manager.robot.params_dict)
    pd["controller"] = golden_params(g)["controller"]
    env = OraclePositionEnv(pd, n, int(g["episode_len"]), (g["Kp"], g["Kv"], g["KR"], g["Kw"]),
                            g["min_init_state"], g["max_init_state"], ranges)
    try:
        task.reset()
        env.reset_masked(np.ones(n, np.uint8), *init)
        gd = task.obs_dict
        mm = task.sim_env.robot_manager.robot.control_allocator.motor_model
        assert np.array_equal(gd["robot_state_tensor"].cpu().numpy(), env.state), "initial reset"
        assert np.array_equal(mm.current_motor_thrust.cpu().numpy(), env.thrust)
        assert np.array_equal(mm.motor_thrust_constant.cpu().numpy(), env.kT)
        n_resets = 0
        for t in range(tr.T):
            draws = tr.draws(t)
            if draws is not None:
                push_reset(*draws)
            obs, rew, term, trunc, info = task.step(torch.from_numpy(np.ascontiguousarray(tr.action(t))).to(DEV))
            o_obs, o_rew, o_crash, o_trunc, o_mask, _ = env.step(tr.action(t), draws)
            n_resets += int(o_mask.sum())
            for name, got, ref in (("state", gd["robot_state_tensor"], env.state), ("thrust", mm.current_motor_thrust, env.thrust),
                                   ("reward", rew, o_rew), ("obs", obs["observations"], o_obs),
                                   ("body angvel", gd["robot_body_angvel"], env.wbody), ("euler", gd["robot_euler_angles"], env.euler)):
                got = got.cpu().numpy()
                if not np.array_equal(got, ref):
                    bad = np.argwhere(got != ref)
                    raise AssertionError(f"step {t}: {name} differs in {len(bad)} entries, first {bad[0]}: "
                                         f"{got[tuple(bad[0])]!r} vs {ref[tuple(bad[0])]!r}; max abs {np.abs(got - ref).max():.3e}")
            assert np.array_equal(term.cpu().numpy(), o_crash.astype(bool)) and np.array_equal(trunc.cpu().numpy(), o_trunc.astype(bool)), t
        assert n_resets >= n  # resets were part of the comparison
    finally:
        cfg.args = {}
        cfg.episode_len_steps = 500


@pytest.mark.parametrize("tag,controller", [("position", "lee_position_control"), ("attitude", "lee_attitude_control"),
                                            ("position_long", "lee_position_control")])
def test_task_api_trace_is_bit_identical_to_the_reference_with_correctly_rounded_functions(tag, controller):
    """The 260 / 160 / 1000-step config-1 traces through the Task API against tests/golden/cr/: the reference's own control,
    reward and reset code evaluated with correctly rounded elementary functions (oracle/cr_torch.py; + the integrator
    restated from PhysX) -- reward, observation, crash / truncation flags of EVERY step and the state wherever the
    fixture keeps it, BIT FOR BIT, no oracle in between.  One constant is taken from the fixture: the allocation
    pseudo-inverse the reference's fp32 torch.linalg.pinv produced on the generating host (the product evaluates it in
    float64; LAPACK's last bits are host dependent, DESIGN.md "numerics")."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    from conftest import TraceReader, golden_params
    from trace_util import TRACES

    g = load_golden(TRACES[tag][0], cr=True)
    tr = TraceReader(g)
    n = g["init_state"].shape[0]
    rs = ReplaySource(DEV)
    zeros3 = np.zeros((n, 3), np.float32)
    for t in ("motor_init_thrust", "motor_init_tau_inc", "motor_init_tau_dec", "motor_init_kT"):
        rs.push(t, np.zeros((n, 4), np.float32))

    def push_reset(us, ti, td, th, kt):
        for name, arr in (("bounds_lo", zeros3), ("bounds_hi", zeros3), ("robot_state", us), ("tau_inc", ti), ("tau_dec", td),
                          ("thrust", th), ("kT", kt)):
            rs.push(name, arr)

    push_reset(g["init_u_state"], g["init_u_tau_inc"], g["init_u_tau_dec"], g["init_u_thrust"], g["init_u_kT"])
    cfg.device, cfg.controller_name = DEV, controller
    cfg.episode_len_steps = int(g["episode_len"])
    cfg.args = {"strict_rng": True, "random_source": rs}
    task = task_registry.make_task("position_setpoint_task", seed=1, num_envs=n, headless=True)
    try:
        P = task.sim_env._params
        pinv = golden_params(g)["alloc_pinv"]
        for j in range(len(pinv)):
            P.alloc_pinv[j] = pinv[j]
        task.reset()
        gd = task.obs_dict
        assert np.array_equal(gd["robot_state_tensor"].cpu().numpy(), g["init_state"]), "initial reset"
        n_resets = 0
        for t in range(tr.T):
            draws = tr.draws(t)
            if draws is not None:
                push_reset(*draws)
            obs, rew, term, trunc, info = task.step(torch.from_numpy(np.ascontiguousarray(tr.action(t))).to(DEV))
            checks = [("reward", rew, g["reward"][t])]
            if tr.kept("obs", t) is not None:
                checks.append(("obs", obs["observations"], tr.kept("obs", t)))
            if tr.kept("state_after_step", t) is not None and not g["reset_mask"][t].any():
                checks.append(("state", gd["robot_state_tensor"], tr.kept("state_after_step", t)))
            for name, got, ref in checks:
                got = got.cpu().numpy()
                if not np.array_equal(got, ref):
                    bad = np.argwhere(got != ref)
                    raise AssertionError(f"step {t}: {name} differs in {len(bad)} entries, first {bad[0]}: "
                                         f"{got[tuple(bad[0])]!r} vs {ref[tuple(bad[0])]!r}; max abs {np.abs(got - ref).max():.3e}")
            assert np.array_equal(term.cpu().numpy(), g["crashes"][t]) and np.array_equal(trunc.cpu().numpy(), g["truncations"][t]), t
            n_resets += int(g["reset_mask"][t].sum())
        assert n_resets >= n and all(len(v) == 0 for v in rs.q.values())
    finally:
        cfg.args = {}
        cfg.episode_len_steps = 500


def test_sync_free_mode_runs_and_resets():
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = DEV, "lee_position_control", 30, {}
    try:
        n = 4096
        task = task_registry.make_task("position_setpoint_task", seed=5, num_envs=n, headless=True)
        task.reset()
        a = torch.rand(n, 4, device=DEV) * 2 - 1
        n_trunc = 0
        for i in range(100):
            obs, rew, term, trunc, _ = task.step(a)
            n_trunc += int(trunc.sum())
        assert torch.isfinite(obs["observations"]).all() and torch.isfinite(rew).all()
        assert int(task.sim_env.sim_steps.max()) <= 31
        assert n_trunc >= 3 * n - 10  # every env timed out ~3 times
        p = task.obs_dict["robot_position"]
        assert float(p.abs().max()) < 20.0
    finally:
        cfg.episode_len_steps = 500


def test_step_is_graph_capture_safe():
    """Sync-free task.step() issues no host synchronisation and no allocation: two steps (one per reset-flag
    parity) captured into a hipGraph and replayed give the same trajectory as eager stepping, resets included."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = DEV, "lee_position_control", 13, {}
    try:
        n = 2048
        ta = task_registry.make_task("position_setpoint_task", seed=8, num_envs=n, headless=True)
        tb = task_registry.make_task("position_setpoint_task", seed=8, num_envs=n, headless=True)
        ta.reset()
        tb.reset()
        act = torch.rand(n, 4, device=DEV) * 2 - 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a side stream, as graph capture wants it
            for _ in range(2):
                ta.step(act)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(2):
            tb.step(act)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(2):
                ta.step(act)
        for _ in range(2):
            tb.step(act)  # the capture itself does not execute
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(ta.obs_dict["robot_state_tensor"], tb.obs_dict["robot_state_tensor"])
        for _ in range(20):  # 40 more steps: past the 13-step episode limit several times
            graph.replay()
            for _ in range(2):
                tb.step(act)
        torch.cuda.synchronize()
        assert torch.equal(ta.obs_dict["robot_state_tensor"], tb.obs_dict["robot_state_tensor"])
        assert torch.equal(ta.task_obs["observations"], tb.task_obs["observations"])
        assert torch.equal(ta.rewards, tb.rewards) and torch.equal(ta.truncations, tb.truncations)
        assert int(tb.sim_env.sim_steps.max()) <= 14
    finally:
        cfg.episode_len_steps = 500


@pytest.mark.parametrize("n,randomize", [(8192, False), (1000, True), (17, False)])
def test_four_lanes_per_env_kernel_is_bit_identical_to_the_one_lane_kernel(n, randomize, monkeypatch):
    """k_env_step_quad_position (agx_quad_math.h: a 3-vector / quaternion per register, components in the lanes of a quad)
    against k_env_step<4, position>: the same IEEE operations in the same order, so every buffer the step touches is equal
    bit for bit -- over several episodes, with per-env (randomised) gains / motor constants too, and with a partial last
    wave.  agx_set_option("env_step_quad", 0) selects the one-lane kernel."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import position_setpoint_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    cfg.device, cfg.controller_name, cfg.episode_len_steps, cfg.args = DEV, "lee_position_control", 40, {}
    ctl_cfg = None
    try:
        if randomize:  # per-env gains (B.gains bound) instead of the uniform kernel-argument constants
            from aerial_gym_simulator_amd.config.controller_config import lee_controller_config as ctl_cfg

            old_rand = ctl_cfg.randomize_params
            ctl_cfg.randomize_params = True
        one = task_registry.make_task("position_setpoint_task", seed=21, num_envs=n, headless=True)
        four = task_registry.make_task("position_setpoint_task", seed=21, num_envs=n, headless=True)
        one.reset()
        four.reset()
        g = torch.Generator(device=DEV).manual_seed(9)
        for t in range(130):
            a = torch.rand(n, 4, device=DEV, generator=g) * 2 - 1
            if t % 17 == 3:
                a = a * 30.0  # beyond the +-10 clip
            _agx_lib.set_option("env_step_quad", 0)
            one.step(a)
            _agx_lib.set_option("env_step_quad", 1)
            four.step(a)
            for k in ("robot_state_tensor", "robot_actions", "robot_prev_actions", "robot_euler_angles", "robot_body_linvel",
                      "robot_body_angvel", "robot_vehicle_orientation", "robot_vehicle_linvel"):
                assert torch.equal(one.obs_dict[k], four.obs_dict[k]), (t, k, (one.obs_dict[k] - four.obs_dict[k]).abs().max())
            assert torch.equal(one.task_obs["observations"], four.task_obs["observations"]), t
            assert torch.equal(one.rewards, four.rewards), (t, (one.rewards - four.rewards).abs().max())
            assert torch.equal(one.truncations, four.truncations) and torch.equal(one.terminations, four.terminations), t
            e1, e2 = one.sim_env, four.sim_env
            assert torch.equal(e1.sim_steps, e2.sim_steps), t
            m1, m2 = e1.robot_manager.robot.control_allocator.motor_model, e2.robot_manager.robot.control_allocator.motor_model
            assert torch.equal(m1.thrust_soa, m2.thrust_soa), (t, (m1.thrust_soa - m2.thrust_soa).abs().max())
            w1, w2 = e1.global_tensor_dict.get("robot_wrench_cmd"), e2.global_tensor_dict.get("robot_wrench_cmd")
            if w1 is not None:
                assert torch.equal(w1, w2), t
    finally:
        cfg.episode_len_steps = 500
        if ctl_cfg is not None:
            ctl_cfg.randomize_params = old_rand


@pytest.mark.parametrize("which", ["navigation_task", "lidar_navigation_task"])
def test_four_lanes_per_env_substep_loop_is_bit_identical_to_the_one_lane_kernel(which, monkeypatch):
    """k_env_step_quad_loop<velocity | acceleration> (10 sub-steps, obstacles split over the lanes of the quad, device
    disturbance draws, navigation reward epilogue) against k_env_step<4, CTRL, false, true>: state, derived tensors,
    motors, rewards, flags, position errors equal bit for bit over several episodes (agx_set_option("env_step_quad", 0) selects the
    one-lane kernel)."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config import task_config as tc
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    cfg = getattr(tc, which + "_config")
    old = (cfg.device, cfg.args, cfg.episode_len_steps)
    cfg.device, cfg.args, cfg.episode_len_steps = DEV, {"rng_seed": 77}, 25
    try:
        n = 150
        one = task_registry.make_task(which, seed=31, num_envs=n, headless=True)
        one.reset()  # (make_task seeds the host generator the first reset draws from: reset before the next make_task)
        four = task_registry.make_task(which, seed=31, num_envs=n, headless=True)
        four.reset()
        import ctypes as C

        for t, want in ((one, "0"), (four, "1")):
            _agx_lib.set_option("env_step_quad", int(want))
            env = t.sim_env
            buf = C.create_string_buffer(128)
            env._lib.agx_env_step_kernel(env._params, env._buffers, n, env.num_physics_steps(), env.task_args, buf, 128)
            assert buf.value.decode().startswith("k_env_step_quad_loop<" if want == "1" else "k_env_step<4,"), buf.value
        g = torch.Generator(device=DEV).manual_seed(2)
        A = one.task_config.action_space_dim
        n_crash = 0
        for t in range(70):
            a = torch.rand(n, A, device=DEV, generator=g) * 2 - 1
            _agx_lib.set_option("env_step_quad", 0)
            o1 = one.step(a)
            _agx_lib.set_option("env_step_quad", 1)
            four.step(a)
            n_crash += int(o1[2].sum())
            for k in ("robot_state_tensor", "robot_actions", "robot_prev_actions", "robot_euler_angles", "robot_body_linvel",
                      "robot_body_angvel", "robot_vehicle_orientation", "robot_vehicle_linvel"):
                assert torch.equal(one.obs_dict[k], four.obs_dict[k]), (t, k, (one.obs_dict[k] - four.obs_dict[k]).abs().max())
            do = (one.task_obs["observations"] != four.task_obs["observations"])
            assert not do.any(), (t, do.any(dim=0).nonzero().flatten().tolist()[:20], int(do.any(dim=1).sum()))
            assert torch.equal(one.rewards, four.rewards), (t, (one.rewards - four.rewards).abs().max())
            assert torch.equal(one.truncations, four.truncations) and torch.equal(one.terminations, four.terminations), t
            m1 = one.sim_env.robot_manager.robot.control_allocator.motor_model
            m2 = four.sim_env.robot_manager.robot.control_allocator.motor_model
            assert torch.equal(m1.thrust_soa, m2.thrust_soa), t
            assert torch.equal(one.pos_err_soa, four.pos_err_soa) and torch.equal(one.prev_pos_err_soa, four.prev_pos_err_soa), t
        assert n_crash >= 1  # the obstacle test was exercised
    finally:
        cfg.device, cfg.args, cfg.episode_len_steps = old


def test_config4_octarotor_lidar_task_runs():
    """BASELINE config 4 at small N: base_octarotor + octarotor_velocity_control + 32x512 LiDAR
    (range + segmentation), 10 sub-steps, disturbances on, sync-free."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import navigation_task_config as cfg
    from aerial_gym_simulator_amd.registry.task_registry import task_registry

    old = (cfg.robot_name, cfg.controller_name, cfg.args)
    cfg.device, cfg.robot_name, cfg.controller_name, cfg.args = DEV, "base_octarotor_with_lidar_32x512", "octarotor_velocity_control", {}
    try:
        n = 64
        task = task_registry.make_task("navigation_task", seed=2, num_envs=n, headless=True)
        task.reset()
        a = torch.rand(n, 4, device=DEV) * 2 - 1
        for _ in range(30):
            obs, rew, term, trunc, _ = task.step(a)
        px, seg = task.obs_dict["depth_range_pixels"], task.obs_dict["segmentation_pixels"]
        assert px.shape == (n, 1, 32, 512) and seg.shape == (n, 1, 32, 512)
        assert torch.isfinite(px).all() and torch.isfinite(obs["observations"]).all() and torch.isfinite(rew).all()
        assert float(px.max()) <= 1.0 and float(px.min()) >= -1.0
        hit = (seg != -2).float().mean()
        assert hit > 0.9  # the walls enclose the env
        assert int(task.sim_env.global_tensor_dict["episode_count"].sum()) >= n  # resets happened
    finally:
        cfg.robot_name, cfg.controller_name, cfg.args = old
        cfg.robot_name, cfg.controller_name = "base_quadrotor_with_camera_64x48", "lee_velocity_control"


@pytest.mark.parametrize("which", ["position", "navigation"])
def test_exchange_rows_written_by_the_obs_kernels(which):
    """AgxEnvBuffers.step_rows: the observation kernels also write obs | reward | terminated |
    truncated into the send buffer of the per-step all-gather (bit-identical copies)."""
    import aerial_gym_simulator_amd  # noqa: F401
    from aerial_gym_simulator_amd.config.task_config import navigation_task_config, position_setpoint_task_config
    from aerial_gym_simulator_amd.registry.task_registry import task_registry
    from aerial_gym_simulator_amd.sharding import StepGather

    cfg = position_setpoint_task_config if which == "position" else navigation_task_config
    old = (cfg.episode_len_steps, cfg.args, getattr(cfg, "controller_name", None))
    cfg.device, cfg.episode_len_steps, cfg.args = DEV, 7, {}
    if which == "position":
        cfg.controller_name = "lee_position_control"
    try:
        n = 200
        task = task_registry.make_task(which + ("_setpoint_task" if which == "position" else "_task"), seed=3, num_envs=n, headless=True)
        task.reset()
        d = task.task_obs["observations"].shape[1]
        sg = StepGather(n, d, DEV, env=task.sim_env, reward=task.rewards)
        a = torch.rand(n, 4, device=DEV) * 2 - 1
        seen_trunc = 0
        for step in range(20):
            obs, rew, term, trunc, _ = task.step(a)
            got = sg.unpack(sg.exchange(task.sim_env._parity))  # world size 1: this step's rows
            assert torch.equal(got[0], obs["observations"]) and torch.equal(got[1], rew)
            assert torch.equal(got[2], term.bool()) and torch.equal(got[3], trunc.bool())
            seen_trunc += int(trunc.sum())
            other = sg.rows[task.sim_env._parity ^ 1]
            assert step == 0 or not torch.equal(other, sg.rows[task.sim_env._parity])  # the other parity holds the previous step
        assert seen_trunc >= n  # the flags were exercised
    finally:
        cfg.episode_len_steps, cfg.args = old[0], old[1]
        if old[2] is not None:
            cfg.controller_name = old[2]
