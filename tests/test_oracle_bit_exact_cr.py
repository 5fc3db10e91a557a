"""CPU: the C oracle against the reference's own torch code evaluated with CORRECTLY ROUNDED elementary functions
(tests/golden/cr/*.npz: `python oracle/gen_golden.py --cr`, oracle/cr_torch.py) -- BIT FOR BIT.

torch's sin / cos / atan2 / asin / exp / sqrt kernels are 1-ulp implementations that differ between its own CPU
vector path, its scalar tails and CUDA, so the reference's recorded numbers carry an implementation-dependent last
bit (tests/test_oracle_vs_reference.py gates those at 1e-5 .. 1e-6).  With that one degree of freedom removed --
the reference's code unmodified, the six functions routed through float64 -- every float the dynamics path produces
is reproduced exactly: derived tensors, controller wrench, motor thrusts, net body wrench, rewards, observations,
flags, over single sub-steps of all ten robot / controller pairs and over the 260 / 160 / 1000-step config-1 traces
with resets (no divergence at all: the traces are not chaotic comparisons any more).  What this pins, beyond the
tolerance tests: the order of every operation, the fused multiply-adds inside torch.cross / torch.norm, torch's
`scalar / tensor` = reciprocal * scalar, python-double scalar arithmetic (dt / 6).  The HIP kernels are bit-identical
to the oracle (tests/test_gpu_*.py), hence to these fixtures."""
import numpy as np
import pytest
from conftest import TraceReader, golden_params, load_golden

STEP_CASES = ["quad_position", "quad_velocity", "quad_attitude", "quad_acceleration", "quad_no_control",
              "octarotor_position", "octarotor_velocity", "octarotor_fully_actuated", "quad_rates", "quad_velocity_steering"]


def link_wrench(u, W):
    """sum_j W[:, j] u_j accumulated motor by motor from 0 in float32 (the order of the oracle and the kernels)"""
    acc = np.zeros((u.shape[0], 6), np.float32)
    for j in range(u.shape[1]):
        acc = acc + W[None, :, j] * u[:, j:j + 1]
    return acc


def test_math_helpers_bit_exact(orc):
    g = load_golden("math_utils", cr=True)
    n = g["q"].shape[0]
    assert np.array_equal(orc.quat_mul(g["q"], g["q2"]), g["quat_mul"])
    assert np.array_equal(orc.quat_from_euler(g["e"]), g["quat_from_euler"])
    assert np.array_equal(orc.tf_apply(g["q"], g["t"], g["v"]), g["tf_apply"])
    state = np.zeros((n, 13), np.float32)
    state[:, 3:7], state[:, 7:10], state[:, 10:13] = g["q"], g["v"], g["v"]
    euler, qveh, vveh, vbody, wbody = orc.update_states(state)
    assert np.array_equal(euler, g["ssa_euler"])
    assert np.array_equal(qveh, g["vehicle_quat"])
    assert np.array_equal(vbody, g["quat_rotate_inverse"]) and np.array_equal(wbody, g["quat_rotate_inverse"])


@pytest.mark.parametrize("case", STEP_CASES)
def test_substep_bit_exact(orc, case):
    """BaseMultirotor.step (rows a1-a14) + the oracle integrator: every recorded output, bit for bit."""
    g = load_golden("step_" + case, cr=True)
    pd = golden_params(g)
    P = orc.make_params(pd)
    W = np.array(pd["wrench_map"], np.float32).reshape(6, -1)
    mask = g["application_mask"]
    K = g["state"].shape[0]
    for k in range(K):
        st, th = g["state"][k].copy(), g["thrust_in"][k].copy()
        dist = g["disturb"][k] if g["disturb"][k].any() else None
        o = orc.substep(P, st, g["action"][k], th, g["kT"], g["tau_inc"], g["tau_dec"], g["Kp"], g["Kv"], g["KR"],
                        g["Kw"], disturb=dist, disturb_max=g["disturb_max"], integrate=True)
        for name, got in (("euler", o.euler), ("qveh", o.qveh), ("vveh", o.vveh), ("vbody", o.vbody), ("wbody", o.wbody)):
            assert np.array_equal(got, g[name][k]), (case, k, name)
        if "no_control" not in case:
            assert np.array_equal(o.wrench_cmd, g["wrench_cmd"][k]), (case, k)
        assert np.array_equal(th, g["thrust_out"][k]), (case, k)
        assert np.array_equal(o.action_clipped, g["action_after"][k]), (case, k)
        # net wrench: the motor links' sum + the root link's entry of the reference's force / torque tensors
        bw = link_wrench(g["force"][k][:, mask, 2], W)
        bw[:, 0:3] += g["force"][k][:, 0, :]
        bw[:, 3:6] += g["torque"][k][:, 0, :]
        assert np.array_equal(o.body_wrench, bw), (case, k)
        if k + 1 < K:
            assert np.array_equal(st, g["state"][k + 1]), (case, k)


@pytest.mark.parametrize("case", STEP_CASES)
def test_substep_on_the_ordinary_fixtures_inputs_bit_exact(orc, case):
    """tests/golden/step_<case>.npz (the reference as torch evaluates it) also carries what the same code returns on the
    same recorded inputs with correctly rounded functions (`*_cr`): the oracle reproduces those bit for bit, and the
    distance between the two recordings is the reference's own last-bit freedom on each sample (reported)."""
    from conftest import PARITY

    g = load_golden("step_" + case)
    pd = golden_params(g)
    P = orc.make_params(pd)
    K = g["state"].shape[0]
    for k in range(K):
        st, th = g["state"][k].copy(), g["thrust_in"][k].copy()
        dist = g["disturb"][k] if g["disturb"][k].any() else None
        o = orc.substep(P, st, g["action"][k], th, g["kT"], g["tau_inc"], g["tau_dec"], g["Kp"], g["Kv"], g["KR"],
                        g["Kw"], disturb=dist, disturb_max=g["disturb_max"], integrate=True)
        assert np.array_equal(th, g["thrust_out_cr"][k]), (case, k)
        assert np.array_equal(o.wbody, g["wbody_cr"][k]), (case, k)
        if "no_control" not in case:
            assert np.array_equal(o.wrench_cmd, g["wrench_cmd_cr"][k]), (case, k)
        assert np.array_equal(st, g["state_next_cr"][k]), (case, k)
        if k + 1 < K:
            PARITY.record(f"reference_own_libm_spread_next_state[{case}]", np.abs(g["state"][k + 1] - g["state_next_cr"][k]).max(), None, "abs")
        PARITY.record(f"reference_own_libm_spread_thrust[{case}]", np.abs(g["thrust_out"][k] - g["thrust_out_cr"][k]).max(), None, "abs")


def test_rewards_bit_exact(orc):
    g = load_golden("reward_position", cr=True)
    crashes = g["crashes_in"].astype(np.uint8)
    r = orc.reward_position(g["state"], g["qveh"], g["wbody"], g["target"], crashes)
    assert np.array_equal(r, g["reward"]) and np.array_equal(crashes.astype(bool), g["crashes_out"])
    assert np.array_equal(orc.obs_position(g["state"], g["vbody"], g["wbody"], g["target"]), g["obs"])
    g = load_golden("reward_navigation", cr=True)
    pe = np.ascontiguousarray(g["prev_pos_err"]).copy()
    ppe = np.zeros_like(pe)
    r = orc.reward_navigation(g["state"], g["qveh"], g["target"], g["action"], g["prev_action"],
                              float(g["curriculum_progress"]), g["rp"], pe, ppe, g["crashes"].astype(np.uint8))
    assert np.array_equal(pe, g["pos_err"]) and np.array_equal(ppe, g["prev_pos_err"]) and np.array_equal(r, g["reward"])


@pytest.mark.parametrize("name", ["trace_position_64", "trace_attitude_64", "trace_position_64_long"])
def test_trace_bit_exact(orc, name):
    """BASELINE config 1 (64 envs, empty_env) free-running from the initial reset: the oracle loop against the trace
    assembled from the reference's control / reward / reset code, every step, every env: state, reward, observation,
    crash / truncation flags and reset set identical -- 260, 160 and 1000 steps, resets included."""
    from oracle_env import OraclePositionEnv

    g = load_golden(name, cr=True)
    pd = golden_params(g)
    n = g["init_state"].shape[0]
    ranges = dict(tau_inc=(0.04, 0.04), tau_dec=(0.04, 0.04), thrust=(0.0, 2.0), kT=(0.00000926312, 0.00001826312))
    env = OraclePositionEnv(pd, n, int(g["episode_len"]), (g["Kp"], g["Kv"], g["KR"], g["Kw"]),
                            g["min_init_state"], g["max_init_state"], ranges)
    env.reset_masked(np.ones(n, np.uint8), g["init_u_state"], g["init_u_tau_inc"], g["init_u_tau_dec"],
                     g["init_u_thrust"], g["init_u_kT"])
    assert np.array_equal(env.state, g["init_state"]) and np.array_equal(env.thrust, g["init_thrust"])
    assert np.array_equal(env.kT, g["init_kT"])
    tr = TraceReader(g)
    resets = 0
    for t in range(tr.T):
        obs, rew, crashes, trunc, reset_mask, state_after = env.step(tr.action(t), tr.draws(t))
        assert np.array_equal(rew, g["reward"][t]), (name, t)
        assert np.array_equal(crashes.astype(bool), g["crashes"][t]) and np.array_equal(trunc.astype(bool), g["truncations"][t]), (name, t)
        assert np.array_equal(reset_mask.astype(bool), g["reset_mask"][t]), (name, t)
        sa, ob = tr.kept("state_after_step", t), tr.kept("obs", t)
        if sa is not None:
            assert np.array_equal(state_after, sa), (name, t)
        if ob is not None:
            assert np.array_equal(obs, ob), (name, t)
        resets += int(reset_mask.sum())
    assert resets >= n  # every env was reset at least once on the way
