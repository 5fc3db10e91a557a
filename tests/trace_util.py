"""Test helper shared by the CPU (oracle) and GPU (Task API) config-1 trace tests."""
import numpy as np

# (fixture, steps over which the SURVEY 8d gate "reward trace <= 1e-4" is asserted, looser bound for the rest)
TRACES = {"position": ("trace_position_64", 60, 1e-3), "attitude": ("trace_attitude_64", 60, 1e-3),
          "position_long": ("trace_position_64_long", 500, None)}


def run_trace_against(step_fn, g, gate_steps, tail_gate, label):
    """Shared by the CPU (oracle) and GPU (Task API) trace tests.  step_fn(t, action, draws) -> (obs, reward,
    crashes, truncations) of step t, free-running.  Asserts, for t < gate_steps: flags bit-exact, reward <= 1e-4
    absolute (rewards are O(1..30)), obs <= 1e-3; afterwards the trace is only followed until its first
    flag mismatch and the first step whose reward error exceeds 1e-4 is REPORTED ("or first-divergence step")."""
    from conftest import PARITY, TraceReader, max_abs

    tr = TraceReader(g)
    worst_gate = worst_tail = worst_obs = 0.0
    first_div, stopped = None, None
    for t in range(tr.T):
        obs, rew, crashes, trunc = step_fn(t, tr.action(t), tr.draws(t))
        flags_ok = (np.array_equal(np.asarray(crashes).astype(bool), g["crashes"][t])
                    and np.array_equal(np.asarray(trunc).astype(bool), g["truncations"][t]))
        if t < gate_steps:
            assert flags_ok, (label, t)
        elif not flags_ok:
            stopped = t  # a diverged trajectory crashed / timed out on a different step: nothing left to compare
            break
        e = max_abs(rew, g["reward"][t])
        if e > 1e-4 and first_div is None:
            first_div = t
        ref_o = tr.kept("obs", t)
        if t < gate_steps:
            worst_gate = max(worst_gate, e)
            if ref_o is not None:
                worst_obs = max(worst_obs, max_abs(obs, ref_o))
        else:
            worst_tail = max(worst_tail, e)
    PARITY.record(f"{label}/reward_abs_err_first_{gate_steps}_steps", worst_gate, 1e-4)
    PARITY.record(f"{label}/obs_abs_err_first_{gate_steps}_steps", worst_obs, 1e-3)
    PARITY.record(f"{label}/reward_abs_err_after_step_{gate_steps}", worst_tail, tail_gate)
    PARITY.record(f"{label}/first_divergence_step(reward_err>1e-4;{tr.T}=none)", tr.T if first_div is None else first_div, None, "step")
    print(f"{label}: reward err over the first {gate_steps} steps {worst_gate:.2e} (gate 1e-4), obs {worst_obs:.2e}; afterwards "
          f"{worst_tail:.2e}; first step with reward err > 1e-4: {first_div}; flag mismatch at: {stopped}; {tr.T} steps")
    assert worst_gate <= 1e-4, (label, worst_gate)
    assert worst_obs <= 1e-3, (label, worst_obs)
    assert first_div is None or first_div >= gate_steps
    if tail_gate is not None:
        assert stopped is None and worst_tail <= tail_gate, (label, stopped, worst_tail)
    return first_div
