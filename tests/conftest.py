import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP GPU (MI355X); run with -m gpu")


def load_golden(name, cr=False):
    """cr=True: the fixture of the same name made by the reference's code with correctly rounded elementary functions
    (tests/golden/cr/, oracle/cr_torch.py): the oracle and the kernels reproduce those BIT FOR BIT."""
    return np.load(os.path.join(GOLDEN, *(["cr"] if cr else []), name + ".npz"))


def golden_params(g):
    return json.loads(str(g["params_json"]))


class TraceReader:
    """Uniform access to the dense (trace_*_64.npz) and sparse (trace_*_64_long.npz: int16 actions in 1/256 units,
    reset draws only for steps with a reset, obs / state every 10th step) trace fixtures."""

    DRAWS = ("u_state", "u_tau_inc", "u_tau_dec", "u_thrust", "u_kT")

    def __init__(self, g):
        self.g = g
        self.sparse = "action_q8" in g.files
        self.T = (g["action_q8"] if self.sparse else g["action"]).shape[0]
        if self.sparse:
            self._reset_row = {int(t): i for i, t in enumerate(g["reset_steps"])}
            self._kept_row = {int(t): i for i, t in enumerate(g["kept_steps"])}
            self._act = g["action_q8"].astype(np.float32) / np.float32(256.0)
        else:
            self._act = g["action"]

    def action(self, t):
        return self._act[t]

    def draws(self, t):
        """the five uniform tensors of the reset at step t, or None when no env resets"""
        if self.sparse:
            i = self._reset_row.get(t)
            return None if i is None else tuple(self.g[k][i] for k in self.DRAWS)
        return tuple(self.g[k][t] for k in self.DRAWS) if self.g["reset_mask"][t].any() else None

    def kept(self, name, t):
        """obs / state_after_step of step t, or None if the sparse fixture did not keep it"""
        if self.sparse:
            i = self._kept_row.get(t)
            return None if i is None else self.g[name][i]
        return self.g[name][t]


@pytest.fixture(scope="session")
def orc():
    import oracle

    oracle.lib()
    return oracle


def rel_err(a, b):
    """Blended error max|a - b| / (1 + max|b|): kept for quantities whose scale is O(1) by construction (unit
    quaternions, normalised images) and for the CPU oracle-vs-reference pins; the per-step state gates use
    `max_abs` / `elem_err` below (north_star: fp32 state within 1e-5 per step, absolute)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))


def max_abs(a, b):
    """max over components of |a - b| -- the gate north_star states for the fp32 state (1e-5 per step)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()) if a.size else 0.0


def elem_err(a, b):
    """max over components of |a - b| / max(1, |b|): "fp32 within 1e-5" read element by element -- absolute for
    magnitudes up to 1, relative above (a state component of 12 rad/s has an fp32 resolution of 1e-6 already).
    Unlike `rel_err` nothing is blended across the array."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / np.maximum(1.0, np.abs(b))).max()) if a.size else 0.0


def err_where_reference_is_defined(got, ref, ref_cr, gate):
    """`elem_err(got, ref)` over the elements where the reference's answer is DEFINED to the gate, + what happens elsewhere.

    `ref` is what the reference's code returned on the recorded inputs, `ref_cr` what the same code returns on the same
    inputs with correctly rounded sin / cos / atan2 / asin / exp / sqrt (oracle/cr_torch.py; recorded next to it by the
    generator).  Where the two differ by more than half the gate, the reference's result depends on its math library by
    more than the gate allows (the motor model's sqrt next to zero thrust turns one ulp of a sine into 2e-5 rad/s of body
    rate): no implementation can be within the gate of both, and `got` must then EQUAL the correctly rounded answer.
    Returns (worst error over the defined elements, number of undefined elements, all of those bit-equal to ref_cr)."""
    got, ref, ref_cr = (np.asarray(x, dtype=np.float64) for x in (got, ref, ref_cr))
    scale = np.maximum(1.0, np.abs(ref))
    undefined = np.abs(ref - ref_cr) / scale > 0.5 * gate
    err = np.abs(got - ref) / scale
    worst = float(err[~undefined].max()) if (~undefined).any() else 0.0
    return worst, int(undefined.sum()), bool(np.array_equal(got[undefined], ref_cr[undefined]))


def max_rel(a, b, floor=0.0):
    """max over components of |a - b| / max(|b|, floor): for small-magnitude quantities (kT ~ 1e-5, thrusts)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), floor)
    ok = den > 0
    return float((np.abs(a - b)[ok] / den[ok]).max()) if ok.any() else 0.0


class ParityLog:
    """Collects the measured maxima behind the gates so that they can be printed with the test run and echoed by
    bench.py (`"parity"` key): name -> {"max": worst value seen, "gate": bound asserted, "unit": ...}."""

    def __init__(self):
        self.rows = {}

    def record(self, name, value, gate=None, unit="abs"):
        r = self.rows.setdefault(name, {"max": 0.0, "gate": gate, "unit": unit, "n": 0})
        r["max"] = max(r["max"], float(value))
        r["n"] += 1
        return value

    def check(self, name, value, gate, unit="abs", ctx=None):
        self.record(name, value, gate, unit)
        if os.environ.get("AGX_PARITY_SOFT") == "1":  # measurement runs (profiles/parity_variants.py): record everything
            return
        assert value <= gate, (name, value, gate, ctx)


PARITY = ParityLog()


@pytest.fixture(scope="session")
def parity():
    return PARITY


def pytest_terminal_summary(terminalreporter):
    if not PARITY.rows:
        return
    terminalreporter.write_sep("-", "measured parity maxima (value / gate)")
    for k in sorted(PARITY.rows):
        r = PARITY.rows[k]
        terminalreporter.write_line(f"  {k:58s} {r['max']:.3e} / {r['gate'] if r['gate'] is not None else '-'}  [{r['unit']}, {r['n']} checks]")
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.environ.get("AGX_PARITY_REPORT", os.path.join(out, "parity_report.json")), "w") as f:
            json.dump(PARITY.rows, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(autouse=True)
def _library_options_back_to_default():
    """tests flip the library's process-wide A/B options (agx_set_option): whatever a test leaves behind is undone"""
    yield
    from aerial_gym_simulator_amd import _lib

    if _lib._lib is not None:
        _lib.set_option("env_step_quad", 1)
        _lib.set_option("ray_split", 0)


@pytest.fixture(autouse=True)
def _seed_everything(request):
    """Scene construction uses python `random` (asset shuffle, like the reference) and torch's global generator:
    seed them per test so a failure can be reproduced."""
    import random
    import zlib

    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    try:
        import torch

        torch.manual_seed(seed)
    except ImportError:
        pass
    yield
