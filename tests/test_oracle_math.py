"""CPU: the elementary functions shared (as two copies of the same explicit kernels) by the oracle and the HIP
dynamics kernels -- oracle/oracle_math.h, csrc/agx_device_math.h -- against libm evaluated in double.  They replace
sinf / cosf / atan2f / asinf / expf so that oracle and GPU agree bit for bit; here their ACCURACY is pinned: within
2.5 ulp of the correctly rounded result on the ranges the dynamics path uses (torch's own CPU / CUDA kernels are
1-ulp implementations that do not agree with each other bit for bit either)."""
import numpy as np


def ulp_err(got, exact):
    exact32 = exact.astype(np.float32)
    ulp = np.spacing(np.abs(exact32)).astype(np.float64)
    ulp = np.maximum(ulp, np.spacing(np.float32(1e-30)))
    return np.abs(got.astype(np.float64) - exact) / ulp


def test_sincos(orc):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-10.5, 10.5, 200000), rng.uniform(-0.6, 0.6, 100000), rng.uniform(-64, 64, 50000),
                        np.array([0.0, np.pi / 2, np.pi, -np.pi, 1e-8, -1e-8])]).astype(np.float32)
    for name, fn in (("sin", np.sin), ("cos", np.cos)):
        got = orc.math_eval(name, x)
        exact = fn(x.astype(np.float64))
        assert np.abs(got - exact).max() < 1.3e-7, name  # absolute: what the quaternion / rotation code sees
        big = np.abs(exact) > 0.1
        assert ulp_err(got[big], exact[big]).max() < 2.5, name


def test_atan2_asin(orc):
    rng = np.random.default_rng(1)
    y = rng.normal(size=400000).astype(np.float32) * rng.choice([1e-3, 1.0, 50.0], 400000).astype(np.float32)
    x = rng.normal(size=400000).astype(np.float32)
    got = orc.math_eval("atan2", y, x)
    exact = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    d = np.abs(got - exact)
    assert np.minimum(d, 2 * np.pi - d).max() < 4.0e-7  # <= 2 ulp of pi
    small = np.abs(exact) < 1.0
    assert ulp_err(got[small], exact[small]).max() < 3.0
    # quadrant / axis cases: torch.atan2 conventions
    ys = np.array([0, 0, 1, -1, 0, 1, -1], np.float32)
    xs = np.array([0, 1, 0, 0, -1, -1, -1], np.float32)
    ref = np.arctan2(ys.astype(np.float64), xs.astype(np.float64))
    assert np.abs(orc.math_eval("atan2", ys, xs) - ref).max() < 3e-7
    a = np.concatenate([rng.uniform(-1, 1, 300000), np.array([0.0, 0.5, -0.5, 1.0, -1.0, 0.99999994, 1e-5])]).astype(np.float32)
    got = orc.math_eval("asin", a)
    exact = np.arcsin(a.astype(np.float64))
    assert ulp_err(got, exact).max() < 3.0
    assert np.abs(got - exact).max() < 2.5e-7


def test_exp(orc):
    rng = np.random.default_rng(2)
    x = np.concatenate([-rng.uniform(0, 90, 300000), rng.uniform(0, 5, 50000), np.array([0.0, -1e-8, -87.0, -100.0, -1e4])]).astype(np.float32)
    got = orc.math_eval("exp", x)
    exact = np.exp(x.astype(np.float64))
    normal = exact > 1.2e-38
    assert ulp_err(got[normal], exact[normal]).max() < 2.5
    assert np.all(got[~normal] < 1.2e-38)  # flushed or subnormal: |error| < 1.2e-38
    assert got[np.argmax(x == 0.0)] == 1.0
