"""CPU: the elementary functions shared (as two copies of the same explicit kernels) by the oracle and the HIP
dynamics kernels -- oracle/oracle_math.h, csrc/agx_device_math.h -- against libm evaluated in double.  They replace
sinf / cosf / atan2f / asinf / expf so that oracle and GPU agree bit for bit; here their ACCURACY is pinned: since
round 3 they are evaluated in float64 and rounded once, i.e. CORRECTLY ROUNDED in practice -- the error is <= 0.5 ulp
(+ 1e-6 ulp of evaluation error) everywhere on the ranges the dynamics path uses and the result equals the float
nearest to the exact value for all but a handful of arguments per 10^7 (torch's own CPU / CUDA kernels are 1-ulp
implementations that do not agree with each other bit for bit; the correctly rounded value is what they all approximate)."""
import numpy as np

HALF_ULP = 0.5 + 1e-5  # libm in double is itself within 1 double-ulp (2e-9 float ulp) of the exact value


def ulp_err(got, exact):
    exact32 = exact.astype(np.float32)
    ulp = np.spacing(np.abs(exact32)).astype(np.float64)
    ulp = np.maximum(ulp, np.spacing(np.float32(1e-30)))
    return np.abs(got.astype(np.float64) - exact) / ulp


def _correctly_rounded_fraction(got, exact):
    return float(np.mean(got == exact.astype(np.float32)))


def test_sincos(orc):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-10.5, 10.5, 400000), rng.uniform(-0.6, 0.6, 200000), rng.uniform(-64, 64, 100000),
                        np.array([0.0, np.pi / 2, np.pi, -np.pi, 1e-8, -1e-8, 1.5707964, 3.1415927, 6.2831855, 47.12389])]).astype(np.float32)
    for name, fn in (("sin", np.sin), ("cos", np.cos)):
        got = orc.math_eval(name, x)
        exact = fn(x.astype(np.float64))
        assert ulp_err(got, exact).max() <= HALF_ULP, name
        assert _correctly_rounded_fraction(got, exact) > 0.99999, name
    assert orc.math_eval("sin", np.zeros(1, np.float32))[0] == 0.0 and orc.math_eval("cos", np.zeros(1, np.float32))[0] == 1.0


def test_atan2_asin(orc):
    rng = np.random.default_rng(1)
    y = rng.normal(size=800000).astype(np.float32) * rng.choice([1e-3, 1.0, 50.0], 800000).astype(np.float32)
    x = rng.normal(size=800000).astype(np.float32)
    got = orc.math_eval("atan2", y, x)
    exact = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert ulp_err(got, exact).max() <= HALF_ULP
    assert _correctly_rounded_fraction(got, exact) > 0.99999
    # quadrant / axis cases: torch.atan2 conventions (atan2(0, 0) = 0; y = +-0 with x < 0 gives +pi: angles are used mod 2 pi)
    ys = np.array([0, 0, 1, -1, 0, 1, -1, -0.0, 1, -1], np.float32)
    xs = np.array([0, 1, 0, 0, -1, -1, -1, -1, 1, 1], np.float32)
    ref = np.arctan2(np.abs(ys).astype(np.float64) * np.where(ys < 0, -1, 1), xs.astype(np.float64)).astype(np.float32)
    assert np.array_equal(orc.math_eval("atan2", ys, xs), ref)
    a = np.concatenate([rng.uniform(-1, 1, 600000), rng.uniform(-1e-3, 1e-3, 100000),
                        np.array([0.0, 0.5, -0.5, 1.0, -1.0, 0.99999994, -0.99999994, 1e-5, 1e-12])]).astype(np.float32)
    got = orc.math_eval("asin", a)
    exact = np.arcsin(a.astype(np.float64))
    assert ulp_err(got, exact).max() <= HALF_ULP
    assert _correctly_rounded_fraction(got, exact) > 0.99999


def test_exp(orc):
    rng = np.random.default_rng(2)
    x = np.concatenate([-rng.uniform(0, 90, 600000), rng.uniform(0, 5, 100000), rng.uniform(-1e-3, 1e-3, 100000),
                        np.array([0.0, -1e-8, -87.0, -87.33654, -100.0, -1e4, 88.0, 88.72283])]).astype(np.float32)
    got = orc.math_eval("exp", x)
    exact = np.exp(x.astype(np.float64))
    normal = exact > 1.2e-38
    assert ulp_err(got[normal], exact[normal]).max() <= HALF_ULP
    assert _correctly_rounded_fraction(got[normal], exact[normal]) > 0.99999
    assert np.all(got[~normal] < 1.2e-38)  # flushed or subnormal: |error| < 1.2e-38
    assert got[np.argmax(x == 0.0)] == 1.0
    assert np.isinf(orc.math_eval("exp", np.array([89.0], np.float32))[0])


def test_against_mpmath_on_hard_cases(orc):
    """arguments next to multiples of pi / 2 (worst cases of the range reduction) and next to 1 for asin, checked against
    50-digit arithmetic instead of libm"""
    import mpmath as mp

    mp.mp.dps = 50
    k = np.arange(1, 41)
    xs = np.concatenate([np.float32(k * np.pi / 2), np.nextafter(np.float32(k * np.pi / 2), np.float32(100)),
                         np.nextafter(np.float32(k * np.pi / 2), np.float32(-100))]).astype(np.float32)
    xs = np.concatenate([xs, -xs])
    s, c = orc.math_eval("sin", xs), orc.math_eval("cos", xs)
    for i, x in enumerate(xs):
        xe = mp.mpf(float(x))
        assert s[i] == np.float32(float(mp.sin(xe))), (x, "sin")
        assert c[i] == np.float32(float(mp.cos(xe))), (x, "cos")
    a = np.array([1 - 2.0 ** -e for e in range(1, 25)] + [2.0 ** -e for e in range(1, 40)], np.float32)
    got = orc.math_eval("asin", a)
    for i, x in enumerate(a):
        assert got[i] == np.float32(float(mp.asin(mp.mpf(float(x))))), (x, "asin")
