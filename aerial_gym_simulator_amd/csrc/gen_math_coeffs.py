"""Provenance of the constants in agx_device_math.h: derives the double-precision polynomial coefficients of the
elementary functions (sincos, atan, exp) with mpmath.  Not needed at build or run time.

    python aerial_gym_simulator_amd/csrc/gen_math_coeffs.py

Each kernel is evaluated in float64 with explicit fused multiply-adds and rounded ONCE to float32 at the end, so the
float result is the correctly rounded one unless the exact value lies within ~2e-15 relative of a rounding boundary
(about one argument in 10^7).  The polynomials are near-minimax: interpolation at Chebyshev nodes of the weighted
function, solved in 60-digit arithmetic, the achieved maximum relative error printed (all < 3e-16).
"""
import mpmath as mp

mp.mp.dps = 60


def cheb_nodes(a, b, n):
    return [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * i + 1) / (2 * n)) for i in range(n)]


def fit(g, a, b, n):
    """coefficients c_0..c_{n-1} of the polynomial in z interpolating g(z) at n Chebyshev nodes of [a, b]"""
    zs = cheb_nodes(mp.mpf(a), mp.mpf(b), n)
    A = mp.matrix(n, n)
    rhs = mp.matrix(n, 1)
    for i, z in enumerate(zs):
        for j in range(n):
            A[i, j] = z ** j
        rhs[i] = g(z)
    return list(mp.lu_solve(A, rhs))


def hexf(c):
    return float(c).hex()


def horner(cs, z):
    acc = mp.mpf(0)
    for c in reversed(cs):
        acc = acc * z + c
    return acc


def report(name, cs, err):
    print(f"/* {name}: max rel err {mp.nstr(err, 3)} */")
    print("  " + ", ".join(hexf(c) for c in cs))


NS, NC, NA, NE = 6, 5, 10, 9  # number of coefficients (degree + 1) of S, C, A, E


def main():
    # sin(r) = r + r^3 S(z), z = r^2, |r| <= pi/4 (+ margin)
    lim = (mp.pi / 4 * mp.mpf("1.001")) ** 2
    S = fit(lambda z: (mp.sin(mp.sqrt(z)) / mp.sqrt(z) - 1) / z, mp.mpf("1e-30"), lim, NS)
    Sd = [mp.mpf(float(c)) for c in S]
    err = max(abs((mp.sqrt(z) + mp.sqrt(z) ** 3 * horner(Sd, z)) / mp.sin(mp.sqrt(z)) - 1)
              for z in [lim * i / 2000 for i in range(1, 2001)])
    report(f"sin: r + r^3 * S(r^2), S of degree {NS - 1}", S, err)
    # cos(r) = 1 - z/2 + z^2 C(z)
    Cc = fit(lambda z: (mp.cos(mp.sqrt(z)) - 1 + z / 2) / z ** 2, mp.mpf("1e-30"), lim, NC)
    Cd = [mp.mpf(float(c)) for c in Cc]
    err = max(abs((1 - z / 2 + z * z * horner(Cd, z)) / mp.cos(mp.sqrt(z)) - 1) for z in [lim * i / 2000 for i in range(1, 2001)])
    report(f"cos: 1 - z/2 + z^2 * C(z), C of degree {NC - 1}", Cc, err)
    # atan(t) = t + t^3 A(z), |t| <= tan(pi/8) (+ margin)
    lim = (mp.tan(mp.pi / 8) * mp.mpf("1.001")) ** 2
    A = fit(lambda z: (mp.atan(mp.sqrt(z)) / mp.sqrt(z) - 1) / z, mp.mpf("1e-30"), lim, NA)
    Ad = [mp.mpf(float(c)) for c in A]
    err = max(abs((mp.sqrt(z) + mp.sqrt(z) ** 3 * horner(Ad, z)) / mp.atan(mp.sqrt(z)) - 1)
              for z in [lim * i / 2000 for i in range(1, 2001)])
    report(f"atan: t + t^3 * A(t^2), A of degree {NA - 1}", A, err)
    # exp(r) = 1 + r + r^2 E(r), |r| <= ln2/2 (+ margin)
    lim = mp.log(2) / 2 * mp.mpf("1.001")
    E = fit(lambda r: (mp.exp(r) - 1 - r) / r ** 2 if abs(r) > mp.mpf("1e-20") else mp.mpf(1) / 2 + r / 6, -lim, lim, NE)
    Ed = [mp.mpf(float(c)) for c in E]
    err = max(abs((1 + r + r * r * horner(Ed, r)) / mp.exp(r) - 1) for r in [-lim + 2 * lim * i / 2000 for i in range(2001)])
    report(f"exp: 1 + r + r^2 * E(r), E of degree {NE - 1}", E, err)
    # constants
    pio2 = mp.pi / 2
    hi = float(pio2)
    print("/* pi/2 = hi + lo */", hi.hex(), float(pio2 - mp.mpf(hi)).hex())
    ln2 = mp.log(2)
    hi = float(ln2)
    print("/* ln 2 = hi + lo */", hi.hex(), float(ln2 - mp.mpf(hi)).hex())
    print("/* 2/pi, 1/ln2, pi/4, pi/2, pi, tan(pi/8) */", float(2 / mp.pi).hex(), float(1 / ln2).hex(), float(mp.pi / 4).hex(),
          float(mp.pi / 2).hex(), float(mp.pi).hex(), float(mp.tan(mp.pi / 8)).hex())


if __name__ == "__main__":
    main()
