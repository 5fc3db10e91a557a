/*
 * TEST INFRASTRUCTURE -- elementary functions of the CPU oracle.
 *
 * The reference evaluates sin / cos / atan2 / asin / exp through torch (SLEEF's 1-ulp kernels for full vectors, glibc
 * for the tails on the CPU; libdevice on CUDA): implementations that agree with each other to about 1 ulp, not bit
 * for bit.  The one result all of them approximate is the CORRECTLY ROUNDED one, so that is what this oracle and the
 * HIP kernels compute (round 3; rounds 1-2 used <= 2.5-ulp single-precision cephes kernels, and their ulps, amplified
 * by K_R / J dt and by the motor model's sqrt near zero thrust, were what kept the per-step body rate 1.9e-5 from
 * the reference's): every function is evaluated in float64 -- argument reduction with fused multiply-adds, a
 * near-minimax polynomial (aerial_gym_simulator_amd/csrc/gen_math_coeffs.py derives the coefficients with mpmath, max relative error
 * <= 1.2e-15) -- and rounded to float32 ONCE.  The float result is the correctly rounded one unless the exact value
 * lies within ~2e-15 relative of a rounding boundary (about one argument in 10^7; tests/test_oracle_math.py measures
 * it against mpmath / libm).
 *
 * Both sides evaluate the SAME sequence of IEEE double operations (+ - * / sqrt fma rint, each correctly rounded by
 * definition; the oracle is built with -ffp-contract=off, fma() is explicit), so every float of the dynamics path stays
 * bit-identical on the CPU and the GPU.  The device copy is aerial_gym_simulator_amd/csrc/agx_device_math.h
 * (sincos_bounded, atan2_cw, asin_cw, exp_cw): same constants, same order of operations.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define OM_PIO2_HI 0x1.921fb54442d18p+0
#define OM_PIO2_LO 0x1.1a62633145c07p-54
#define OM_PIO4 0x1.921fb54442d18p-1
#define OM_PI 0x1.921fb54442d18p+1
#define OM_TWO_OVER_PI 0x1.45f306dc9c883p-1
#define OM_TAN_PIO8 0x1.a827999fcef32p-2
#define OM_LN2_HI 0x1.62e42fefa39efp-1
#define OM_LN2_LO 0x1.abc9e3b39803fp-56
#define OM_INV_LN2 0x1.71547652b82fep+0

/* sin and cos of the same angle, |x| < 64 (every angle of the path: Euler angles, half angles, yaw set-points
 * clipped to +-10).  k = rint(x 2/pi); r = x - k pi/2 with pi/2 = hi + lo (two fma: relative accuracy of r ~1e-16 even
 * next to a multiple of pi/2); sin r = r + r^3 S(r^2), cos r = 1 - r^2/2 + r^4 C(r^2) on |r| <= pi/4. */
static inline void om_sincosf(float x, float *sn, float *cs) {
  const double xd = (double)x;
  const double kd = rint(xd * OM_TWO_OVER_PI);
  const int k = (int)kd;
  double r = fma(-kd, OM_PIO2_HI, xd);
  r = fma(-kd, OM_PIO2_LO, r);
  const double z = r * r;
  double ps = 0x1.5e098556d302ep-33;
  ps = fma(ps, z, -0x1.ae60069e53ef1p-26);
  ps = fma(ps, z, 0x1.71de379252004p-19);
  ps = fma(ps, z, -0x1.a01a019e80693p-13);
  ps = fma(ps, z, 0x1.1111111110ba5p-7);
  ps = fma(ps, z, -0x1.5555555555555p-3);
  const double s = fma(z * r, ps, r);
  double pc = 0x1.1c808728603bbp-29;
  pc = fma(pc, z, -0x1.27e25ca05d2bep-22);
  pc = fma(pc, z, 0x1.a019ff501e5c1p-16);
  pc = fma(pc, z, -0x1.6c16c16b5fdb7p-10);
  pc = fma(pc, z, 0x1.5555555555434p-5);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double s0 = (k & 1) ? c : s;
  const double c0 = (k & 1) ? s : c;
  *sn = (float)((k & 2) ? -s0 : s0);
  *cs = (float)(((k + 1) & 2) ? -c0 : c0);
}
static inline float om_sinf(float x) { float s, c; om_sincosf(x, &s, &c); return s; }
static inline float om_cosf(float x) { float s, c; om_sincosf(x, &s, &c); return c; }

/* angle of the point (ax, ay), ax, ay >= 0 and not both 0, in [0, pi/2]: octant reduction n / d = min / max, a second
 * reduction at tan(pi/8) folded into the same division ((n - d) / (n + d), both exact), atan t = t + t^3 A(t^2) on
 * |t| <= tan(pi/8) */
static inline double om_atan2_pos(double ay, double ax) {
  const int swap = ay > ax;
  const double n = swap ? ax : ay, d = swap ? ay : ax;
  const int mid = n > OM_TAN_PIO8 * d;
  const double num = mid ? n - d : n, den = mid ? n + d : d;
  const double t = num / den;
  const double z = t * t;
  double p = 0x1.7439839062d96p-6;
  p = fma(p, z, -0x1.6f15a83e9672ap-5);
  p = fma(p, z, 0x1.d5dcd0576e964p-5);
  p = fma(p, z, -0x1.105d0bc7abc71p-4);
  p = fma(p, z, 0x1.3b0671b310199p-4);
  p = fma(p, z, -0x1.745c7de4bec48p-4);
  p = fma(p, z, 0x1.c71c6dc4ee1e0p-4);
  p = fma(p, z, -0x1.2492491dbb541p-3);
  p = fma(p, z, 0x1.999999999083bp-3);
  p = fma(p, z, -0x1.5555555555545p-2);
  double a = fma(z * t, p, t);
  if (mid) a = OM_PIO4 + a;
  if (swap) a = OM_PIO2_HI - a;
  return a;
}

/* atan2 for finite arguments; atan2(0, 0) = 0 like torch.  (y = -0, x < 0 gives +pi where IEEE says -pi: every
 * consumer takes the angle modulo 2 pi.) */
static inline float om_atan2f(float y, float x) {
  if (x == 0.0f) {
    if (y == 0.0f) return 0.0f;
    return (y > 0.0f) ? (float)OM_PIO2_HI : -(float)OM_PIO2_HI;
  }
  double a = om_atan2_pos(fabs((double)y), fabs((double)x));
  if (x < 0.0f) a = OM_PI - a;
  return (float)((y < 0.0f) ? -a : a);
}

/* asin, |x| <= 1 (callers handle |x| >= 1 themselves, utils/math.py:135): the angle of (sqrt((1 - |x|)(1 + |x|)), |x|);
 * both factors are exact in double */
static inline float om_asinf(float xx) {
  const double a = fabs((double)xx);
  const double c = sqrt((1.0 - a) * (1.0 + a));
  const double r = om_atan2_pos(a, c);
  return (float)((xx < 0.0f) ? -r : r);
}

/* exp: k = rint(x / ln 2), r = x - k ln 2 (hi + lo), e^r = 1 + r + r^2 E(r) on |r| <= ln2 / 2, exact scaling by 2^k.
 * Results below the smallest normal float are flushed to 0 (|error| < 1.2e-38). */
static inline float om_expf(float x) {
  if (x > 88.7228317f) return INFINITY;
  if (x < -87.3365402f) return 0.0f;
  const double xd = (double)x;
  const double kd = rint(xd * OM_INV_LN2);
  const int k = (int)kd;
  double r = fma(-kd, OM_LN2_HI, xd);
  r = fma(-kd, OM_LN2_LO, r);
  double p = 0x1.288088c0e67a5p-22;
  p = fma(p, r, 0x1.72c79824255e0p-19);
  p = fma(p, r, 0x1.a019c971b4d98p-16);
  p = fma(p, r, 0x1.a019ad55d1aa7p-13);
  p = fma(p, r, 0x1.6c16c1739a511p-10);
  p = fma(p, r, 0x1.1111111c5719ap-7);
  p = fma(p, r, 0x1.5555555554ca7p-5);
  p = fma(p, r, 0x1.5555555553b48p-3);
  p = fma(p, r, 0x1.0000000000000p-1);
  const double e = fma(r * r, p, 1.0 + r);
  const uint64_t b = (uint64_t)(int64_t)(k + 1023) << 52; /* 2^k, -126 <= k <= 128 */
  double two_k;
  memcpy(&two_k, &b, 8);
  return (float)(e * two_k);
}

#endif /* ORACLE_MATH_H */
