// Device-side vector / quaternion helpers for the gfx950 kernels.
//
// Everything is register-resident value types (no arrays indexed at run time, so
// nothing spills to scratch).  Quaternions are xyzw.  The formulas are the ones the
// reference evaluates in aerial_gym/utils/math.py (file:line cited per function) so
// that fp32 results track the reference to rounding; the library is compiled with
// -ffp-contract=off, so every + - * / sqrt is one correctly rounded IEEE operation
// (this is what makes crash flags, segmentation ids and depth bit-reproducible on
// the CPU reference used by the parity tests).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agx {

struct V3 {
  float x, y, z;
};
struct Q4 {
  float x, y, z, w;
};
struct M33 {
  float m00, m01, m02, m10, m11, m12, m20, m21, m22;
};

#define AGX_DEV __device__ __forceinline__

constexpr float kPi = 3.14159274101257324f;     // float(torch.pi)
constexpr float kTwoPi = 6.28318548202514648f;  // float(2 * torch.pi)

AGX_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
AGX_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
AGX_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
AGX_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
AGX_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
AGX_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
AGX_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// torch.cross: ATen evaluates a_j b_k - a_k b_j inside one compiled expression, which the compiler contracts into a fused
// multiply-subtract, fma(a_j, b_k, -(a_k b_j)) -- the second product is rounded, the first is not (bit-equal to torch on
// 10^5 random vectors; nvcc's default fmad gives the same form).  Every cross product of the dynamics path is a torch.cross
// in the reference (utils/math.py:63,319-346, base_lee_controller.py:144,181-183).  Sums of separately written torch ops
// (a * b + c as two calls) round separately: that is `dot` and everything else here (contraction is off).
AGX_DEV V3 cross(V3 a, V3 b) {
  return V3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}
// the plain three-rounding form (geometry that is not torch's: scene normals, the ray-cast's Warp restatement)
AGX_DEV V3 cross_plain(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// torch.norm(x, dim) of a 3-vector: the reduction kernel accumulates acc = fma(x_k, x_k, acc) from x_0^2 (bit-equal on
// 10^5 random vectors)
AGX_DEV float norm(V3 a) { return sqrtf(__builtin_fmaf(a.z, a.z, __builtin_fmaf(a.y, a.y, a.x * a.x))); }
AGX_DEV V3 qvec(Q4 q) { return V3{q.x, q.y, q.z}; }
AGX_DEV Q4 conj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }

// torch `%` with a positive modulus (remainder, sign of divisor) -- utils/math.py:144-152
// fmodf(a, m) is exact by definition; for |a| < 2m it is a, a - m or a + m -- and those subtractions are exact too
// (Sterbenz: m <= |a| < 2m) -- so the angles of this path (|a| < 4 pi) take three instructions instead of ocml's
// general remainder loop (~40): same bits.
AGX_DEV float fmod_exact(float a, float m) {
  const float aa = fabsf(a);
  if (aa < m) return a;
  if (aa < 2.0f * m) return a < 0.0f ? a + m : a - m;
  return fmodf(a, m);
}
AGX_DEV float pymod(float a, float m) {
  float r = fmod_exact(a, m);
  if (r != 0.0f && (r < 0.0f)) r += m;
  return r;
}
AGX_DEV float ssa(float a) { return pymod(a + kPi, kTwoPi) - kPi; }  // utils/math.py:150-152

// utils/math.py:329-336  v(2w^2-1) + 2w(q x v) + 2q(q.v)
AGX_DEV V3 quat_rotate(Q4 q, V3 v) {
  float s = 2.0f * (q.w * q.w) - 1.0f;
  V3 c = cross(qvec(q), v);
  float d = dot(qvec(q), v);
  return V3{v.x * s + c.x * q.w * 2.0f + q.x * d * 2.0f, v.y * s + c.y * q.w * 2.0f + q.y * d * 2.0f,
            v.z * s + c.z * q.w * 2.0f + q.z * d * 2.0f};
}
// utils/math.py:340-347  a - b + c
AGX_DEV V3 quat_rotate_inverse(Q4 q, V3 v) {
  float s = 2.0f * (q.w * q.w) - 1.0f;
  V3 c = cross(qvec(q), v);
  float d = dot(qvec(q), v);
  return V3{v.x * s - c.x * q.w * 2.0f + q.x * d * 2.0f, v.y * s - c.y * q.w * 2.0f + q.y * d * 2.0f,
            v.z * s - c.z * q.w * 2.0f + q.z * d * 2.0f};
}
// utils/math.py:314-320  t = 2 (q x v); v + w t + q x t
AGX_DEV V3 quat_apply(Q4 q, V3 v) {
  V3 t = cross(qvec(q), v) * 2.0f;
  V3 u = cross(qvec(q), t);
  return V3{v.x + q.w * t.x + u.x, v.y + q.w * t.y + u.y, v.z + q.w * t.z + u.z};
}
// utils/math.py:375-376
AGX_DEV V3 tf_apply(Q4 q, V3 t, V3 v) {
  V3 r = quat_apply(q, v);
  return V3{r.x + t.x, r.y + t.y, r.z + t.z};
}
// utils/math.py:243-263
AGX_DEV Q4 quat_mul(Q4 a, Q4 b) {
  float ww = (a.z + a.x) * (b.x + b.y);
  float yy = (a.w - a.y) * (b.w + b.z);
  float zz = (a.w + a.y) * (b.w - b.z);
  float xx = ww + yy + zz;
  float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
  Q4 o;
  o.w = qq - ww + (a.z - a.y) * (b.y - b.z);
  o.x = qq - xx + (a.x + a.w) * (b.x + b.w);
  o.y = qq - yy + (a.w - a.x) * (b.y + b.z);
  o.z = qq - zz + (a.z + a.y) * (b.w - b.x);
  return o;
}
// utils/math.py:267-293
AGX_DEV M33 quat_to_rotmat(Q4 q) {
  float xx = q.x * q.x, xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
  float yy = q.y * q.y, yz = q.y * q.z, yw = q.y * q.w;
  float zz = q.z * q.z, zw = q.z * q.w;
  M33 m;
  m.m00 = 1.0f - 2.0f * (yy + zz);
  m.m01 = 2.0f * (xy - zw);
  m.m02 = 2.0f * (xz + yw);
  m.m10 = 2.0f * (xy + zw);
  m.m11 = 1.0f - 2.0f * (xx + zz);
  m.m12 = 2.0f * (yz - xw);
  m.m20 = 2.0f * (xz - yw);
  m.m21 = 2.0f * (yz + xw);
  m.m22 = 1.0f - 2.0f * (xx + yy);
  return m;
}
// ---- elementary functions -------------------------------------------------------------------------
// sincos_bounded / atan2_cw / asin_cw / exp_cw: evaluated in float64 (fp64 vector ops run at half the fp32 rate on
// gfx950) and rounded to float32 ONCE, i.e. correctly rounded in practice (the float result differs from the nearest
// float only when the exact value lies within ~2e-15 relative of a rounding boundary).  torch's sin / atan2 / exp
// (SLEEF, glibc, libdevice) are 1-ulp implementations that differ among themselves; the correctly rounded value is the
// one they all approximate, and with it the per-step body rates stay within 1e-5 of the reference's recorded ones
// (DESIGN.md "numerics"; rounds 1-2 used <= 2.5-ulp fp32 cephes kernels: 1.9e-5).  Argument reduction with explicit
// fused multiply-adds, near-minimax polynomials (coefficients: gen_math_coeffs.py next to this file).  The CPU restatement the
// parity tests check against evaluates the SAME sequence of IEEE double operations, so with contraction off the
// dynamics path is bit-reproducible there.
constexpr double kPio2Hi = 0x1.921fb54442d18p+0, kPio2Lo = 0x1.1a62633145c07p-54, kPio4 = 0x1.921fb54442d18p-1;
constexpr double kPiD = 0x1.921fb54442d18p+1, kTwoOverPi = 0x1.45f306dc9c883p-1, kTanPio8 = 0x1.a827999fcef32p-2;
constexpr double kLn2Hi = 0x1.62e42fefa39efp-1, kLn2Lo = 0x1.abc9e3b39803fp-56, kInvLn2 = 0x1.71547652b82fep+0;

AGX_DEV double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }

// sin and cos of the same angle.  Every angle on this path is bounded (|x| < 64: Euler angles, half angles, yaw
// set-points clipped to +-10): k = rint(x 2/pi), r = x - k pi/2 with pi/2 = hi + lo.
AGX_DEV void sincos_bounded(float x, float &sn, float &cs) {
  const double xd = (double)x;
  const double kd = __builtin_rint(xd * kTwoOverPi);
  const int k = (int)kd;
  double r = fmad(-kd, kPio2Hi, xd);
  r = fmad(-kd, kPio2Lo, r);
  const double z = r * r;
  double ps = 0x1.5e098556d302ep-33;
  ps = fmad(ps, z, -0x1.ae60069e53ef1p-26);
  ps = fmad(ps, z, 0x1.71de379252004p-19);
  ps = fmad(ps, z, -0x1.a01a019e80693p-13);
  ps = fmad(ps, z, 0x1.1111111110ba5p-7);
  ps = fmad(ps, z, -0x1.5555555555555p-3);
  const double s = fmad(z * r, ps, r);
  double pc = 0x1.1c808728603bbp-29;
  pc = fmad(pc, z, -0x1.27e25ca05d2bep-22);
  pc = fmad(pc, z, 0x1.a019ff501e5c1p-16);
  pc = fmad(pc, z, -0x1.6c16c16b5fdb7p-10);
  pc = fmad(pc, z, 0x1.5555555555434p-5);
  const double c = fmad(z * z, pc, fmad(-0.5, z, 1.0));
  const double s0 = (k & 1) ? c : s;
  const double c0 = (k & 1) ? s : c;
  sn = (float)((k & 2) ? -s0 : s0);
  cs = (float)(((k + 1) & 2) ? -c0 : c0);
}

// angle of the point (ax, ay), ax, ay >= 0 and not both 0, in [0, pi/2]: octant reduction min / max with the second
// reduction at tan(pi/8) folded into the same division ((n - d) / (n + d), both exact)
AGX_DEV double atan2_pos(double ay, double ax) {
  const bool swap = ay > ax;
  const double n = swap ? ax : ay, d = swap ? ay : ax;
  const bool mid = n > kTanPio8 * d;
  const double num = mid ? n - d : n, den = mid ? n + d : d;
  const double t = num / den;
  const double z = t * t;
  double p = 0x1.7439839062d96p-6;
  p = fmad(p, z, -0x1.6f15a83e9672ap-5);
  p = fmad(p, z, 0x1.d5dcd0576e964p-5);
  p = fmad(p, z, -0x1.105d0bc7abc71p-4);
  p = fmad(p, z, 0x1.3b0671b310199p-4);
  p = fmad(p, z, -0x1.745c7de4bec48p-4);
  p = fmad(p, z, 0x1.c71c6dc4ee1e0p-4);
  p = fmad(p, z, -0x1.2492491dbb541p-3);
  p = fmad(p, z, 0x1.999999999083bp-3);
  p = fmad(p, z, -0x1.5555555555545p-2);
  double a = fmad(z * t, p, t);
  if (mid) a = kPio4 + a;
  if (swap) a = kPio2Hi - a;
  return a;
}
// atan2 for finite arguments; atan2(0, 0) = 0 like torch (every consumer takes the angle modulo 2 pi)
AGX_DEV float atan2_cw(float y, float x) {
  if (x == 0.0f) {
    if (y == 0.0f) return 0.0f;
    return (y > 0.0f) ? (float)kPio2Hi : -(float)kPio2Hi;
  }
  double a = atan2_pos(__builtin_fabs((double)y), __builtin_fabs((double)x));
  if (x < 0.0f) a = kPiD - a;
  return (float)((y < 0.0f) ? -a : a);
}
// asin, |x| <= 1: the angle of (sqrt((1 - |x|)(1 + |x|)), |x|); both factors are exact in double
AGX_DEV float asin_cw(float xx) {
  const double a = __builtin_fabs((double)xx);
  const double c = __builtin_sqrt((1.0 - a) * (1.0 + a));
  const double r = atan2_pos(a, c);
  return (float)((xx < 0.0f) ? -r : r);
}
// exp; results below the smallest normal float are flushed to 0
AGX_DEV float exp_cw(float x) {
  if (x > 88.7228317f) return INFINITY;
  if (x < -87.3365402f) return 0.0f;
  const double xd = (double)x;
  const double kd = __builtin_rint(xd * kInvLn2);
  const int k = (int)kd;
  double r = fmad(-kd, kLn2Hi, xd);
  r = fmad(-kd, kLn2Lo, r);
  double p = 0x1.288088c0e67a5p-22;
  p = fmad(p, r, 0x1.72c79824255e0p-19);
  p = fmad(p, r, 0x1.a019c971b4d98p-16);
  p = fmad(p, r, 0x1.a019ad55d1aa7p-13);
  p = fmad(p, r, 0x1.6c16c1739a511p-10);
  p = fmad(p, r, 0x1.1111111c5719ap-7);
  p = fmad(p, r, 0x1.5555555554ca7p-5);
  p = fmad(p, r, 0x1.5555555553b48p-3);
  p = fmad(p, r, 0x1.0000000000000p-1);
  const double e = fmad(r * r, p, 1.0 + r);
  const double two_k = __longlong_as_double((long long)(k + 1023) << 52);  // 2^k, -126 <= k <= 128
  return (float)(e * two_k);
}

// utils/math.py:156-172
AGX_DEV Q4 quat_from_euler(float roll, float pitch, float yaw) {
  float sy, cy, sr, cr, sp, cp;
  sincos_bounded(yaw * 0.5f, sy, cy);
  sincos_bounded(roll * 0.5f, sr, cr);
  sincos_bounded(pitch * 0.5f, sp, cp);
  Q4 q;
  q.w = cy * cr * cp + sy * sr * sp;
  q.x = cy * sr * cp - sy * cr * sp;
  q.y = cy * cr * sp + sy * sr * cp;
  q.z = sy * cr * cp - cy * sr * sp;
  return q;
}
// vehicle_frame_quat_from_quat (utils/math.py:176-180): quat_from_euler(0 * roll, 0 * pitch, yaw).  With finite roll and
// pitch the two half angles are +-0, their sines +-0 and cosines exactly 1, and every product / sum of the general formula
// collapses to (0, 0, sin(yaw / 2), cos(yaw / 2)) up to the sign of a zero: one sincos instead of three, equal values.
AGX_DEV Q4 quat_from_yaw(float yaw) {
  float sy, cy;
  sincos_bounded(yaw * 0.5f, sy, cy);
  return Q4{0.0f, 0.0f, sy, cy};
}
// the yaw of euler_xyz_0_2pi on its own (the lean step needs the vehicle-frame quaternion, not roll and pitch)
AGX_DEV float yaw_0_2pi(Q4 q) {
  float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
  float cosy_cosp = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
  float yaw = atan2_cw(siny_cosp, cosy_cosp);
  return pymod(yaw, kTwoPi);
}
// utils/math.py:124-146, angles in [0, 2 pi)
AGX_DEV V3 euler_xyz_0_2pi(Q4 q) {
  float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
  float cosr_cosp = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
  float roll = atan2_cw(sinr_cosp, cosr_cosp);
  float sinp = 2.0f * (q.w * q.y - q.z * q.x);
  float pitch;
  if (fabsf(sinp) >= 1.0f) {
    float sg = (sinp > 0.0f) ? 1.0f : ((sinp < 0.0f) ? -1.0f : 0.0f);
    pitch = (kPi / 2.0f) * sg;
  } else {
    pitch = asin_cw(sinp);
  }
  return V3{pymod(roll, kTwoPi), pymod(pitch, kTwoPi), yaw_0_2pi(q)};
}

// pytorch3d matrix_to_quaternion (argmax branch; base_lee_controller.py:188-189 reorders to xyzw)
AGX_DEV Q4 rotmat_to_quat(M33 m) {
  float t0 = 1.0f + m.m00 + m.m11 + m.m22;
  float t1 = 1.0f + m.m00 - m.m11 - m.m22;
  float t2 = 1.0f - m.m00 + m.m11 - m.m22;
  float t3 = 1.0f - m.m00 - m.m11 + m.m22;
  float q0 = t0 > 0.0f ? sqrtf(t0) : 0.0f;
  float q1 = t1 > 0.0f ? sqrtf(t1) : 0.0f;
  float q2 = t2 > 0.0f ? sqrtf(t2) : 0.0f;
  float q3 = t3 > 0.0f ? sqrtf(t3) : 0.0f;
  int best = 0;
  float qb = q0;
  if (q1 > qb) { best = 1; qb = q1; }
  if (q2 > qb) { best = 2; qb = q2; }
  if (q3 > qb) { best = 3; qb = q3; }
  float r, i, j, k;
  if (best == 0) {
    r = q0 * q0; i = m.m21 - m.m12; j = m.m02 - m.m20; k = m.m10 - m.m01;
  } else if (best == 1) {
    r = m.m21 - m.m12; i = q1 * q1; j = m.m10 + m.m01; k = m.m02 + m.m20;
  } else if (best == 2) {
    r = m.m02 - m.m20; i = m.m10 + m.m01; j = q2 * q2; k = m.m12 + m.m21;
  } else {
    r = m.m10 - m.m01; i = m.m20 + m.m02; j = m.m21 + m.m12; k = q3 * q3;
  }
  float den = 2.0f * (qb > 0.1f ? qb : 0.1f);
  return Q4{i / den, j / den, k / den, r / den};
}

}  // namespace agx
