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
AGX_DEV V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
AGX_DEV float norm(V3 a) { return sqrtf(dot(a, a)); }
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
// sincos_bounded / atan2_cw / asin_cw / exp_cw are explicit single-precision kernels (cephes-style range reduction +
// minimax polynomial; S. Moshier's sinf.c, atanf.c, asinf.c, expf.c) written as individually rounded IEEE operations.
// The CPU restatement the parity tests check against evaluates the SAME sequences, so with contraction off the dynamics path is
// bit-reproducible there (ocml's and glibc's sinf / atan2f / expf agree only to ~1 ulp, like torch's own CPU and CUDA
// kernels); accuracy vs libm is <= 2.5 ulp on the ranges used (DESIGN.md "numerics").
//
// sin and cos of the same angle.  Every angle on this path is bounded (|x| < 64: Euler angles,
// half angles, yaw set-points clipped to +-10), so a 3-term Cody-Waite reduction by pi/2 is
// exact enough and the Payne-Hanek slow path of the generic sinf/cosf (hundreds of
// instructions and ~100 VGPRs of dead weight per call site) is not needed.
AGX_DEV void sincos_bounded(float x, float &sn, float &cs) {
  const float kTwoOverPi = 0.636619772367581343f;
  float kf = rintf(x * kTwoOverPi);
  int k = (int)kf;
  // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188216e-8 (cephes DP1..3 x 2)
  float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
  float z = r * r;
  float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  float s0 = (k & 1) ? pc : ps;
  float c0 = (k & 1) ? ps : pc;
  sn = (k & 2) ? -s0 : s0;
  cs = ((k + 1) & 2) ? -c0 : c0;
}

// cephes atanf: reduction at tan(3 pi / 8) and tan(pi / 8)
AGX_DEV float atan_cw(float xx) {
  float x = fabsf(xx), y;
  if (x > 2.414213562373095f) {
    y = 1.5707963267948966f;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) {
    y = 0.7853981633974483f;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  float z = x * x;
  float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
  y = y + p;
  return (xx < 0.0f) ? -y : y;
}
// atan2 for finite arguments; atan2(0, 0) = 0 like torch (every consumer takes the angle modulo 2 pi)
AGX_DEV float atan2_cw(float y, float x) {
  const float half_pi = 1.5707963267948966f;
  if (x == 0.0f) {
    if (y == 0.0f) return 0.0f;
    return (y > 0.0f) ? half_pi : -half_pi;
  }
  float z = atan_cw(y / x);
  if (x < 0.0f) z = (y < 0.0f) ? z - kPi : z + kPi;
  return z;
}
// cephes asinf, |x| <= 1
AGX_DEV float asin_cw(float xx) {
  float a = fabsf(xx), x, z;
  bool flag = false;
  if (a < 1.0e-4f) return xx;
  if (a > 0.5f) {
    z = 0.5f * (1.0f - a);
    x = sqrtf(z);
    flag = true;
  } else {
    x = a;
    z = x * x;
  }
  z = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
  if (flag) {
    z = z + z;
    z = 1.5707963267948966f - z;
  }
  return (xx < 0.0f) ? -z : z;
}
AGX_DEV float pow2i(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }  // 2^n, -126 <= n <= 127
// cephes expf; results below the smallest normal are flushed to 0
AGX_DEV float exp_cw(float x) {
  if (x > 88.7228317f) return INFINITY;
  if (x < -87.3365402f) return 0.0f;
  float z = floorf(1.44269504088896341f * x + 0.5f);
  float r = x - z * 0.693359375f;
  r = r - z * -2.12194440e-4f;
  int n = (int)z;
  float rr = r * r;
  float p = (((((1.9875691500e-4f * r + 1.3981999507e-3f) * r + 8.3334519073e-3f) * r + 4.1665795894e-2f) * r + 1.6666665459e-1f) * r +
             5.0000001201e-1f) * rr + r + 1.0f;
  int n1 = n / 2, n2 = n - n1;
  return p * pow2i(n1) * pow2i(n2);
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
  float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
  float cosy_cosp = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
  float yaw = atan2_cw(siny_cosp, cosy_cosp);
  return V3{pymod(roll, kTwoPi), pymod(pitch, kTwoPi), pymod(yaw, kTwoPi)};
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
