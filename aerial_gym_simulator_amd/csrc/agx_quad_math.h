// Four lanes per env: vector helpers for kernels that keep a 3-vector / quaternion in ONE register, component c in lane c of
// the env's lane quad (x y z w), and move components with DPP quad permutes (operand modifiers of the vector ALU: no LDS,
// no extra pass).
//
// Why: at the benchmark sizes (8192 envs = 128 waves on 1024 SIMDs) a launch lasts as long as ONE wave takes to issue its
// instructions (~3.5 ns each, whatever their dependencies: profiles/r02_step_latency_experiments.txt).  With one lane per
// env every component of every vector operation is its own instruction; with a quad per env v * s, a + b, a cross product or
// a quaternion rotation are a handful of instructions for all components at once, and the elementary functions
// (atan2, sincos, exp, sqrt, correctly rounded division) are evaluated once for up to four arguments.
//
// Contract: every function performs, per component, EXACTLY the IEEE operations of its namesake in agx_device_math.h, in the
// same order (a - b is sometimes written a + (-b), (a * b) as (b * a): identical results) -- the two formulations are
// bit-identical, which the GPU parity tests check (tests/test_gpu_dynamics.py, test_gpu_task_trace.py).  Scalars (norms,
// angles, dot products) are replicated over the quad.  Lane 3 of a 3-vector holds don't-care data.
#pragma once
#include "agx_device_math.h"

namespace agx {
namespace quad {

template <int A, int B, int C, int D>
AGX_DEV float perm(float x) {  // lane l of the result = lane {A, B, C, D}[l] of the same quad
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), A | (B << 2) | (C << 4) | (D << 6), 0xF, 0xF, true));
}
template <int K>
AGX_DEV float bc(float x) { return perm<K, K, K, K>(x); }     // component K in every lane
AGX_DEV float rot1(float x) { return perm<1, 2, 0, 3>(x); }   // (y, z, x, w)
AGX_DEV float rot2(float x) { return perm<2, 0, 1, 3>(x); }   // (z, x, y, w)
AGX_DEV int lane_in_quad() { return (int)(threadIdx.x & 3u); }
AGX_DEV float neg_if(bool c, float x) { return __uint_as_float(__float_as_uint(x) ^ (c ? 0x80000000u : 0u)); }
// value of lane 0 / 1 / 2 (/ 3) picked by the lane's position in its quad
AGX_DEV float by_lane(int l, float x, float y, float z) { return l == 0 ? x : (l == 1 ? y : z); }
AGX_DEV float by_lane(int l, float x, float y, float z, float w) { return l == 0 ? x : (l == 1 ? y : (l == 2 ? z : w)); }

// a.x*b.x + a.y*b.y + a.z*b.z, replicated
AGX_DEV float dot3(float a, float b) {
  const float p = a * b;
  return (bc<0>(p) + bc<1>(p)) + bc<2>(p);
}
AGX_DEV float dot4(float a, float b) {
  const float p = a * b;
  return ((bc<0>(p) + bc<1>(p)) + bc<2>(p)) + bc<3>(p);
}
// torch.cross (agx_device_math.h `cross`): component i = fma(a_j, b_k, -(a_k b_j))
AGX_DEV float cross3(float a, float b) {
  const float d = __builtin_fmaf(a, rot1(b), -(rot1(a) * b));  // lane x: fma(a.x, b.y, -(a.y b.x)) = the z component, y: x, z: y
  return rot1(d);
}
// torch.norm of a 3-vector (agx_device_math.h `norm`): sqrt(fma(z, z, fma(y, y, x x))), replicated
AGX_DEV float norm3(float a) {
  const float x = bc<0>(a), y = bc<1>(a), z = bc<2>(a);
  return sqrtf(__builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
}
// utils/math.py:329-336
AGX_DEV float quat_rotate(float q, float v) {
  const float w = bc<3>(q);
  const float s = 2.0f * (w * w) - 1.0f;
  const float c = cross3(q, v);
  const float d = dot3(q, v);
  return (v * s + c * w * 2.0f) + q * d * 2.0f;
}
// utils/math.py:340-347
AGX_DEV float quat_rotate_inverse(float q, float v) {
  const float w = bc<3>(q);
  const float s = 2.0f * (w * w) - 1.0f;
  const float c = cross3(q, v);
  const float d = dot3(q, v);
  return (v * s - c * w * 2.0f) + q * d * 2.0f;
}
// utils/math.py:314-320
AGX_DEV float quat_apply(float q, float v) {
  const float t = cross3(q, v) * 2.0f;
  const float u = cross3(q, t);
  return (v + bc<3>(q) * t) + u;
}
AGX_DEV float conj(float q) { return neg_if(lane_in_quad() != 3, q); }
// utils/math.py:243-263: the eight products of the reference's factorisation, four per instruction
AGX_DEV float quat_mul(float a, float b) {
  const int l = lane_in_quad();
  // (a.z - a.x, a.w - a.y, a.w + a.y, a.z + a.x) * (b.x - b.y, b.w + b.z, b.w - b.z, b.x + b.y) = (., yy, zz, ww)
  const float ia = perm<2, 3, 3, 2>(a) + neg_if(l < 2, perm<0, 1, 1, 0>(a));
  const float ib = perm<0, 3, 3, 0>(b) + neg_if((l & 1) == 0, perm<1, 2, 2, 1>(b));
  const float i = ia * ib;
  const float xx = (bc<3>(i) + bc<1>(i)) + bc<2>(i);
  const float qq = 0.5f * (xx + bc<0>(i));
  const float z = l == 0 ? xx : i;  // (xx, yy, zz, ww)
  // (a.x + a.w, a.w - a.x, a.z + a.y, a.z - a.y) * (b.x + b.w, b.y + b.z, b.w - b.x, b.y - b.z)
  const float pa = perm<0, 3, 2, 2>(a) + neg_if((l & 1) == 1, perm<3, 0, 1, 1>(a));
  const float pb = perm<0, 1, 3, 1>(b) + neg_if(l >= 2, perm<3, 2, 0, 2>(b));
  return (qq - z) + pa * pb;
}

// utils/math.py:124-146 (euler_xyz_0_2pi): roll, pitch, yaw in lanes 0, 1, 2, each in [0, 2 pi)
AGX_DEV float euler_xyz_0_2pi(float q) {
  const int l = lane_in_quad();
  const float w = bc<3>(q);
  const float t1 = w * q;               // (wx, wy, wz, .)
  const float t2 = rot1(q) * rot2(q);   // (yz, zx, xy, .)
  const float num = 2.0f * (l == 1 ? t1 - t2 : t1 + t2);  // sinr_cosp, sinp, siny_cosp
  const float sq = q * q;
  const float ww = bc<3>(sq), xx = bc<0>(sq), yy = bc<1>(sq), zz = bc<2>(sq);
  const float a1 = (l == 2 ? ww + xx : ww - xx) - yy;
  const float den = l == 2 ? a1 - zz : a1 + zz;  // cosr_cosp (lane 0), cosy_cosp (lane 2)
  // roll and yaw are atan2(num, den), the pitch is asin(num) = the angle of the point (sqrt((1 - |num|)(1 + |num|)), |num|):
  // ONE evaluation of atan2_pos for the three (lane 1 takes the asin's abscissa), then what atan2_cw / asin_cw do around it --
  // the same double operations on the same operands as the two separate calls, hence the same bits
  const double an = __builtin_fabs((double)num);
  const double ax = l == 1 ? __builtin_sqrt((1.0 - an) * (1.0 + an)) : __builtin_fabs((double)den);
  const double ang = atan2_pos(an, ax);  // (NaN where a special case below applies: discarded)
  // atan2_cw
  const double aq = den < 0.0f ? kPiD - ang : ang;
  float at = (float)(num < 0.0f ? -aq : aq);
  if (den == 0.0f) at = num == 0.0f ? 0.0f : (num > 0.0f ? (float)kPio2Hi : -(float)kPio2Hi);
  // asin_cw, and the clamp of utils/math.py:136-140
  float pitch = (float)(num < 0.0f ? -ang : ang);
  if (fabsf(num) >= 1.0f) {
    const float sg = (num > 0.0f) ? 1.0f : ((num < 0.0f) ? -1.0f : 0.0f);
    pitch = (kPi / 2.0f) * sg;
  }
  return pymod(l == 1 ? pitch : at, kTwoPi);
}

// pytorch3d matrix_to_quaternion of the matrix with COLUMNS b1, b2, b3 (agx_device_math.h rotmat_to_quat), xyzw in lanes
AGX_DEV float rotmat_cols_to_quat(float b1, float b2, float b3) {
  const int l = lane_in_quad();
  const float m00 = bc<0>(b1), m11 = bc<1>(b2), m22 = bc<2>(b3);
  // t0 = 1 + m00 + m11 + m22, t1 = 1 + m00 - m11 - m22, t2 = 1 - m00 + m11 - m22, t3 = 1 - m00 - m11 + m22
  const float t = ((1.0f + neg_if(l >= 2, m00)) + neg_if((l & 1) == 1, m11)) + neg_if(l == 1 || l == 2, m22);
  const float ql = t > 0.0f ? sqrtf(t) : 0.0f;
  const float q0 = bc<0>(ql), q1 = bc<1>(ql), q2 = bc<2>(ql), q3 = bc<3>(ql);
  int best = 0;
  float qb = q0;
  if (q1 > qb) { best = 1; qb = q1; }
  if (q2 > qb) { best = 2; qb = q2; }
  if (q3 > qb) { best = 3; qb = q3; }
  const float qq = qb * qb;
  // x1 = (m21, m02, m10) = (b2.z, b3.x, b1.y), x2 = (m12, m20, m01) = (b3.y, b1.z, b2.x)
  const float x1 = by_lane(l, rot2(b2), rot2(b3), rot2(b1));
  const float x2 = by_lane(l, rot1(b3), rot1(b1), rot1(b2));
  const float am = x1 - x2;  // (m21 - m12, m02 - m20, m10 - m01)
  const float sm = x1 + x2;  // (m21 + m12, m02 + m20, m10 + m01)
  // (the permutes are evaluated on the whole quad BEFORE the per-lane pick: a DPP read inside lane-dependent control flow
  // would find its source lane switched off)
  float c;
  if (best == 0) {  // `best` is the same in the four lanes of a quad
    c = l == 3 ? qq : am;                                  // (A0, A1, A2, q0^2)
  } else if (best == 1) {
    const float a0 = bc<0>(am), ps = perm<0, 2, 1, 3>(sm);
    c = l == 0 ? qq : (l == 3 ? a0 : ps);                  // (q1^2, S2, S1, A0)
  } else if (best == 2) {
    const float a1 = bc<1>(am), ps = perm<2, 1, 0, 3>(sm);
    c = l == 1 ? qq : (l == 3 ? a1 : ps);                  // (S2, q2^2, S0, A1)
  } else {
    const float a2 = bc<2>(am), ps = perm<1, 0, 2, 3>(sm);
    c = l == 2 ? qq : (l == 3 ? a2 : ps);                  // (S1, S0, q3^2, A2)
  }
  const float den = 2.0f * (qb > 0.1f ? qb : 0.1f);
  return c / den;
}

}  // namespace quad
}  // namespace agx
