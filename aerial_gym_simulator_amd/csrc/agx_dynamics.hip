// Fused per-env dynamics for gfx950: update_states -> Lee controller -> allocation ->
// motor model -> drag / disturbance -> rigid-body integration -> collision flag,
// k physics sub-steps per launch, one lane per env, SoA loads/stores (coalesced: lane i
// touches X[c*N + i], i.e. 256 contiguous bytes per wave instruction).
//
// The reference runs this as ~590 tiny torch ops per sub-step plus PhysX
// (SURVEY.md section 2.1 C/D); here the whole env step is one kernel whose HBM traffic
// is the state itself: 13 floats in/out, M thrusts in/out, A actions, 16 derived
// floats out, 3M motor parameters and 12 gains in.
//
// No MFMA: there is no dense contraction in this path (the 6xM allocation products are
// per-env matrix-vector products with constant matrices held in SGPRs).
#include "agx_common.h"
// Arithmetic of the state path (controller, motor model, integrator, rewards).  Default: every + - * / sqrt is one
// correctly rounded IEEE operation (no fma contraction) and the elementary functions are the explicit kernels of
// agx_device_math.h, i.e. exactly the sequence the CPU restatement of the parity tests evaluates: the whole env step is
// BIT-IDENTICAL to that restatement (tests assert array_equal), which in turn is pinned to the reference's own outputs on the
// CPU.  Measured cost of exactness on MI355X (profiles/r02_parity_variants.json): k_env_step 10.3 -> 11.2 us at 8192 envs,
// 138 -> 154 us at 2^21 envs.  The two switches below re-enable the faster, ~1e-6-accurate arithmetic for A/B runs:
//   AGX_DYN_CONTRACT=1  let the compiler contract a*b+c into v_fma_f32
//   AGX_DYN_FAST_RCP=1  hardware v_rcp / v_rsq / v_sqrt (+ one Newton step) instead of correctly rounded division / sqrt
#ifndef AGX_DYN_CONTRACT
#define AGX_DYN_CONTRACT 0
#endif
#ifndef AGX_DYN_WAVES
// waves per SIMD the straight-line env-step kernels are compiled for.  With the scalar-base SoA addressing (soa_at) they
// need <= 168 VGPRs and fit 3; a limit of 4 (128 VGPRs) spills.  Measured (profiles/r01_soa_addressing.txt).
#define AGX_DYN_WAVES 3
#endif
#ifndef AGX_DYN_WAVES_LEAN_LAWS
#define AGX_DYN_WAVES_LEAN_LAWS 4  // env_step_single_waves
#endif
#ifndef AGX_DYN_WAVES_LOOP
// the k-loop variants (k > 1 sub-steps per launch) keep the loop-carried motor / action state next to everything the
// straight-line kernel needs: at 3 waves per SIMD (168 VGPRs) they spilled 24-136 VGPRs to scratch; 2 waves (256) fit.
#define AGX_DYN_WAVES_LOOP 2
#endif
#if AGX_DYN_CONTRACT
#pragma clang fp contract(fast)
#else
#pragma clang fp contract(off)
#endif
#include "agx_device_math.h"
#include "agx_nav_parts.h"
#include "agx_quad_math.h"
#include "agx_rng.h"
#include "agx_step_signal.h"

#include <cstdlib>
#include <type_traits>

#ifndef AGX_DYN_FAST_RCP
#define AGX_DYN_FAST_RCP 0
#endif
namespace agx {
#if AGX_DYN_FAST_RCP
// one Newton step each: 1 ulp -> about 0.5 ulp (not correctly rounded, not meant to be), 3 / 4 instructions
AGX_DEV float srcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return fmaf(r, fmaf(-x, r, 1.0f), r);
}
AGX_DEV float fdiv(float a, float b) { return a * srcp(b); }
AGX_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
AGX_DEV float fdiv(float a, float b) { return a / b; }
AGX_DEV float fsqrt(float x) { return sqrtf(x); }
#endif
// v / |v| the way torch evaluates it: the norm first, then one division per component
AGX_DEV V3 normalized(V3 v) {
  float nv = norm(v);
  return V3{fdiv(v.x, nv), fdiv(v.y, nv), fdiv(v.z, nv)};
}
}  // namespace agx

namespace agx {

struct EnvState {
  V3 p;
  Q4 q;
  V3 v, w;
};
struct Derived {
  V3 euler;
  Q4 qveh;
  V3 vveh, vbody, wbody;
};
struct Gains {
  V3 kp, kv, kr, kw;
};
struct Wrench {
  V3 f, t;
};


// SoA element (component c of env i): uniform column base (scalar unit) + one 32-bit byte offset per lane,
// i.e. the `global_load v, v_off, s[base]` addressing form instead of a 64-bit VGPR address per access.
// Round 4: as a BUFFER access -- `buffer_load_dword v, v_off, s[descriptor], s_column offen`: the array's base in a 128-bit
// descriptor (scalar registers, rebuilt where it is used: four scalar instructions), the column offset c n sizeof(T) in a scalar
// register, the lane's part i sizeof(T) in ONE vector register shared by every access of the kernel.  The pointer form above
// compiles to that `global_load v, v_off, s[base]` only when the instruction selector finds the offset's 32 -> 64-bit extension
// in the access's own basic block; behind any run-time condition it does not, and each access cost a 64-bit vector add and a
// register pair: 288 of the ~2500 vector instructions of k_env_step<4, position, single> and its largest block of live registers
// (profiles/r04_at_scale_experiments.txt).  n x 16 columns x 4 bytes < 2^32.
template <class T>
struct SoaRef {
  static_assert(sizeof(T) == 4, "32-bit elements");
  T *base;
  unsigned col_bytes, lane_bytes;
  AGX_DEV __amdgpu_buffer_rsrc_t rsrc() const {
    // raw buffer (stride 0), every offset in range, gfx9 data format word (composable_kernel: CK_BUFFER_RESOURCE_3RD_DWORD)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<typename std::remove_const<T>::type *>(base), 0, -1, 0x00020000);
  }
  AGX_DEV operator typename std::remove_const<T>::type() const {
    return __builtin_bit_cast(typename std::remove_const<T>::type,
                              __builtin_amdgcn_raw_buffer_load_b32(rsrc(), (int)lane_bytes, (int)col_bytes, 0));
  }
  AGX_DEV void operator=(typename std::remove_const<T>::type v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rsrc(), (int)lane_bytes, (int)col_bytes, 0);
  }
};
template <class T>
AGX_DEV SoaRef<T> soa_at(T *base, int c, int n, int i) {
  return SoaRef<T>{base, (unsigned)c * (unsigned)n * (unsigned)sizeof(T), (unsigned)i * (unsigned)sizeof(T)};
}
#define AGX_AT(p, c) agx::soa_at((p), (c), n, i)
// The lane-quad kernels index a column by the lane's component (c0 + l): the per-lane part of the address, (l n + i) sizeof(T),
// is computed ONCE as a 32-bit byte offset (n <= 65536 there) and every access is scalar column base + that offset -- instead of
// a 64-bit multiply-add and two 64-bit adds on the vector unit per access, in front of the kernel's first load.
template <class T>
AGX_DEV T &soa_at_off(T *base, int c, int n, unsigned off_bytes) {
  T *col = base + (ptrdiff_t)c * (ptrdiff_t)n;
  return *reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(col)) + (size_t)off_bytes);
}
// (Stays the pointer form: as buffer accesses the 8192-env step was 3 % SLOWER -- 12.8 vs 12.4 us, measured -- these kernels run
//  one wave per SIMD and are bound by that wave's own instruction chain, to which the descriptor set-up and the extra branches of
//  the `pointer ? load : uniform` arms add; the one-lane kernels are bound by throughput and registers, where they pay.)
#define AGX_QAT(p, c, off) agx::soa_at_off((p), (c), n, (off))

AGX_DEV EnvState load_state(const float *__restrict__ s, int n, int i) {
  EnvState e;
  e.p = V3{AGX_AT(s, 0), AGX_AT(s, 1), AGX_AT(s, 2)};
  e.q = Q4{AGX_AT(s, 3), AGX_AT(s, 4), AGX_AT(s, 5), AGX_AT(s, 6)};
  e.v = V3{AGX_AT(s, 7), AGX_AT(s, 8), AGX_AT(s, 9)};
  e.w = V3{AGX_AT(s, 10), AGX_AT(s, 11), AGX_AT(s, 12)};
  return e;
}
AGX_DEV void store_state(float *__restrict__ s, int n, int i, const EnvState &e) {
  AGX_AT(s, 0) = e.p.x; AGX_AT(s, 1) = e.p.y; AGX_AT(s, 2) = e.p.z;
  AGX_AT(s, 3) = e.q.x; AGX_AT(s, 4) = e.q.y; AGX_AT(s, 5) = e.q.z; AGX_AT(s, 6) = e.q.w;
  AGX_AT(s, 7) = e.v.x; AGX_AT(s, 8) = e.v.y; AGX_AT(s, 9) = e.v.z;
  AGX_AT(s, 10) = e.w.x; AGX_AT(s, 11) = e.w.y; AGX_AT(s, 12) = e.w.z;
}
AGX_DEV void store_derived(float *__restrict__ d, int n, int i, const Derived &x) {
  AGX_AT(d, 0) = x.euler.x; AGX_AT(d, 1) = x.euler.y; AGX_AT(d, 2) = x.euler.z;
  AGX_AT(d, 3) = x.qveh.x; AGX_AT(d, 4) = x.qveh.y; AGX_AT(d, 5) = x.qveh.z; AGX_AT(d, 6) = x.qveh.w;
  AGX_AT(d, 7) = x.vveh.x; AGX_AT(d, 8) = x.vveh.y; AGX_AT(d, 9) = x.vveh.z;
  AGX_AT(d, 10) = x.vbody.x; AGX_AT(d, 11) = x.vbody.y; AGX_AT(d, 12) = x.vbody.z;
  AGX_AT(d, 13) = x.wbody.x; AGX_AT(d, 14) = x.wbody.y; AGX_AT(d, 15) = x.wbody.z;
}
AGX_DEV void store_body_velocities(float *__restrict__ d, int n, int i, const Derived &x) {
  AGX_AT(d, 10) = x.vbody.x; AGX_AT(d, 11) = x.vbody.y; AGX_AT(d, 12) = x.vbody.z;
  AGX_AT(d, 13) = x.wbody.x; AGX_AT(d, 14) = x.wbody.y; AGX_AT(d, 15) = x.wbody.z;
}
AGX_DEV Derived load_derived(const float *__restrict__ d, int n, int i) {
  Derived x;
  x.euler = V3{AGX_AT(d, 0), AGX_AT(d, 1), AGX_AT(d, 2)};
  x.qveh = Q4{AGX_AT(d, 3), AGX_AT(d, 4), AGX_AT(d, 5), AGX_AT(d, 6)};
  x.vveh = V3{AGX_AT(d, 7), AGX_AT(d, 8), AGX_AT(d, 9)};
  x.vbody = V3{AGX_AT(d, 10), AGX_AT(d, 11), AGX_AT(d, 12)};
  x.wbody = V3{AGX_AT(d, 13), AGX_AT(d, 14), AGX_AT(d, 15)};
  return x;
}
AGX_DEV Gains uniform_gains(const AgxRobotParams &P) {
  Gains k;
  k.kp = V3{P.gains_uniform[0], P.gains_uniform[1], P.gains_uniform[2]};
  k.kv = V3{P.gains_uniform[3], P.gains_uniform[4], P.gains_uniform[5]};
  k.kr = V3{P.gains_uniform[6], P.gains_uniform[7], P.gains_uniform[8]};
  k.kw = V3{P.gains_uniform[9], P.gains_uniform[10], P.gains_uniform[11]};
  return k;
}
AGX_DEV Gains load_gains(const float *__restrict__ g, int n, int i) {
  Gains k;
  k.kp = V3{AGX_AT(g, 0), AGX_AT(g, 1), AGX_AT(g, 2)};
  k.kv = V3{AGX_AT(g, 3), AGX_AT(g, 4), AGX_AT(g, 5)};
  k.kr = V3{AGX_AT(g, 6), AGX_AT(g, 7), AGX_AT(g, 8)};
  k.kw = V3{AGX_AT(g, 9), AGX_AT(g, 10), AGX_AT(g, 11)};
  return k;
}

// BaseMultirotor.update_states, base_multirotor.py:287-294
AGX_DEV Derived update_states(const EnvState &s) {
  Derived d;
  V3 e = euler_xyz_0_2pi(s.q);
  d.euler = V3{ssa(e.x), ssa(e.y), ssa(e.z)};
  // vehicle_frame_quat_from_quat: euler * [0, 0, 1] (utils/math.py:176-180)
  d.qveh = quat_from_yaw(e.z * 1.0f);  // = quat_from_euler(e.x * 0, e.y * 0, e.z * 1), see agx_device_math.h
  d.vveh = quat_rotate_inverse(d.qveh, s.v);
  d.vbody = quat_rotate_inverse(s.q, s.v);
  d.wbody = quat_rotate_inverse(s.q, s.w);
  return d;
}

// The lean step (AGX_LAUNCH_LEAN) does not maintain Euler angles / vehicle-frame velocity, and under the laws that read neither
// (position, fully actuated; no controller) does not EVALUATE them either: roll and pitch are two of the three float64
// function evaluations of update_states.  What remains is what the task epilogue and the observation read, the same
// operations on the same operands: vehicle-frame quaternion (from the yaw), body-frame velocities.
AGX_DEV Derived update_states_lean(const EnvState &s) {
  Derived d{};
  d.qveh = quat_from_yaw(yaw_0_2pi(s.q) * 1.0f);
  d.vbody = quat_rotate_inverse(s.q, s.v);
  d.wbody = quat_rotate_inverse(s.q, s.w);
  return d;
}
// ... and the observation behind a reset reads the body-frame velocities only
AGX_DEV Derived update_states_body(const EnvState &s) {
  Derived d{};
  d.vbody = quat_rotate_inverse(s.q, s.v);
  d.wbody = quat_rotate_inverse(s.q, s.w);
  return d;
}

// base_lee_controller.py:120-134
// ZERO_VEL: the caller's velocity set-point is the constant 0 (position / fully actuated control): rotating it gives 0
template <bool ZERO_VEL = false>
AGX_DEV V3 compute_acceleration(const EnvState &s, Q4 qveh, V3 sp_pos, V3 sp_vel, const Gains &g) {
  V3 sp_vel_w = ZERO_VEL ? V3{0.0f, 0.0f, 0.0f} : quat_rotate(qveh, sp_vel);
  V3 pe = sp_pos - s.p;
  V3 ve = sp_vel_w - s.v;
  return V3{g.kp.x * pe.x + g.kv.x * ve.x, g.kp.y * pe.y + g.kv.y * ve.y, g.kp.z * pe.z + g.kv.z * ve.z};
}

// base_lee_controller.py:136-154 (sp_w.z is clamped in place by the caller-visible ref)
// ZERO_RATE: the caller's angular-velocity set-point is the constant 0
template <bool ZERO_RATE = false>
AGX_DEV V3 compute_body_torque(const AgxRobotParams &P, Q4 q, V3 wb, Q4 qd, V3 &sp_w, const Gains &g) {
  sp_w.z = fminf(fmaxf(sp_w.z, -P.max_yaw_rate), P.max_yaw_rate);
  Q4 qe = quat_mul(conj(q), qd);
  M33 R = quat_to_rotmat(qe);
  V3 rot_err = V3{0.5f * (-(R.m21 - R.m12)), 0.5f * (R.m20 - R.m02), 0.5f * (-(R.m10 - R.m01))};
  V3 wsp_b = ZERO_RATE ? V3{0.0f, 0.0f, 0.0f} : quat_rotate(qe, sp_w);
  V3 Jw = V3{P.inertia[0] * wb.x + P.inertia[1] * wb.y + P.inertia[2] * wb.z,
             P.inertia[3] * wb.x + P.inertia[4] * wb.y + P.inertia[5] * wb.z,
             P.inertia[6] * wb.x + P.inertia[7] * wb.y + P.inertia[8] * wb.z};
  V3 ff = cross(wb, Jw);
  V3 we = wb - wsp_b;
  return V3{-g.kr.x * rot_err.x - g.kw.x * we.x + ff.x, -g.kr.y * rot_err.y - g.kw.y * we.y + ff.y,
            -g.kr.z * rot_err.z - g.kw.z * we.z + ff.z};
}

// base_lee_controller.py:173-194
AGX_DEV Q4 desired_orientation_pos_vel(V3 f, float yaw) {
  V3 b3 = normalized(f);
  float sy, cy;
  sincos_bounded(yaw, sy, cy);
  V3 tmp = V3{cy, sy, 0.0f};
  V3 b2 = normalized(cross(b3, tmp));
  V3 b1 = cross(b2, b3);
  M33 R{b1.x, b2.x, b3.x, b1.y, b2.y, b3.y, b1.z, b2.z, b3.z};
  return rotmat_to_quat(R);
}

// base_lee_controller.py:158-169
AGX_DEV Q4 desired_orientation_forces_yaw(V3 f, float yaw) {
  float pitch = atan2_cw(f.x, f.z);
  float roll = atan2_cw(-f.y, sqrtf(f.z * f.z + f.x * f.x));
  return quat_from_euler(roll, pitch, yaw);
}

// base_lee_controller.py:201-215 (stale matrix entries only ever multiply zero rates)
AGX_DEV V3 euler_rates_to_body_rates(V3 euler, V3 r) {
  float sp, cp, sr, cr;
  sincos_bounded(euler.y, sp, cp);
  sincos_bounded(euler.x, sr, cr);
  return V3{1.0f * r.x + 0.0f * r.y + (-sp) * r.z, 0.0f * r.x + cr * r.y + (sr * cp) * r.z,
            0.0f * r.x + (-sr) * r.y + (cr * cp) * r.z};
}

// One env's controller (control/controllers/*.py).  a[] holds the +-10 clipped action and
// is mutated where the reference mutates it.
template <int CTRL>
AGX_DEV Wrench run_controller(const AgxRobotParams &P, const EnvState &s, const Derived &d, float (&a)[AGX_MAX_ACTIONS],
                              const Gains &g) {
  Wrench w{V3{0, 0, 0}, V3{0, 0, 0}};
  const V3 grav = V3{P.gravity[0], P.gravity[1], P.gravity[2]};
  const float m = P.mass;
  const V3 zero = V3{0, 0, 0};
  switch (CTRL) {
    case AGX_CTRL_POSITION: {  // position_control.py:20-51
      V3 acc = compute_acceleration<true>(s, d.qveh, V3{a[0], a[1], a[2]}, zero, g);
      V3 f = (acc - grav) * m;
      M33 R = quat_to_rotmat(s.q);
      w.f.z = f.x * R.m02 + f.y * R.m12 + f.z * R.m22;
      Q4 qd = desired_orientation_pos_vel(f, a[3]);
      V3 wsp = zero;
      w.t = compute_body_torque<true>(P, s.q, d.wbody, qd, wsp, g);
    } break;
    case AGX_CTRL_VELOCITY: {  // velocity_control.py:18-51
      V3 acc = compute_acceleration(s, d.qveh, s.p, V3{a[0], a[1], a[2]}, g);
      V3 f = (acc - grav) * m;
      M33 R = quat_to_rotmat(s.q);
      w.f.z = f.x * R.m02 + f.y * R.m12 + f.z * R.m22;
      Q4 qd = desired_orientation_pos_vel(f, d.euler.z);
      V3 wsp = euler_rates_to_body_rates(d.euler, V3{0, 0, a[3]});
      w.t = compute_body_torque(P, s.q, d.wbody, qd, wsp, g);
    } break;
    case AGX_CTRL_ATTITUDE: {  // attitude_control.py:16-43
      w.f.z = (a[0] + 1.0f) * m * norm(grav);
      V3 wsp = euler_rates_to_body_rates(d.euler, V3{0, 0, a[3]});
      Q4 qd = quat_from_euler(a[1], a[2], d.euler.z);
      w.t = compute_body_torque(P, s.q, d.wbody, qd, wsp, g);
    } break;
    case AGX_CTRL_RATES: {  // rates_control.py:16-30 (line 25's broadcast bug -> z component)
      w.f.z = (a[0] - grav.z) * m;
      V3 wsp = V3{a[1], a[2], a[3]};
      w.t = compute_body_torque(P, s.q, d.wbody, s.q, wsp, g);
      a[3] = wsp.z;  // in-place yaw-rate clamp
    } break;
    case AGX_CTRL_ACCELERATION: {  // acceleration_control.py:16-45
      V3 f = (V3{a[0], a[1], a[2]} - grav) * m;
      M33 R = quat_to_rotmat(s.q);
      w.f.z = f.x * R.m02 + f.y * R.m12 + f.z * R.m22;
      Q4 qd = desired_orientation_forces_yaw(f, d.euler.z);
      V3 wsp = euler_rates_to_body_rates(d.euler, V3{0, 0, a[3]});
      w.t = compute_body_torque(P, s.q, d.wbody, qd, wsp, g);
    } break;
    case AGX_CTRL_VEL_STEERING: {  // velocity_steeing_angle_controller.py:15-45
      V3 acc = compute_acceleration(s, d.qveh, s.p, V3{a[0], a[1], a[2]}, g);
      V3 f = (acc - grav) * m;
      M33 R = quat_to_rotmat(s.q);
      w.f.z = f.x * R.m02 + f.y * R.m12 + f.z * R.m22;
      Q4 qd = desired_orientation_pos_vel(f, a[3]);
      V3 wsp = euler_rates_to_body_rates(d.euler, zero);
      w.t = compute_body_torque(P, s.q, d.wbody, qd, wsp, g);
    } break;
    case AGX_CTRL_FULLY_ACTUATED: {  // fully_actuated_control.py:14-32
      float nq = sqrtf(a[3] * a[3] + a[4] * a[4] + a[5] * a[5] + a[6] * a[6]);
      nq = nq < 1e-9f ? 1e-9f : nq;
      a[3] = a[3] / nq; a[4] = a[4] / nq; a[5] = a[5] / nq; a[6] = a[6] / nq;
      V3 acc = compute_acceleration<true>(s, d.qveh, V3{a[0], a[1], a[2]}, zero, g);
      V3 f = (acc - grav) * m;
      w.f = quat_rotate_inverse(s.q, f);
      V3 wsp = zero;
      w.t = compute_body_torque<true>(P, s.q, d.wbody, Q4{a[3], a[4], a[5], a[6]}, wsp, g);
    } break;
    default: break;
  }
  return w;
}

// control/motor_model.py:88-250
AGX_DEV float clamp_minmax(float x, float lo, float hi) { return fmaxf(fminf(x, hi), lo); }
AGX_DEV float sgnf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
AGX_DEV float motor_rate(float err, float mix, float max_rate) { return clamp_minmax(mix * err, -max_rate, max_rate); }
AGX_DEV float rk4_delta(float ref, float cur, float mix, float max_rate, float dt, float dt_over_6) {
  float k1 = motor_rate(ref - cur, mix, max_rate);
  float k2 = motor_rate(ref - (cur + 0.5f * dt * k1), mix, max_rate);
  float k3 = motor_rate(ref - (cur + 0.5f * dt * k2), mix, max_rate);
  float k4 = motor_rate(ref - (cur + dt * k3), mix, max_rate);
  return dt_over_6 * (k1 + 2.0f * k2 + 2.0f * k3 + k4);
}
AGX_DEV float motor_update(const AgxRobotParams &P, float ref, float cur, float kT, float tau_inc, float tau_dec) {
  const float dt = P.dt;
  ref = fminf(fmaxf(ref, P.min_thrust), P.max_thrust);
  float err = ref - cur;
  float tc = (sgnf(cur) * sgnf(err) < 0.0f) ? tau_dec : tau_inc;
  float mix = fdiv(1.0f, P.use_discrete_approximation ? dt + tc : tc);
  if (P.use_rps) {
    float cur_rpm = fsqrt(fdiv(cur, kT));
    float des_rpm = fsqrt(fdiv(ref, kT));
    if (P.integration_rk4)
      cur_rpm += rk4_delta(des_rpm, cur_rpm, mix, P.max_rate, dt, P.dt_over_6);
    else
      cur_rpm += motor_rate(des_rpm - cur_rpm, mix, P.max_rate) * dt;
    return kT * (cur_rpm * cur_rpm);
  }
  if (P.integration_rk4) return cur + rk4_delta(ref, cur, mix, P.max_rate, dt, P.dt_over_6);
  return cur + motor_rate(err, mix, P.max_rate) * dt;
}

// Rigid-body update replacing gym.simulate (PhysX): see DESIGN.md "integrator".
AGX_DEV void integrate(const AgxRobotParams &P, EnvState &s, V3 Fb, V3 Tb) {
  const float dt = P.dt;
  V3 Fw = quat_rotate(s.q, Fb);
  V3 wb = quat_rotate_inverse(s.q, s.w);
  V3 Jw = V3{P.inertia[0] * wb.x + P.inertia[1] * wb.y + P.inertia[2] * wb.z,
             P.inertia[3] * wb.x + P.inertia[4] * wb.y + P.inertia[5] * wb.z,
             P.inertia[6] * wb.x + P.inertia[7] * wb.y + P.inertia[8] * wb.z};
  V3 rhs = Tb - cross(wb, Jw);
  V3 dwb = V3{P.inertia_inv[0] * rhs.x + P.inertia_inv[1] * rhs.y + P.inertia_inv[2] * rhs.z,
              P.inertia_inv[3] * rhs.x + P.inertia_inv[4] * rhs.y + P.inertia_inv[5] * rhs.z,
              P.inertia_inv[6] * rhs.x + P.inertia_inv[7] * rhs.y + P.inertia_inv[8] * rhs.z};
  V3 wb_new = V3{wb.x + dt * dwb.x, wb.y + dt * dwb.y, wb.z + dt * dwb.z};
  V3 w_new = quat_rotate(s.q, wb_new);
  V3 v_new = V3{s.v.x + dt * fdiv(Fw.x, P.mass), s.v.y + dt * fdiv(Fw.y, P.mass), s.v.z + dt * fdiv(Fw.z, P.mass)};
  v_new = V3{v_new.x + P.gravity[0] * dt, v_new.y + P.gravity[1] * dt, v_new.z + P.gravity[2] * dt};
  float ml = fmaxf(1.0f - P.linear_damping * dt, 0.0f);
  float ma = fmaxf(1.0f - P.angular_damping * dt, 0.0f);
  v_new = v_new * ml;
  w_new = w_new * ma;
  float v2 = dot(v_new, v_new), w2 = dot(w_new, w_new);
  if (v2 > P.max_linear_velocity * P.max_linear_velocity) v_new = v_new * fdiv(P.max_linear_velocity, fsqrt(v2));
  if (w2 > P.max_angular_velocity * P.max_angular_velocity) w_new = w_new * fdiv(P.max_angular_velocity, fsqrt(w2));
  s.p = V3{s.p.x + v_new.x * dt, s.p.y + v_new.y * dt, s.p.z + v_new.z * dt};
  float wm2 = dot(w_new, w_new);
  if (wm2 != 0.0f) {
    float wm = fsqrt(wm2);
    float half = dt * wm * 0.5f;
    float sn, cs;
    sincos_bounded(half, sn, cs);  // |half| = dt |w| / 2 <= 0.5 (|w| <= 100 rad/s)
    float sc = fdiv(sn, wm);
    float x1 = w_new.x * sc, y1 = w_new.y * sc, z1 = w_new.z * sc;
    Q4 q = s.q;
    float rx = x1 * q.w + y1 * q.z - z1 * q.y;
    float ry = y1 * q.w + z1 * q.x - x1 * q.z;
    float rz = z1 * q.w + x1 * q.y - y1 * q.x;
    float rw = -(x1 * q.x) - y1 * q.y - z1 * q.z;
    rx += q.x * cs; ry += q.y * cs; rz += q.z * cs; rw += q.w * cs;
    float nn = fsqrt(rx * rx + ry * ry + rz * rz + rw * rw);
    s.q = Q4{fdiv(rx, nn), fdiv(ry, nn), fdiv(rz, nn), fdiv(rw, nn)};
  }
  s.v = v_new;
  s.w = w_new;
}

#pragma clang fp contract(off)  // everything below: one IEEE operation per + - * /

// sphere (robot collision sphere, quad.urdf:16) vs obstacle OBBs; replaces the PhysX
// contact-force test of env_manager.py:358-362.  The predicate uses only IEEE + - *
// (bit-reproducible, written out here so no contracted helper is inlined); the culling in
// front of it is conservative, so the flag is exact.
AGX_DEV bool sphere_hits_box(V3 p, V3 c, Q4 q, V3 h, float r2) {
  // quat_rotate_inverse(q, p - c), utils/math.py:340-347
  V3 v = V3{p.x - c.x, p.y - c.y, p.z - c.z};
  float s = 2.0f * (q.w * q.w) - 1.0f;
  V3 cr = V3{q.y * v.z - q.z * v.y, q.z * v.x - q.x * v.z, q.x * v.y - q.y * v.x};
  float d = q.x * v.x + q.y * v.y + q.z * v.z;
  V3 l = V3{v.x * s - cr.x * q.w * 2.0f + q.x * d * 2.0f, v.y * s - cr.y * q.w * 2.0f + q.y * d * 2.0f,
            v.z * s - cr.z * q.w * 2.0f + q.z * d * 2.0f};
  float ex = fabsf(l.x) - h.x, ey = fabsf(l.y) - h.y, ez = fabsf(l.z) - h.z;
  float d2 = 0.0f;
  if (ex > 0.0f) d2 += ex * ex;
  if (ey > 0.0f) d2 += ey * ey;
  if (ez > 0.0f) d2 += ez * ez;
  return d2 < r2;
}

// One pass over the env's K boxes for ALL k sub-step positions (kept in LDS,
// traj[(s*3+c)*bd + tid]): each box is fetched once per env step instead of once per sub-step,
// and boxes whose bounding sphere cannot reach the AABB of the k positions cost 4 loads.
AGX_DEV bool collide_trajectory(const float *__restrict__ boxes, int nb, int n, int i, const float *traj, int k, int bd,
                                int tid, V3 lo, V3 hi, float r) {
  bool hit = false;
  const float r2 = r * r;
  // the cull data (centre, bounding radius) of box b + 1 is fetched while box b is processed: with few envs the
  // loop is a chain of dependent HBM round trips otherwise (101 us at 256 envs x 106 boxes)
  const float *b0 = boxes + i;
  float ncx = 0.0f, ncy = 0.0f, ncz = 0.0f, nrad = 0.0f;
  if (nb > 0) { ncx = b0[0]; ncy = b0[(size_t)n]; ncz = b0[2 * (size_t)n]; nrad = b0[10 * (size_t)n]; }
  for (int b = 0; b < nb; ++b) {
    const float *bx = boxes + (size_t)b * 11 * n + i;
    V3 c = V3{ncx, ncy, ncz};
    float reach = nrad + r + 1.0e-3f;
    if (b + 1 < nb) {
      const float *bn = bx + (size_t)11 * n;
      ncx = bn[0]; ncy = bn[(size_t)n]; ncz = bn[2 * (size_t)n]; nrad = bn[10 * (size_t)n];
    }
    float dx = fmaxf(fmaxf(lo.x - c.x, c.x - hi.x), 0.0f);
    float dy = fmaxf(fmaxf(lo.y - c.y, c.y - hi.y), 0.0f);
    float dz = fmaxf(fmaxf(lo.z - c.z, c.z - hi.z), 0.0f);
    if (dx * dx + dy * dy + dz * dz > reach * reach) continue;
    Q4 q = Q4{bx[3 * (size_t)n], bx[4 * (size_t)n], bx[5 * (size_t)n], bx[6 * (size_t)n]};
    V3 h = V3{bx[7 * (size_t)n], bx[8 * (size_t)n], bx[9 * (size_t)n]};
    for (int s = 0; s < k; ++s) {
      V3 p = V3{traj[(s * 3 + 0) * bd + tid], traj[(s * 3 + 1) * bd + tid], traj[(s * 3 + 2) * bd + tid]};
      hit = hit || sphere_hits_box(p, c, q, h, r2);
    }
  }
  return hit;
}

AGX_DEV float exp_reward(float mag, float ex, float v) { return mag * exp_cw(-(v * v) * ex); }
AGX_DEV float exp_penalty(float mag, float ex, float v) { return mag * (exp_cw(-(v * v) * ex) - 1.0f); }

// position_setpoint_task.py:245-282 on registers; returns the reward, ORs the distance crash
AGX_DEV float reward_position(const EnvState &s, Q4 qveh, V3 wb, V3 tgt, bool &crash) {
  V3 pe = quat_apply(conj(qveh), tgt - s.p);  // quat_apply_inverse
  float dist = norm(pe);
  float pos_reward = 3.0f * exp_cw(-8.0f * dist * dist) + 2.0f * exp_cw(-4.0f * dist * dist);
  float dist_reward = (20.0f - dist) / 40.0f;
  V3 up = quat_rotate(s.q, V3{0.0f, 0.0f, 1.0f});  // quat_axis(q, 2)
  float tilt = fabsf(1.0f - up.z);
  float up_reward = (1.0f / (0.1f + tilt * tilt)) * 0.2f;  // `0.2 / tensor` is tensor.reciprocal() * 0.2 in torch (eager and TorchScript)
  float spin = norm(wb);
  float ang_reward = (1.0f / (1.0f + spin * spin)) * 3.0f;
  float total = pos_reward + dist_reward + pos_reward * (up_reward + ang_reward);
  total = 1.0f * total;
  if (dist > 8.0f) crash = true;
  if (crash) total = -20.0f;
  return total;
}

// navigation_task.py:416-521 on registers
AGX_DEV float reward_navigation(const float *rp, float cpf, V3 pe, V3 ppe, float a0, float a2, float a3, float p0, float p2,
                                float p3, bool crash) {
  float mult = 1.0f + 2.0f * cpf;
  float dist = norm(pe), prev_dist = norm(ppe);
  float pos_reward = exp_reward(rp[0], rp[1], dist);
  float close_reward = exp_reward(rp[2], rp[3], dist);
  float closer = prev_dist - dist;
  float closer_reward = (closer > 0.0f) ? rp[4] * closer : 2.0f * rp[4] * closer;
  float dist_reward = (20.0f - dist) / 20.0f;
  float dx = a0 - p0, dz = a2 - p2, dyaw = a3 - p3;
  float diff_pen = exp_penalty(rp[5], rp[6], dx) + exp_penalty(rp[7], rp[8], dz) + exp_penalty(rp[9], rp[10], dyaw);
  float abs_pen = cpf * exp_penalty(rp[11], rp[12], a0) + cpf * exp_penalty(rp[13], rp[14], a2) +
                  cpf * exp_penalty(rp[15], rp[16], a3);
  float total_pen = diff_pen + abs_pen;
  float r = mult * (pos_reward + close_reward + closer_reward + dist_reward) + total_pen;
  if (crash) r = rp[17];
  return r;
}

// ---------------------------------------------------------------------------------------
// The env step: k fused physics sub-steps + (optionally) the task's reward / crash /
// truncation / reset set as an epilogue on the same registers.
// ---------------------------------------------------------------------------------------
// NavigationTask bookkeeping (navigation_task.py:311-326) on the registers of the step's epilogue -- the arithmetic of k_nav_bookkeeping
// (agx_task_glue.hip): near = norm(target - p) < radius.  `store`: this lane stores the env's flags (one lane per env).  Must be
// reached by every lane of the wave that runs the epilogue (the counters are bumped once per wave).
AGX_DEV void nav_bookkeeping_epilogue(const AgxTaskArgs &T, int i, bool store, V3 tgt, V3 p, bool crashed, bool trunc) {
  const bool near = norm(tgt - p) < T.success_radius;
  const bool succ = store && trunc && near && !crashed;
  const bool tout = store && trunc && !succ && !crashed;
  if (store) {
    T.successes[i] = succ ? 1 : 0;
    T.timeouts[i] = tout ? 1 : 0;
  }
  const unsigned long long act = __ballot(true);
  const unsigned long long ms = __ballot(succ), mc = __ballot(store && crashed), mt = __ballot(tout);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) {
    if (ms) atomicAdd(T.counters + 0, __popcll(ms));
    if (mc) atomicAdd(T.counters + 1, __popcll(mc));
    if (mt) atomicAdd(T.counters + 2, __popcll(mt));
  }
}

// SINGLE: exactly one sub-step (empty_env, BASELINE config 1/2): straight-line code, no loop-
// carried copies of the loop invariants.
// WIDE: the launch uses one-wave workgroups (n <= 65536 envs: at most one wave per SIMD is resident anyway), so the
// kernel is compiled for ONE wave per SIMD and may use the whole 512-entry register file: no spill in any variant.
// !WIDE: 256-thread workgroups at AGX_DYN_WAVES waves per SIMD for batches that fill the chip several times over.
// Waves per SIMD a straight-line (SINGLE, 256-thread) instance is compiled for.  With the SoA accesses as buffer accesses (SoaRef)
// the laws without Euler-angle feedback fit 128 VGPRs without a spill (position: 113; was 148 with 64-bit address pairs): 4 waves.
constexpr int env_step_single_waves(int M, int CTRL) {
  return (M <= 6 && (CTRL == AGX_CTRL_NONE || CTRL == AGX_CTRL_POSITION || CTRL == AGX_CTRL_FULLY_ACTUATED || CTRL == AGX_CTRL_WRENCH))
             ? AGX_DYN_WAVES_LEAN_LAWS
             : (CTRL == AGX_CTRL_ACCELERATION ? 2 : AGX_DYN_WAVES);  // (the acceleration law: 168-181 VGPRs, spills at 3 waves)
}
template <int M, int CTRL, bool SINGLE, bool WIDE>
__global__ void __launch_bounds__(WIDE ? 64 : 256, WIDE ? 1 : (SINGLE ? env_step_single_waves(M, CTRL) : AGX_DYN_WAVES_LOOP))
    k_env_step(AgxRobotParams P, AgxEnvBuffers B, int n, const float *__restrict__ actions_in, int k_arg, AgxTaskArgs T) {
  const int k = SINGLE ? 1 : k_arg;
  extern __shared__ float traj[];  // [k][3][blockDim] sub-step positions (only with obstacles)
  const int tid = threadIdx.x, bd = blockDim.x;
  const int i = blockIdx.x * bd + tid;
  bool reset = false;
  // peer push: one wave holds the step until the slot of its rows is free (flags loaded here, looked at when the kernel is done)
  if (blockIdx.x == 0 && tid < 64) push_publish_previous(B);
  const uint32_t push_peek = (blockIdx.x == 0 && tid < 64) ? push_wait_peek(B) : 0u;
  if (i < n) {
    const int A = P.num_actions;
    // AGX_LAUNCH_LEAN (launch_flags bit 2): the tensors that only exist to be LOOKED AT through the tensor dict are not
    // maintained -- Euler angles, vehicle-frame quaternion / velocity, robot_actions / robot_prev_actions (40 + 48 of the
    // 330 bytes an env moves per step); the body-frame velocities stay (the observation kernel reads them)
    const bool lean = (B.launch_flags & 4) != 0;
    EnvState s = load_state(B.state, n, i);
    float u[M], kT[M], tinc[M], tdec[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
      u[j] = AGX_AT(B.motor_thrust, j);
      kT[j] = P.use_rps ? AGX_AT(B.motor_kT, j) : 1.0f;
      tinc[j] = B.motor_tau_inc ? AGX_AT(B.motor_tau_inc, j) : P.tau_inc_uniform;
      tdec[j] = B.motor_tau_dec ? AGX_AT(B.motor_tau_dec, j) : P.tau_dec_uniform;
    }
    // EXTERNAL controller (a user class evaluated by the host between launches): actions_in is ITS OUTPUT, the body
    // wrench [N][6]; robot_actions / robot_prev_actions (A columns) are maintained by the host and only read here
    constexpr bool EXT = CTRL == AGX_CTRL_WRENCH;
    float a_in[AGX_MAX_ACTIONS], a_old[AGX_MAX_ACTIONS];
    // (the row index in 32 bits: n x 8 actions < 2^32.  The 64-bit multiply the compiler made of (size_t)i * A carried a
    // don't-care register into its high half -- one a state load was still writing -- and waited for that load first)
    const unsigned arow = (unsigned)i * (unsigned)(EXT ? 6 : A);
#pragma unroll
    for (int c = 0; c < AGX_MAX_ACTIONS; ++c) {
      a_in[c] = (c < (EXT ? 6 : A)) ? actions_in[arow + (unsigned)c] : 0.0f;
      a_old[c] = (c < A && !lean) ? AGX_AT(B.actions, c) : 0.0f;
    }
    Derived d{};
    if (k == 0 && T.kind != AGX_TASK_NONE) d = load_derived(B.derived, n, i);
    // What the bookkeeping / task epilogue reads is requested HERE, with the state.  Behind the stores of this kernel the
    // compiler cannot move a load up (the buffers may alias for all it knows), and each load issued down there is a memory
    // round trip of its own on the wave's critical path that also sits out every store in front of it (gfx9 counts loads and
    // stores in one vmcnt): step counter -> target -> previous error were three such trips per wave.
    const bool more_launches = (B.launch_flags & 2) != 0;  // (launch_flags: see below)
    const bool task_epilogue = T.kind != AGX_TASK_NONE && !more_launches;
    const int steps_in = B.sim_steps[i];
    const int crashed_in = (B.launch_flags & 1) ? B.crashes[i] : 0;
    V3 tgt{0, 0, 0}, ppe{0, 0, 0};
    if (task_epilogue) {
      tgt = V3{AGX_AT(T.target, 0), AGX_AT(T.target, 1), AGX_AT(T.target, 2)};
      if (T.kind != AGX_TASK_POSITION) ppe = V3{AGX_AT(T.pos_err, 0), AGX_AT(T.pos_err, 1), AGX_AT(T.pos_err, 2)};
    }
    float a_prev_in[AGX_MAX_ACTIONS];  // robot_prev_actions as the last step left them (a k = 0 launch or an external controller reads them)
#pragma unroll
    for (int c = 0; c < AGX_MAX_ACTIONS; ++c) a_prev_in[c] = ((k == 0 || EXT) && c < A && !lean) ? AGX_AT(B.prev_actions, c) : 0.0f;
    // the gains LAST: with uniform gains (B.gains null) the registers they are moved into are the ones the other arm loads into,
    // and the compiler waits for every load in flight before the move -- behind the last load that wait costs nothing
    Gains g{};
    if (CTRL != AGX_CTRL_NONE && CTRL != AGX_CTRL_WRENCH) g = B.gains ? load_gains(B.gains, n, i) : uniform_gains(P);
    Wrench wc{V3{0, 0, 0}, V3{0, 0, 0}};
    const bool root_link = P.root_link_mode != 0;
    const int sub_base = (B.launch_flags >> 8) & 0xFF;  // physics sub-step this launch starts at (split env steps)
    bool has_drag = false;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      has_drag = has_drag || P.lin_drag_linear[c] != 0.0f || P.lin_drag_quadratic[c] != 0.0f || P.ang_drag_linear[c] != 0.0f ||
                 P.ang_drag_quadratic[c] != 0.0f;
    V3 tlo = s.p, thi = s.p;
    constexpr bool kLawReadsNoAngles = CTRL == AGX_CTRL_POSITION || CTRL == AGX_CTRL_FULLY_ACTUATED || CTRL == AGX_CTRL_NONE || CTRL == AGX_CTRL_WRENCH;
    for (int sub = 0; sub < k; ++sub) {
      d = (kLawReadsNoAngles && lean) ? update_states_lean(s) : update_states(s);
      float a[AGX_MAX_ACTIONS];
#pragma unroll
      for (int c = 0; c < AGX_MAX_ACTIONS; ++c) a[c] = clamp_minmax(a_in[c], -10.0f, 10.0f);  // clip_actions
      // EXTERNAL ROBOT (AGX_LAUNCH_BODY_WRENCH, host-evaluated robot.step()): actions_in is the net body wrench itself
      const bool body_wrench = EXT && (B.launch_flags & AGX_LAUNCH_BODY_WRENCH) != 0;  // wave-uniform
      if (CTRL == AGX_CTRL_NONE) {
#pragma unroll
        for (int j = 0; j < M; ++j) u[j] = motor_update(P, a[j], u[j], kT[j], tinc[j], tdec[j]);
      } else if (!body_wrench) {
        if (EXT) wc = Wrench{V3{a_in[0], a_in[1], a_in[2]}, V3{a_in[3], a_in[4], a_in[5]}};  // as handed in, not clipped
      else wc = run_controller<CTRL>(P, s, d, a, g);
        const float w6[6] = {wc.f.x, wc.f.y, wc.f.z, wc.t.x, wc.t.y, wc.t.z};
#pragma unroll
        for (int j = 0; j < M; ++j) {
          float r = 0.0f;
#pragma unroll
          for (int c = 0; c < 6; ++c) r += P.alloc_pinv[6 * j + c] * w6[c];
          u[j] = motor_update(P, r, u[j], kT[j], tinc[j], tdec[j]);
        }
      }
      float bw[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < M; ++j) acc += (root_link ? P.alloc[M * r + j] : P.wrench_map[M * r + j]) * u[j];
        bw[r] = body_wrench ? a_in[r] : acc;
      }
      // The ROOT link's entry of robot_force / robot_torque_tensor: the allocator's wrench in root-link mode, else 0.
      // simulate_drag (base_multirotor.py:260-285; pre-physics body velocities) and apply_disturbance (:213-234) accumulate
      // into it with `+=`, in that order; the net wrench on the rigid composite is the motor links' sum plus that entry.  So
      // with forces at the motor links, drag AND disturbance are summed first and added to the links' sum once (`root`);
      // with one of the two, or in root-link mode, that is the running sum below.  All-zero drag coefficients (base
      // quadrotor) add +-0 to every component: skipped (a scalar test of kernel arguments).
      const bool any_dist = !body_wrench && (B.disturb != nullptr || B.disturb_prob > 0.0f);
      const bool split_root = !root_link && has_drag && any_dist;  // wave-uniform
      float dr[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, di[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (has_drag) {
        float vbn = norm(d.vbody);
        dr[0] = (-P.lin_drag_linear[0] * d.vbody.x) + (-P.lin_drag_quadratic[0] * vbn * d.vbody.x);
        dr[1] = (-P.lin_drag_linear[1] * d.vbody.y) + (-P.lin_drag_quadratic[1] * vbn * d.vbody.y);
        dr[2] = (-P.lin_drag_linear[2] * d.vbody.z) + (-P.lin_drag_quadratic[2] * vbn * d.vbody.z);
        dr[3] = (-P.ang_drag_linear[0] * d.wbody.x) + (-P.ang_drag_quadratic[0] * fabsf(d.wbody.x) * d.wbody.x);
        dr[4] = (-P.ang_drag_linear[1] * d.wbody.y) + (-P.ang_drag_quadratic[1] * fabsf(d.wbody.y) * d.wbody.y);
        dr[5] = (-P.ang_drag_linear[2] * d.wbody.z) + (-P.ang_drag_quadratic[2] * fabsf(d.wbody.z) * d.wbody.z);
      }
      if (B.disturb) {  // draws supplied by the host
        const float *dd = B.disturb + (size_t)(sub_base + sub) * 7 * n + i;
        float occ = dd[0];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          float lo = -B.disturb_max[c], hi = B.disturb_max[c];
          di[c] = ((hi - lo) * dd[(size_t)(1 + c) * n] + lo) * occ;
        }
      } else if (B.disturb_prob > 0.0f) {  // same, drawn in place: 7 uniforms per env and sub-step
        float ud[7];
        rng_fill<7>(B.rng_seed, B.env_index_base + i, agx::step_index(B), RNG_DISTURB + sub_base + sub, ud);
        float occ = ud[0] < B.disturb_prob ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          float lo = -B.disturb_max[c], hi = B.disturb_max[c];
          di[c] = ((hi - lo) * ud[1 + c] + lo) * occ;
        }
      }
      if (!body_wrench && (has_drag || any_dist)) {  // an absent term is +0: x + 0 = x
#pragma unroll
        for (int c = 0; c < 6; ++c) bw[c] = split_root ? bw[c] + (dr[c] + di[c]) : (bw[c] + dr[c]) + di[c];
      }
      if (B.body_force && sub == k - 1) {  // what the IMU's force sensor sees (agx_imu_update)
        AGX_AT(B.body_force, 0) = bw[0]; AGX_AT(B.body_force, 1) = bw[1]; AGX_AT(B.body_force, 2) = bw[2];
      }
      integrate(P, s, V3{bw[0], bw[1], bw[2]}, V3{bw[3], bw[4], bw[5]});
      if (B.boxes) {
        traj[(sub * 3 + 0) * bd + tid] = s.p.x;
        traj[(sub * 3 + 1) * bd + tid] = s.p.y;
        traj[(sub * 3 + 2) * bd + tid] = s.p.z;
        if (sub == 0) { tlo = s.p; thi = s.p; }
        tlo = V3{fminf(tlo.x, s.p.x), fminf(tlo.y, s.p.y), fminf(tlo.z, s.p.z)};
        thi = V3{fmaxf(thi.x, s.p.x), fmaxf(thi.y, s.p.y), fmaxf(thi.z, s.p.z)};
      }
    }
    // EnvManager.reset_tensors + compute_observations (env_manager.py:342-344, 358-362)
    // launch_flags (external controllers run ONE launch per physics sub-step): bit 0 = an earlier launch of this env
    // step already ran: accumulate its crash flag; bit 1 = more launches follow: no step counter / truncation / task epilogue
    bool crashed = crashed_in != 0;
    if (B.boxes && k > 0) crashed = collide_trajectory(B.boxes, B.num_boxes, n, i, traj, k, bd, tid, tlo, thi, P.collision_radius) || crashed;
    store_state(B.state, n, i, s);
    if (k > 0) {
      if (lean) store_body_velocities(B.derived, n, i, d);
      else store_derived(B.derived, n, i, d);
#pragma unroll
      for (int j = 0; j < M; ++j) AGX_AT(B.motor_thrust, j) = u[j];
      if (B.wrench_cmd) {
        AGX_AT(B.wrench_cmd, 0) = wc.f.x; AGX_AT(B.wrench_cmd, 1) = wc.f.y; AGX_AT(B.wrench_cmd, 2) = wc.f.z;
        AGX_AT(B.wrench_cmd, 3) = wc.t.x; AGX_AT(B.wrench_cmd, 4) = wc.t.y; AGX_AT(B.wrench_cmd, 5) = wc.t.z;
      }
    }
    // RobotManagerIGE.pre_physics_step runs every sub-step: prev <- cur, cur <- action
    float a_cur[AGX_MAX_ACTIONS], a_prev[AGX_MAX_ACTIONS];
#pragma unroll
    for (int c = 0; c < AGX_MAX_ACTIONS; ++c) {
      a_cur[c] = (k > 0 && !EXT) ? a_in[c] : a_old[c];
      a_prev[c] = (k >= 2 && !EXT) ? a_in[c] : ((k == 1 && !EXT) ? a_old[c] : a_prev_in[c]);
      if (c < A && k > 0 && !EXT && !lean) {
        AGX_AT(B.prev_actions, c) = a_prev[c];
        AGX_AT(B.actions, c) = a_cur[c];
      }
    }
    const int steps = steps_in + (more_launches ? 0 : 1);
    if (!more_launches) B.sim_steps[i] = steps;
    bool trunc = false;
    if (task_epilogue) {
      float rew;
      if (T.kind == AGX_TASK_POSITION) {
        rew = reward_position(s, d.qveh, d.wbody, tgt, crashed);
      } else {
        AGX_AT(T.prev_pos_err, 0) = ppe.x; AGX_AT(T.prev_pos_err, 1) = ppe.y; AGX_AT(T.prev_pos_err, 2) = ppe.z;
        V3 pe = quat_rotate_inverse(d.qveh, tgt - s.p);
        AGX_AT(T.pos_err, 0) = pe.x; AGX_AT(T.pos_err, 1) = pe.y; AGX_AT(T.pos_err, 2) = pe.z;
        rew = reward_navigation(T.rp, T.curriculum_progress, pe, ppe, a_cur[0], a_cur[2], a_cur[3], a_prev[0], a_prev[2],
                                a_prev[3], crashed);
      }
      T.reward[i] = rew;
      trunc = steps > T.episode_len;
      reset = (crashed && T.reset_on_collision) || trunc;
      B.reset_mask[i] = reset ? 1 : 0;
      if (T.successes) nav_bookkeeping_epilogue(T, i, true, tgt, s.p, crashed, trunc);  // (wave-uniform pointer test)
    }
    B.crashes[i] = crashed ? 1 : 0;
    if (!more_launches) B.truncations[i] = trunc ? 1 : 0;
  }
  if (T.kind != AGX_TASK_NONE && __ballot(reset) != 0ull && (tid & 63) == 0) atomicOr(B.reset_flag + B.flag_parity, 1);
  if (blockIdx.x == 0 && tid < 64) push_wait_finish(B, push_peek);
}

// ---------------------------------------------------------------------------------------
// The one-sub-step env step of the Lee position controller on a quadrotor (BASELINE configs 1/2) with FOUR lanes per
// env (agx_quad_math.h): a wave carries 16 envs, 8192 envs are 512 waves on the 1024 SIMDs instead of 128, and a wave
// issues about half the vector instructions of the one-lane-per-env kernel.  Every value is produced by the same IEEE
// operations in the same order as in k_env_step<4, AGX_CTRL_POSITION, true, .>; the GPU parity tests run against the
// CPU restatement through this kernel.  Not covered (the launcher falls back to k_env_step): obstacles, drag, disturbances,
// split launches, other controllers / motor counts.
// ---------------------------------------------------------------------------------------
namespace q4 = quad;
AGX_DEV unsigned long long vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// BaseMultirotor.update_states (base_multirotor.py:287-294) of one env on its lane quad
struct QuadDerived {
  float euler, qveh, vveh, vbody, wbody;
};
// `extra` / `esn`, `ecs`: the half-yaw's sine and cosine are needed in lanes 2 and 3 only, so lanes 0 and 1 of the same
// evaluation take another angle of the caller's (the position law's yaw set-point) and hand back its sine / cosine
AGX_DEV QuadDerived update_states_quad(float q, float v, float w, float extra, float &esn, float &ecs) {
  const int l = q4::lane_in_quad();
  QuadDerived d;
  const float e = q4::euler_xyz_0_2pi(q);
  d.euler = ssa(e);
  float sy, cy;
  const float half_yaw = (q4::bc<2>(e) * 1.0f) * 0.5f;  // vehicle_frame_quat_from_quat: quat_from_yaw
  sincos_bounded(l < 2 ? extra : half_yaw, sy, cy);
  esn = sy;
  ecs = cy;
  d.qveh = l == 2 ? sy : (l == 3 ? cy : 0.0f);
  d.vveh = q4::quat_rotate_inverse(d.qveh, v);
  d.vbody = q4::quat_rotate_inverse(q, v);
  d.wbody = q4::quat_rotate_inverse(q, w);
  return d;
}
AGX_DEV QuadDerived update_states_quad(float q, float v, float w) {
  float sn, cs;
  return update_states_quad(q, v, w, 0.0f, sn, cs);
}

// Per-lane constants of the quad kernels: component l of a vector, row l of a matrix, motors l (and l + 4 of an 8-motor robot)
// (indexed kernel-argument loads)
template <int M>
struct QuadConsts {
  float grav, in0, in1, in2, ii0, ii1, ii2, pinv[M / 4][6], mapf[M], mapt[M], mass, dt;
};
template <int M>
AGX_DEV QuadConsts<M> load_quad_consts(const AgxRobotParams &P, int l, int l3) {
  QuadConsts<M> C;
  C.grav = P.gravity[l3];
  C.in0 = P.inertia[3 * l3 + 0]; C.in1 = P.inertia[3 * l3 + 1]; C.in2 = P.inertia[3 * l3 + 2];
  C.ii0 = P.inertia_inv[3 * l3 + 0]; C.ii1 = P.inertia_inv[3 * l3 + 1]; C.ii2 = P.inertia_inv[3 * l3 + 2];
#pragma unroll
  for (int h = 0; h < M / 4; ++h)
#pragma unroll
    for (int c = 0; c < 6; ++c) C.pinv[h][c] = P.alloc_pinv[6 * (l + 4 * h) + c];  // motor l + 4 h
  const float *wmap = P.root_link_mode != 0 ? P.alloc : P.wrench_map;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    C.mapf[j] = wmap[M * l3 + j];        // force row l
    C.mapt[j] = wmap[M * (3 + l3) + j];  // torque row l
  }
  C.mass = P.mass;
  C.dt = P.dt;
  return C;
}
// f . third column of quat_to_rotmat(q) = (2 (xz + yw), 2 (yz - xw), 1 - 2 (xx + yy)): the thrust command of the Lee laws
AGX_DEV float quad_thrust_along_body_z(float q, float f, int l) {
  const float t1 = q * q4::bc<2>(q), t2 = q4::perm<1, 0, 2, 3>(q) * q4::bc<3>(q);
  const float c2a = 2.0f * (l == 1 ? t1 - t2 : t1 + t2);
  const float sqq = q * q;
  const float m22 = 1.0f - 2.0f * (q4::bc<0>(sqq) + q4::bc<1>(sqq));
  return q4::dot3(f, l == 2 ? m22 : c2a);
}
// base_lee_controller.py:173-194 (desired_orientation_pos_vel)
// (sy, cy: sine and cosine of the yaw set-point, each valid in the lane that uses it -- cy in lane 0, sy in lane 1)
AGX_DEV float quad_desired_orientation_pos_vel_sc(float f, float sy, float cy, int l) {
  const float b3 = fdiv(f, q4::norm3(f));
  const float tmp = l == 0 ? cy : (l == 1 ? sy : 0.0f);
  const float cb = q4::cross3(b3, tmp);
  const float b2 = fdiv(cb, q4::norm3(cb));
  const float b1 = q4::cross3(b2, b3);
  return q4::rotmat_cols_to_quat(b1, b2, b3);
}
AGX_DEV float quad_desired_orientation_pos_vel(float f, float yaw, int l) {
  float sy, cy;
  sincos_bounded(yaw, sy, cy);
  return quad_desired_orientation_pos_vel_sc(f, sy, cy, l);
}
// base_lee_controller.py:136-154 (compute_body_torque); ZERO_RATE: the angular-velocity set-point is the constant 0
template <bool ZERO_RATE, int M>
AGX_DEV float quad_body_torque(const QuadConsts<M> &C, float q, float qd, float wb, float wsp, float kr, float kw, int l) {
  const float qe = q4::quat_mul(q4::conj(q), qd);
  const float pp = q4::rot1(qe) * q4::rot2(qe);  // (yz, zx, xy)
  const float pw = qe * q4::bc<3>(qe);           // (xw, yw, zw)
  const float mp = 2.0f * (pp + pw);             // (m21, m02, m10)
  const float mm = 2.0f * (pp - pw);             // (m12, m20, m01)
  const float rot_err = 0.5f * (l == 1 ? mm - mp : -(mp - mm));
  const float jw = (C.in0 * q4::bc<0>(wb) + C.in1 * q4::bc<1>(wb)) + C.in2 * q4::bc<2>(wb);
  const float ff = q4::cross3(wb, jw);
  const float we = ZERO_RATE ? wb : wb - q4::quat_rotate(qe, wsp);
  return ((-kr) * rot_err - kw * we) + ff;
}
// allocation (lane l = motors l, l + 4) + motor model + body wrench (lane l = row l of the force / of the torque);
// `force`: the commanded body force, (0, 0, thrust) for the Lee laws
template <int M>
AGX_DEV void quad_allocate(const AgxRobotParams &P, const QuadConsts<M> &C, float force, float torque, float (&u)[M / 4],
                           const float (&kT)[M / 4], const float (&tinc)[M / 4], const float (&tdec)[M / 4], float &fb, float &tb) {
  const float w0 = q4::bc<0>(force), w1 = q4::bc<1>(force), w2 = q4::bc<2>(force);
  const float w3 = q4::bc<0>(torque), w4 = q4::bc<1>(torque), w5 = q4::bc<2>(torque);
#pragma unroll
  for (int h = 0; h < M / 4; ++h) {
    float r = 0.0f;
    r += C.pinv[h][0] * w0;
    r += C.pinv[h][1] * w1;
    r += C.pinv[h][2] * w2;
    r += C.pinv[h][3] * w3;
    r += C.pinv[h][4] * w4;
    r += C.pinv[h][5] * w5;
    u[h] = motor_update(P, r, u[h], kT[h], tinc[h], tdec[h]);
  }
  fb = 0.0f;
  tb = 0.0f;
#pragma unroll
  for (int h = 0; h < M / 4; ++h) {
    const float u0 = q4::bc<0>(u[h]), u1 = q4::bc<1>(u[h]), u2 = q4::bc<2>(u[h]), u3 = q4::bc<3>(u[h]);
    fb += C.mapf[4 * h + 0] * u0; fb += C.mapf[4 * h + 1] * u1; fb += C.mapf[4 * h + 2] * u2; fb += C.mapf[4 * h + 3] * u3;
    tb += C.mapt[4 * h + 0] * u0; tb += C.mapt[4 * h + 1] * u1; tb += C.mapt[4 * h + 2] * u2; tb += C.mapt[4 * h + 3] * u3;
  }
}
// the rigid-body update (integrate(), DESIGN.md "integrator") on the quad
template <int M>
AGX_DEV void quad_integrate(const AgxRobotParams &P, const QuadConsts<M> &C, float &p, float &q, float &v, float &w, float fb, float tb,
                            int l) {
  const float dt = C.dt;
  const float fw = q4::quat_rotate(q, fb);
  const float wbi = q4::quat_rotate_inverse(q, w);
  const float jwi = (C.in0 * q4::bc<0>(wbi) + C.in1 * q4::bc<1>(wbi)) + C.in2 * q4::bc<2>(wbi);
  const float rhs = tb - q4::cross3(wbi, jwi);
  const float dwb = (C.ii0 * q4::bc<0>(rhs) + C.ii1 * q4::bc<1>(rhs)) + C.ii2 * q4::bc<2>(rhs);
  const float wb_new = wbi + dt * dwb;
  float w_new = q4::quat_rotate(q, wb_new);
  float v_new = v + dt * fdiv(fw, C.mass);
  v_new = v_new + C.grav * dt;
  const float ml = fmaxf(1.0f - P.linear_damping * dt, 0.0f);
  const float ma = fmaxf(1.0f - P.angular_damping * dt, 0.0f);
  v_new = v_new * ml;
  w_new = w_new * ma;
  const float v2 = q4::dot3(v_new, v_new), w2 = q4::dot3(w_new, w_new);
  if (v2 > P.max_linear_velocity * P.max_linear_velocity) v_new = v_new * fdiv(P.max_linear_velocity, fsqrt(v2));
  if (w2 > P.max_angular_velocity * P.max_angular_velocity) w_new = w_new * fdiv(P.max_angular_velocity, fsqrt(w2));
  p = p + v_new * dt;
  const float wm2 = q4::dot3(w_new, w_new);
  if (wm2 != 0.0f) {
    const float wm = fsqrt(wm2);
    const float half = dt * wm * 0.5f;
    float sn, cs;
    sincos_bounded(half, sn, cs);
    const float sc = fdiv(sn, wm);
    const float x1 = w_new * sc;  // (x1, y1, z1)
    // (x1 w + y1 z - z1 y, y1 w + z1 x - x1 z, z1 w + x1 y - y1 x, -(x1 x) - y1 y - z1 z)
    const float r3 = (x1 * q4::bc<3>(q) + q4::rot1(x1) * q4::rot2(q)) - q4::rot2(x1) * q4::rot1(q);
    const float xq = x1 * q;
    const float rw = (-q4::bc<0>(xq) - q4::bc<1>(xq)) - q4::bc<2>(xq);
    float rq = l == 3 ? rw : r3;
    rq += q * cs;
    const float nn = fsqrt(q4::dot4(rq, rq));
    q = fdiv(rq, nn);
  }
  v = v_new;
  w = w_new;
}

__global__ void __launch_bounds__(64, 1)
    k_env_step_quad_position(AgxRobotParams P, AgxEnvBuffers B, int n, const float *__restrict__ actions_in, AgxTaskArgs T) {
  const int tid = threadIdx.x;
  const int l = tid & 3, l3 = l < 3 ? l : 2;  // component of a 4-vector / of a 3-vector (lane 3 repeats z: don't care)
  const int i = blockIdx.x * 16 + (tid >> 2);  // env
  const unsigned ol = ((unsigned)l * (unsigned)n + (unsigned)i) * 4u, ol3 = ((unsigned)l3 * (unsigned)n + (unsigned)i) * 4u;  // AGX_QAT
  bool reset = false;
  if (blockIdx.x == 0) push_publish_previous(B);  // peer push: the previous step's rows have landed everywhere
  const uint32_t push_peek = blockIdx.x == 0 ? push_wait_peek(B) : 0u;  // ... and this step's slot: looked at when the kernel is done
  if (i < n) {
    // ---- loads: one instruction per vector
    float p = AGX_QAT(B.state, 0, ol3), q = AGX_QAT(B.state, 3, ol), v = AGX_QAT(B.state, 7, ol3), w = AGX_QAT(B.state, 10, ol3);
    float u[1] = {AGX_QAT(B.motor_thrust, 0, ol)};  // motor l
    const float kT[1] = {P.use_rps ? AGX_QAT(B.motor_kT, 0, ol) : 1.0f};
    const float tinc[1] = {B.motor_tau_inc ? AGX_QAT(B.motor_tau_inc, 0, ol) : P.tau_inc_uniform};
    const float tdec[1] = {B.motor_tau_dec ? AGX_QAT(B.motor_tau_dec, 0, ol) : P.tau_dec_uniform};
    const float a_in = actions_in[(size_t)i * 4 + l];
    const float a_old = AGX_QAT(B.actions, 0, ol);
    const float kp = B.gains ? AGX_QAT(B.gains, 0, ol3) : P.gains_uniform[0 + l3];
    const float kv = B.gains ? AGX_QAT(B.gains, 3, ol3) : P.gains_uniform[3 + l3];
    const float kr = B.gains ? AGX_QAT(B.gains, 6, ol3) : P.gains_uniform[6 + l3];
    const float kw = B.gains ? AGX_QAT(B.gains, 9, ol3) : P.gains_uniform[9 + l3];
    // what the task epilogue reads is requested HERE, with the state: behind the stores below the compiler cannot move a load up
    // (the buffers may alias for all it knows), and a load issued there is a second memory round trip on the kernel's critical
    // path -- one that also waits for every store in front of it (gfx9 counts loads and stores in the same vmcnt)
    const int steps_in = B.sim_steps[i];
    const float tgt = T.kind == AGX_TASK_POSITION ? AGX_QAT(T.target, 0, ol3) : 0.0f;
    const QuadConsts<4> C = load_quad_consts<4>(P, l, l3);

    // ---- update_states + controller (position_control.py:20-51)
    const float a = clamp_minmax(a_in, -10.0f, 10.0f);  // clip_actions
    float sy_sp, cy_sp;  // of the yaw set-point (lanes 0, 1), out of the evaluation that serves the vehicle-frame quaternion
    const QuadDerived d = update_states_quad(q, v, w, q4::bc<3>(a), sy_sp, cy_sp);
    // compute_acceleration (velocity set-point 0): kp (sp - p) + kv (0 - v)
    const float pe = a - p;
    const float ve = 0.0f - v;
    const float acc = kp * pe + kv * ve;
    const float f = (acc - C.grav) * C.mass;
    const float fz = quad_thrust_along_body_z(q, f, l);
    const float qd = quad_desired_orientation_pos_vel_sc(f, sy_sp, cy_sp, l);
    const float torque = quad_body_torque<true>(C, q, qd, d.wbody, 0.0f, kr, kw, l);

    // ---- allocation + motor model + body wrench, rigid-body update
    float fb, tb;
    quad_allocate<4>(P, C, l == 2 ? fz : 0.0f, torque, u, kT, tinc, tdec, fb, tb);
    if (B.body_force && l < 3) AGX_QAT(B.body_force, 0, ol) = fb;
    quad_integrate(P, C, p, q, v, w, fb, tb, l);

    // ---- stores: state, derived, motors, controller output, actions
    if (l < 3) AGX_QAT(B.state, 0, ol) = p;
    AGX_QAT(B.state, 3, ol) = q;
    if (l < 3) {
      AGX_QAT(B.state, 7, ol) = v;
      AGX_QAT(B.state, 10, ol) = w;
      AGX_QAT(B.derived, 0, ol) = d.euler;
      AGX_QAT(B.derived, 7, ol) = d.vveh;
      AGX_QAT(B.derived, 10, ol) = d.vbody;
      AGX_QAT(B.derived, 13, ol) = d.wbody;
    }
    AGX_QAT(B.derived, 3, ol) = d.qveh;
    AGX_QAT(B.motor_thrust, 0, ol) = u[0];
    if (B.wrench_cmd) {
      if (l < 3) {
        AGX_QAT(B.wrench_cmd, 0, ol) = l == 2 ? fz : 0.0f;
        AGX_QAT(B.wrench_cmd, 3, ol) = torque;
      }
    }
    AGX_QAT(B.prev_actions, 0, ol) = a_old;  // RobotManagerIGE.pre_physics_step: prev <- cur, cur <- action
    AGX_QAT(B.actions, 0, ol) = a_in;

    // ---- EnvManager bookkeeping + the position task's reward / truncation / reset set (position_setpoint_task.py:245-282)
    const int steps = steps_in + 1;
    bool crashed = false, trunc = false;
    float rew = 0.0f;
    if (T.kind == AGX_TASK_POSITION) {
      const float pe_t = q4::quat_apply(q4::conj(d.qveh), tgt - p);  // quat_apply_inverse
      const float dist = q4::norm3(pe_t);
      // 3 exp(-8 d^2) + 2 exp(-4 d^2): both exponentials in one evaluation (lanes 0 / 1)
      const float ex = exp_cw((l == 0 ? -8.0f : -4.0f) * dist * dist);
      const float pos_reward = 3.0f * q4::bc<0>(ex) + 2.0f * q4::bc<1>(ex);
      const float dist_reward = (20.0f - dist) / 40.0f;
      const float axis_z = l == 2 ? 1.0f : 0.0f;
      const float up = q4::bc<2>(q4::quat_rotate(q, axis_z));  // quat_axis(q, 2).z
      const float tilt = fabsf(1.0f - up);
      const float spin = q4::norm3(d.wbody);
      // 0.2 / (0.1 + tilt^2) = reciprocal * 0.2 (torch's scalar / tensor) and (1 / (1 + spin^2)) * 3: one division (lanes 0 / 1)
      const float quo = (1.0f / (l == 0 ? 0.1f + tilt * tilt : 1.0f + spin * spin)) * (l == 0 ? 0.2f : 3.0f);
      const float up_reward = q4::bc<0>(quo);
      const float ang_reward = q4::bc<1>(quo);
      float total = pos_reward + dist_reward + pos_reward * (up_reward + ang_reward);
      total = 1.0f * total;
      if (dist > 8.0f) crashed = true;
      if (crashed) total = -20.0f;
      rew = total;
      trunc = steps > T.episode_len;
      reset = (crashed && T.reset_on_collision) || trunc;
    }
    if (l == 0) {
      B.sim_steps[i] = steps;
      if (T.kind == AGX_TASK_POSITION) {
        T.reward[i] = rew;
        B.reset_mask[i] = reset ? 1 : 0;
      }
      B.crashes[i] = crashed ? 1 : 0;
      B.truncations[i] = trunc ? 1 : 0;
    }
  }
  if (T.kind != AGX_TASK_NONE && __ballot(reset) != 0ull && (tid & 63) == 0) atomicOr(B.reset_flag + B.flag_parity, 1);
  if (blockIdx.x == 0) push_wait_finish(B, push_peek);
}

// ---------------------------------------------------------------------------------------
// Four lanes per env for the sub-step LOOP (BASELINE configs 2 / 4, the LiDAR navigation task, the reference's default
// attitude-controlled position task): quadrotor, any of the six Lee laws, k sub-steps, obstacles, device disturbance
// draws, task epilogue.  Same contract as k_env_step_quad_position: per component the IEEE operations of
// k_env_step<4, CTRL, false, .> in the same order.  The obstacle test splits the env's boxes over the four lanes (the flag
// is a boolean OR: any order).
// ---------------------------------------------------------------------------------------
// base_lee_controller.py:201-215 on the quad: (1 0 -sp; 0 cr sr cp; 0 -sr cr cp) (0, 0, rz)
AGX_DEV float euler_rates_to_body_rates_quad(float euler, float rz) {
  const int l = q4::lane_in_quad();
  float sn, cs;
  sincos_bounded(euler, sn, cs);  // lane 0: roll, lane 1: pitch
  const float sr = q4::bc<0>(sn), cr = q4::bc<0>(cs), sp = q4::bc<1>(sn), cp = q4::bc<1>(cs);
  const float m0 = q4::by_lane(l, 1.0f, 0.0f, 0.0f);
  const float m1 = q4::by_lane(l, 0.0f, cr, -sr);
  const float m2 = q4::by_lane(l, -sp, sr * cp, cr * cp);
  return (m0 * 0.0f + m1 * 0.0f) + m2 * rz;
}
// utils/math.py:156-172 (quat_from_euler_xyz) with (roll, pitch, yaw) in lanes 0..2 of `ang`
AGX_DEV float quat_from_euler_quad(float ang) {
  const int l = q4::lane_in_quad();
  float sn, cs;
  sincos_bounded(ang * 0.5f, sn, cs);
  const float sr = q4::bc<0>(sn), cr = q4::bc<0>(cs), sp = q4::bc<1>(sn), cp = q4::bc<1>(cs), sy = q4::bc<2>(sn), cy = q4::bc<2>(cs);
  // x: cy sr cp - sy cr sp   y: cy cr sp + sy sr cp   z: sy cr cp - cy sr sp   w: cy cr cp + sy sr sp
  const float a1 = l == 2 ? sy : cy, b1 = l == 0 ? sr : cr, c1 = l == 1 ? sp : cp;
  const float a2 = l == 2 ? cy : sy, b2 = l == 0 ? cr : sr, c2 = l == 1 ? cp : sp;
  const float t1 = a1 * b1 * c1, t2 = a2 * b2 * c2;
  return (l == 0 || l == 2) ? t1 - t2 : t1 + t2;
}

// One env's control law on its lane quad (control/controllers/*.py; run_controller<CTRL> above is the one-lane form):
// commanded body force ((0, 0, thrust) for the Lee laws) and body torque from the clipped action `a` (a0..a3 in lanes 0..3;
// the fully actuated law: position set-point in `a`, orientation set-point xyzw in `a2`).
template <int CTRL, int M>
AGX_DEV void quad_controller(const AgxRobotParams &P, const QuadConsts<M> &C, float p, float q, float v, const QuadDerived &d, float a,
                             float a2, float kp, float kv, float kr, float kw, int l, float &force, float &torque) {
  const float yaw = q4::bc<2>(d.euler);
  float fz = 0.0f;
  if (CTRL == AGX_CTRL_FULLY_ACTUATED) {  // fully_actuated_control.py:14-32
    float nq = sqrtf(q4::dot4(a2, a2));
    nq = nq < 1e-9f ? 1e-9f : nq;
    const float qd = a2 / nq;
    const float acc = kp * (a - p) + kv * (0.0f - v);
    const float f = (acc - C.grav) * C.mass;
    force = q4::quat_rotate_inverse(q, f);
    torque = quad_body_torque<true>(C, q, qd, d.wbody, 0.0f, kr, kw, l);
    return;
  }
  if (CTRL == AGX_CTRL_POSITION || CTRL == AGX_CTRL_VELOCITY || CTRL == AGX_CTRL_VEL_STEERING) {
    float acc;
    if (CTRL == AGX_CTRL_POSITION) {  // position_control.py:20-51: kp (sp - p) + kv (0 - v)
      acc = kp * (a - p) + kv * (0.0f - v);
    } else {  // velocity_control.py:18-51, velocity_steeing_angle_controller.py:15-45: set-point = the current position
      const float sp_vel_w = q4::quat_rotate(d.qveh, a);  // (a0, a1, a2) in the vehicle frame
      acc = kp * (p - p) + kv * (sp_vel_w - v);
    }
    const float f = (acc - C.grav) * C.mass;
    fz = quad_thrust_along_body_z(q, f, l);
    const float qd = quad_desired_orientation_pos_vel(f, CTRL == AGX_CTRL_VELOCITY ? yaw : q4::bc<3>(a), l);
    if (CTRL == AGX_CTRL_POSITION) {
      torque = quad_body_torque<true>(C, q, qd, d.wbody, 0.0f, kr, kw, l);
    } else {
      float wsp = euler_rates_to_body_rates_quad(d.euler, CTRL == AGX_CTRL_VELOCITY ? q4::bc<3>(a) : 0.0f);
      if (l == 2) wsp = fminf(fmaxf(wsp, -P.max_yaw_rate), P.max_yaw_rate);
      torque = quad_body_torque<false>(C, q, qd, d.wbody, wsp, kr, kw, l);
    }
  } else if (CTRL == AGX_CTRL_ACCELERATION) {  // acceleration_control.py:16-45
    const float f = (a - C.grav) * C.mass;
    fz = quad_thrust_along_body_z(q, f, l);
    // desired_orientation_forces_yaw(f, yaw): pitch = atan2(f.x, f.z), roll = atan2(-f.y, sqrt(f.z^2 + f.x^2))
    const float fx = q4::bc<0>(f), fy = q4::bc<1>(f), fzc = q4::bc<2>(f);
    const float num = q4::by_lane(l, -fy, fx, 0.0f);
    const float den = q4::by_lane(l, sqrtf(fzc * fzc + fx * fx), fzc, 1.0f);
    const float ang = atan2_cw(num, den);
    const float qd = quat_from_euler_quad(l == 2 ? yaw : ang);
    float wsp = euler_rates_to_body_rates_quad(d.euler, q4::bc<3>(a));
    if (l == 2) wsp = fminf(fmaxf(wsp, -P.max_yaw_rate), P.max_yaw_rate);
    torque = quad_body_torque<false>(C, q, qd, d.wbody, wsp, kr, kw, l);
  } else if (CTRL == AGX_CTRL_ATTITUDE) {  // attitude_control.py:16-43
    const float g0 = P.gravity[0], g1 = P.gravity[1], g2 = P.gravity[2];
    fz = (q4::bc<0>(a) + 1.0f) * C.mass * norm(V3{g0, g1, g2});  // torch.norm(gravity)
    float wsp = euler_rates_to_body_rates_quad(d.euler, q4::bc<3>(a));
    if (l == 2) wsp = fminf(fmaxf(wsp, -P.max_yaw_rate), P.max_yaw_rate);
    const float qd = quat_from_euler_quad(q4::by_lane(l, q4::bc<1>(a), q4::bc<2>(a), yaw));
    torque = quad_body_torque<false>(C, q, qd, d.wbody, wsp, kr, kw, l);
  } else {  // AGX_CTRL_RATES: rates_control.py:16-30 (line 25's broadcast bug -> z component)
    fz = (q4::bc<0>(a) - P.gravity[2]) * C.mass;
    float wsp = q4::perm<1, 2, 3, 3>(a);  // (a1, a2, a3)
    if (l == 2) wsp = fminf(fmaxf(wsp, -P.max_yaw_rate), P.max_yaw_rate);
    torque = quad_body_torque<false>(C, q, q, d.wbody, wsp, kr, kw, l);
  }
  force = l == 2 ? fz : 0.0f;
}

template <int M, int CTRL>
__global__ void __launch_bounds__(64, 1)
    k_env_step_quad_loop(AgxRobotParams P, AgxEnvBuffers B, int n, const float *__restrict__ actions_in, int k, AgxTaskArgs T) {
  static_assert((M == 4 && CTRL >= AGX_CTRL_POSITION && CTRL <= AGX_CTRL_VEL_STEERING) ||
                    (M == 8 && (CTRL == AGX_CTRL_FULLY_ACTUATED || CTRL == AGX_CTRL_POSITION || CTRL == AGX_CTRL_VELOCITY)),
                "the six Lee laws of the quadrotor; the octarotor (two motors per lane) under its three laws: fully actuated, Lee "
                "position, Lee velocity (control/__init__.py:94-96)");
  constexpr bool FA = CTRL == AGX_CTRL_FULLY_ACTUATED;  // 7 actions: position set-point (3) + orientation set-point xyzw (4)
  constexpr int A = FA ? 7 : 4;
  constexpr int MH = M / 4;
  extern __shared__ float traj[];  // [k][3][16] sub-step positions of the wave's 16 envs (only with obstacles)
  const int tid = threadIdx.x;
  const int l = tid & 3, l3 = l < 3 ? l : 2, slot = tid >> 2;
  const int i = blockIdx.x * 16 + slot;
  const unsigned ol = ((unsigned)l * (unsigned)n + (unsigned)i) * 4u, ol3 = ((unsigned)l3 * (unsigned)n + (unsigned)i) * 4u;  // AGX_QAT
  bool reset = false;
  if (blockIdx.x == 0) push_publish_previous(B);  // peer push: the previous step's rows have landed everywhere
  const uint32_t push_peek = blockIdx.x == 0 ? push_wait_peek(B) : 0u;  // ... and this step's slot: looked at when the kernel is done
  if (i < n) {
    float p = AGX_QAT(B.state, 0, ol3), q = AGX_QAT(B.state, 3, ol), v = AGX_QAT(B.state, 7, ol3), w = AGX_QAT(B.state, 10, ol3);
    float u[MH], kT[MH], tinc[MH], tdec[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) {  // motors l and l + 4
      u[h] = AGX_QAT(B.motor_thrust, 4 * h, ol);
      kT[h] = P.use_rps ? AGX_QAT(B.motor_kT, 4 * h, ol) : 1.0f;
      tinc[h] = B.motor_tau_inc ? AGX_QAT(B.motor_tau_inc, 4 * h, ol) : P.tau_inc_uniform;
      tdec[h] = B.motor_tau_dec ? AGX_QAT(B.motor_tau_dec, 4 * h, ol) : P.tau_dec_uniform;
    }
    const float a_in = actions_in[(size_t)i * A + l];  // (a0 .. a3); fully actuated: position set-point in lanes 0..2
    const float a_old = AGX_QAT(B.actions, 0, ol);
    const float a_in2 = FA ? actions_in[(size_t)i * A + 3 + l] : 0.0f;  // fully actuated: orientation set-point xyzw
    const float a_old2 = FA ? AGX_QAT(B.actions, 3, ol) : 0.0f;
    const float kp = B.gains ? AGX_QAT(B.gains, 0, ol3) : P.gains_uniform[0 + l3];
    const float kv = B.gains ? AGX_QAT(B.gains, 3, ol3) : P.gains_uniform[3 + l3];
    const float kr = B.gains ? AGX_QAT(B.gains, 6, ol3) : P.gains_uniform[6 + l3];
    const float kw = B.gains ? AGX_QAT(B.gains, 9, ol3) : P.gains_uniform[9 + l3];
    const QuadConsts<M> C = load_quad_consts<M>(P, l, l3);
    // what the epilogue reads, requested with the state (see k_env_step: a load behind the stores is a round trip of its own)
    const int steps_in = B.sim_steps[i];
    const float tgt = T.kind != AGX_TASK_NONE ? AGX_QAT(T.target, 0, ol3) : 0.0f;
    const float ppe = (T.kind != AGX_TASK_NONE && T.kind != AGX_TASK_POSITION) ? AGX_QAT(T.pos_err, 0, ol3) : 0.0f;
    const float a_prev_in = k == 0 ? AGX_QAT(B.prev_actions, 0, ol) : 0.0f;
    const float a_prev_in2 = (FA && k == 0) ? AGX_QAT(B.prev_actions, 3, ol) : 0.0f;
    const float dmax = B.disturb_max[l3], dmax_t = B.disturb_max[3 + l3];
    const float a = clamp_minmax(a_in, -10.0f, 10.0f);  // clip_actions (the same every sub-step)
    const float a2 = clamp_minmax(a_in2, -10.0f, 10.0f);
    // Obstacles: lane l tests boxes l, l + 4, ...  The cull data (centre, bounding radius) of kBoxBatch of them is requested in
    // ONE go -- a box per loop trip was a dependent memory round trip per trip (27 of them on BASELINE configs[2], with one wave
    // per SIMD and nothing to hide them behind) -- and the first batch before the sub-step loop, whose arithmetic covers it.
    constexpr int kBoxBatch = M == 8 ? 12 : 16;  // (4 x 16 registers held over the sub-step loop; the octarotor instances stay <= 256 VGPRs)
    struct BoxCull { float cx[kBoxBatch], cy[kBoxBatch], cz[kBoxBatch], rad[kBoxBatch]; };
    const int nb = (B.boxes && k > 0) ? B.num_boxes : 0;
    auto load_cull = [&](int b0, BoxCull &K) {
#pragma unroll
      for (int u = 0; u < kBoxBatch; ++u) {
        const int b = b0 + 4 * u;
        const float *bx = B.boxes + (size_t)(b < nb ? b : b0) * 11 * n + i;  // past the end: this lane's first box again, not used
        K.cx[u] = bx[0]; K.cy[u] = bx[(size_t)n]; K.cz[u] = bx[2 * (size_t)n]; K.rad[u] = bx[10 * (size_t)n];
      }
    };
    BoxCull cull0{};
    if (l < nb) load_cull(l, cull0);
    QuadDerived d{};
    float force = 0.0f, torque = 0.0f, fb = 0.0f;
    float tlo = p, thi = p;
    for (int sub = 0; sub < k; ++sub) {
      d = update_states_quad(q, v, w);
      quad_controller<CTRL>(P, C, p, q, v, d, a, a2, kp, kv, kr, kw, l, force, torque);
      // ---- allocation + motor model + body wrench
      float tb;
      quad_allocate<M>(P, C, force, torque, u, kT, tinc, tdec, fb, tb);
      if (B.disturb) {  // apply_disturbance (base_multirotor.py:213-234), draws supplied by the host
        const float *dd = B.disturb + (size_t)sub * 7 * n + i;
        const float occ = dd[0];
        fb += ((dmax - (-dmax)) * dd[(size_t)(1 + l3) * n] + (-dmax)) * occ;
        tb += ((dmax_t - (-dmax_t)) * dd[(size_t)(4 + l3) * n] + (-dmax_t)) * occ;
      } else if (B.disturb_prob > 0.0f) {  // same, drawn in place (every lane of the quad draws the env's 7 uniforms)
        float ud[7];
        rng_fill<7>(B.rng_seed, B.env_index_base + i, agx::step_index(B), RNG_DISTURB + sub, ud);
        const float occ = ud[0] < B.disturb_prob ? 1.0f : 0.0f;
        fb += ((dmax - (-dmax)) * q4::by_lane(l3, ud[1], ud[2], ud[3]) + (-dmax)) * occ;
        tb += ((dmax_t - (-dmax_t)) * q4::by_lane(l3, ud[4], ud[5], ud[6]) + (-dmax_t)) * occ;
      }
      quad_integrate(P, C, p, q, v, w, fb, tb, l);
      if (B.boxes) {
        if (l < 3) traj[(sub * 3 + l) * 16 + slot] = p;
        if (sub == 0) { tlo = p; thi = p; }
        tlo = fminf(tlo, p);
        thi = fmaxf(thi, p);
      }
    }
    if (B.body_force && l < 3 && k > 0) AGX_QAT(B.body_force, 0, ol) = fb;
    // ---- obstacles: the env's boxes over the four lanes
    bool crashed = false;
    if (B.boxes && k > 0) {
      const V3 lo = V3{q4::bc<0>(tlo), q4::bc<1>(tlo), q4::bc<2>(tlo)}, hi = V3{q4::bc<0>(thi), q4::bc<1>(thi), q4::bc<2>(thi)};
      const float rad = P.collision_radius, r2 = rad * rad;
      bool hit = false;
      auto test_batch = [&](int b0, const BoxCull &K) {
#pragma unroll
        for (int u = 0; u < kBoxBatch; ++u) {
          const int b = b0 + 4 * u;
          const V3 c = V3{K.cx[u], K.cy[u], K.cz[u]};
          const float reach = K.rad[u] + rad + 1.0e-3f;
          const float dx = fmaxf(fmaxf(lo.x - c.x, c.x - hi.x), 0.0f);
          const float dy = fmaxf(fmaxf(lo.y - c.y, c.y - hi.y), 0.0f);
          const float dz = fmaxf(fmaxf(lo.z - c.z, c.z - hi.z), 0.0f);
          if (b < nb && !(dx * dx + dy * dy + dz * dz > reach * reach)) {  // (rare: the boxes the trajectory's AABB reaches)
            const float *bx = B.boxes + (size_t)b * 11 * n + i;
            const Q4 bq = Q4{bx[3 * (size_t)n], bx[4 * (size_t)n], bx[5 * (size_t)n], bx[6 * (size_t)n]};
            const V3 bh = V3{bx[7 * (size_t)n], bx[8 * (size_t)n], bx[9 * (size_t)n]};
            for (int sub = 0; sub < k; ++sub) {
              const V3 ps = V3{traj[(sub * 3 + 0) * 16 + slot], traj[(sub * 3 + 1) * 16 + slot], traj[(sub * 3 + 2) * 16 + slot]};
              hit = hit || sphere_hits_box(ps, c, bq, bh, r2);
            }
          }
        }
      };
      if (l < nb) test_batch(l, cull0);
      for (int b0 = l + 4 * kBoxBatch; b0 < nb; b0 += 4 * kBoxBatch) {
        BoxCull K;
        load_cull(b0, K);
        test_batch(b0, K);
      }
      crashed = ((vote(hit) >> (tid & 60)) & 0xFull) != 0ull;
    }
    // ---- stores
    if (l < 3) AGX_QAT(B.state, 0, ol) = p;
    AGX_QAT(B.state, 3, ol) = q;
    if (l < 3) {
      AGX_QAT(B.state, 7, ol) = v;
      AGX_QAT(B.state, 10, ol) = w;
    }
    if (k > 0) {
      if (l < 3) {
        AGX_QAT(B.derived, 0, ol) = d.euler;
        AGX_QAT(B.derived, 7, ol) = d.vveh;
        AGX_QAT(B.derived, 10, ol) = d.vbody;
        AGX_QAT(B.derived, 13, ol) = d.wbody;
      }
      AGX_QAT(B.derived, 3, ol) = d.qveh;
#pragma unroll
      for (int h = 0; h < MH; ++h) AGX_QAT(B.motor_thrust, 4 * h, ol) = u[h];
      if (B.wrench_cmd && l < 3) {
        AGX_QAT(B.wrench_cmd, 0, ol) = force;
        AGX_QAT(B.wrench_cmd, 3, ol) = torque;
      }
      // RobotManagerIGE.pre_physics_step runs every sub-step: prev <- cur, cur <- action
      if (!FA || l < 3) {
        AGX_QAT(B.prev_actions, 0, ol) = k >= 2 ? a_in : a_old;
        AGX_QAT(B.actions, 0, ol) = a_in;
      }
      if (FA) {
        AGX_QAT(B.prev_actions, 3, ol) = k >= 2 ? a_in2 : a_old2;
        AGX_QAT(B.actions, 3, ol) = a_in2;
      }
    }
    // ---- EnvManager bookkeeping + task epilogue (scalar code, the same in the four lanes; lane 0 stores)
    const float acur = k > 0 ? a_in : a_old;
    const float aprev = k >= 2 ? a_in : (k == 1 ? a_old : a_prev_in);
    // action component 3 as the navigation reward reads it: a3, or the first orientation component of the 7-D command
    const float acur3 = FA ? q4::bc<0>(k > 0 ? a_in2 : a_old2) : q4::bc<3>(acur);
    const float aprev3 = FA ? q4::bc<0>(k >= 2 ? a_in2 : (k == 1 ? a_old2 : a_prev_in2)) : q4::bc<3>(aprev);
    const int steps = steps_in + 1;
    bool trunc = false;
    float rew = 0.0f;
    if (T.kind != AGX_TASK_NONE) {
      if (T.kind == AGX_TASK_POSITION) {
        EnvState s;
        s.p = V3{q4::bc<0>(p), q4::bc<1>(p), q4::bc<2>(p)};
        s.q = Q4{q4::bc<0>(q), q4::bc<1>(q), q4::bc<2>(q), q4::bc<3>(q)};
        s.v = V3{0, 0, 0};
        s.w = V3{0, 0, 0};
        rew = reward_position(s, Q4{q4::bc<0>(d.qveh), q4::bc<1>(d.qveh), q4::bc<2>(d.qveh), q4::bc<3>(d.qveh)},
                              V3{q4::bc<0>(d.wbody), q4::bc<1>(d.wbody), q4::bc<2>(d.wbody)},
                              V3{q4::bc<0>(tgt), q4::bc<1>(tgt), q4::bc<2>(tgt)}, crashed);
      } else {
        const float pe = q4::quat_rotate_inverse(d.qveh, tgt - p);
        if (l < 3) {
          AGX_QAT(T.prev_pos_err, 0, ol) = ppe;
          AGX_QAT(T.pos_err, 0, ol) = pe;
        }
        rew = reward_navigation(T.rp, T.curriculum_progress, V3{q4::bc<0>(pe), q4::bc<1>(pe), q4::bc<2>(pe)},
                                V3{q4::bc<0>(ppe), q4::bc<1>(ppe), q4::bc<2>(ppe)}, q4::bc<0>(acur), q4::bc<2>(acur), acur3,
                                q4::bc<0>(aprev), q4::bc<2>(aprev), aprev3, crashed);
      }
      trunc = steps > T.episode_len;
      reset = (crashed && T.reset_on_collision) || trunc;
      if (T.successes)  // (wave-uniform; lane 0 of the env's quad stores)
        nav_bookkeeping_epilogue(T, i, l == 0, V3{q4::bc<0>(tgt), q4::bc<1>(tgt), q4::bc<2>(tgt)}, V3{q4::bc<0>(p), q4::bc<1>(p), q4::bc<2>(p)},
                                 crashed, trunc);
    }
    if (l == 0) {
      B.sim_steps[i] = steps;
      if (T.kind != AGX_TASK_NONE) {
        T.reward[i] = rew;
        B.reset_mask[i] = reset ? 1 : 0;
      }
      B.crashes[i] = crashed ? 1 : 0;
      B.truncations[i] = trunc ? 1 : 0;
    }
  }
  if (T.kind != AGX_TASK_NONE && __ballot(reset) != 0ull && (tid & 63) == 0) atomicOr(B.reset_flag + B.flag_parity, 1);
  if (blockIdx.x == 0) push_wait_finish(B, push_peek);
}

__global__ void __launch_bounds__(256) k_update_states(AgxEnvBuffers B, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  EnvState s = load_state(B.state, n, i);
  store_derived(B.derived, n, i, update_states(s));
}

// EnvManager.compute_observations (env_manager.py:358-362) on its own: crashes[i] |= the robot's collision sphere at its CURRENT
// position overlaps an obstacle box -- the predicate of the fused step (sphere_hits_box) without a trajectory.
__global__ void __launch_bounds__(256) k_collide_spheres_boxes(AgxEnvBuffers B, int n, float radius) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const V3 p = V3{AGX_AT(B.state, 0), AGX_AT(B.state, 1), AGX_AT(B.state, 2)};
  const float r2 = radius * radius;
  bool hit = false;
  for (int b = 0; b < B.num_boxes; ++b) {
    const float *bx = B.boxes + (size_t)b * 11 * n + i;
    const V3 c = V3{bx[0], bx[(size_t)n], bx[2 * (size_t)n]};
    const float reach = bx[10 * (size_t)n] + radius + 1.0e-3f;  // the box's bounding radius: conservative cull, the flag is exact
    const float dx = p.x - c.x, dy = p.y - c.y, dz = p.z - c.z;
    if (dx * dx + dy * dy + dz * dz > reach * reach) continue;
    const Q4 q = Q4{bx[3 * (size_t)n], bx[4 * (size_t)n], bx[5 * (size_t)n], bx[6 * (size_t)n]};
    const V3 h = V3{bx[7 * (size_t)n], bx[8 * (size_t)n], bx[9 * (size_t)n]};
    hit = hit || sphere_hits_box(p, c, q, h, r2);
  }
  if (hit) B.crashes[i] = 1;
}

__global__ void __launch_bounds__(256) k_controller_wrench(AgxRobotParams P, AgxEnvBuffers B, int n, const float *__restrict__ action) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  EnvState s = load_state(B.state, n, i);
  Derived d = load_derived(B.derived, n, i);
  Gains g = B.gains ? load_gains(B.gains, n, i) : uniform_gains(P);
  float a[AGX_MAX_ACTIONS];
#pragma unroll
  for (int c = 0; c < AGX_MAX_ACTIONS; ++c)
    a[c] = (c < P.num_actions) ? clamp_minmax(action[(size_t)i * P.num_actions + c], -10.0f, 10.0f) : 0.0f;
  Wrench wc{V3{0, 0, 0}, V3{0, 0, 0}};
  switch (P.controller) {
    case AGX_CTRL_POSITION: wc = run_controller<AGX_CTRL_POSITION>(P, s, d, a, g); break;
    case AGX_CTRL_VELOCITY: wc = run_controller<AGX_CTRL_VELOCITY>(P, s, d, a, g); break;
    case AGX_CTRL_ATTITUDE: wc = run_controller<AGX_CTRL_ATTITUDE>(P, s, d, a, g); break;
    case AGX_CTRL_RATES: wc = run_controller<AGX_CTRL_RATES>(P, s, d, a, g); break;
    case AGX_CTRL_ACCELERATION: wc = run_controller<AGX_CTRL_ACCELERATION>(P, s, d, a, g); break;
    case AGX_CTRL_VEL_STEERING: wc = run_controller<AGX_CTRL_VEL_STEERING>(P, s, d, a, g); break;
    case AGX_CTRL_FULLY_ACTUATED: wc = run_controller<AGX_CTRL_FULLY_ACTUATED>(P, s, d, a, g); break;
    default: break;
  }
  AGX_AT(B.wrench_cmd, 0) = wc.f.x; AGX_AT(B.wrench_cmd, 1) = wc.f.y; AGX_AT(B.wrench_cmd, 2) = wc.f.z;
  AGX_AT(B.wrench_cmd, 3) = wc.t.x; AGX_AT(B.wrench_cmd, 4) = wc.t.y; AGX_AT(B.wrench_cmd, 5) = wc.t.z;
}

// BaseMultirotor.step(action) of the reference as ONE launch (agx_robot_step; the robot plug-in's super().step()):
// update_states, clip, controller, allocation + motor model, the per-body force / torque tensors, drag, disturbance.
// One lane per env, runtime motor count and control law: a plug-in path evaluated between host calls, not a hot loop.
__global__ void __launch_bounds__(256) k_robot_step(AgxRobotParams P, AgxEnvBuffers B, int n, const float *__restrict__ action,
                                                    AgxRobotStepArgs R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int M = P.num_motors, A = P.num_actions, NB = R.num_bodies;
  EnvState s = load_state(B.state, n, i);
  const Derived d = update_states(s);
  store_derived(B.derived, n, i, d);
  float a[AGX_MAX_ACTIONS];
#pragma unroll
  for (int c = 0; c < AGX_MAX_ACTIONS; ++c) a[c] = (c < A) ? clamp_minmax(action[(size_t)i * A + c], -10.0f, 10.0f) : 0.0f;  // clip_actions
  float u[AGX_MAX_MOTORS];
  Wrench wc{V3{0, 0, 0}, V3{0, 0, 0}};
  if (P.controller != AGX_CTRL_NONE) {
    Gains g = B.gains ? load_gains(B.gains, n, i) : uniform_gains(P);
    switch (P.controller) {
      case AGX_CTRL_POSITION: wc = run_controller<AGX_CTRL_POSITION>(P, s, d, a, g); break;
      case AGX_CTRL_VELOCITY: wc = run_controller<AGX_CTRL_VELOCITY>(P, s, d, a, g); break;
      case AGX_CTRL_ATTITUDE: wc = run_controller<AGX_CTRL_ATTITUDE>(P, s, d, a, g); break;
      case AGX_CTRL_RATES: wc = run_controller<AGX_CTRL_RATES>(P, s, d, a, g); break;
      case AGX_CTRL_ACCELERATION: wc = run_controller<AGX_CTRL_ACCELERATION>(P, s, d, a, g); break;
      case AGX_CTRL_VEL_STEERING: wc = run_controller<AGX_CTRL_VEL_STEERING>(P, s, d, a, g); break;
      case AGX_CTRL_FULLY_ACTUATED: wc = run_controller<AGX_CTRL_FULLY_ACTUATED>(P, s, d, a, g); break;
      default: break;
    }
  }
  const float w6[6] = {wc.f.x, wc.f.y, wc.f.z, wc.t.x, wc.t.y, wc.t.z};
#pragma unroll
  for (int j = 0; j < AGX_MAX_MOTORS; ++j) {
    u[j] = 0.0f;
    if (j < M) {
      float ref = a[j];  // no_control: the action IS the motor command
      if (P.controller != AGX_CTRL_NONE) {
        ref = 0.0f;
#pragma unroll
        for (int c = 0; c < 6; ++c) ref += P.alloc_pinv[6 * j + c] * w6[c];
      }
      const float kT = P.use_rps ? AGX_AT(B.motor_kT, j) : 1.0f;
      const float tinc = B.motor_tau_inc ? AGX_AT(B.motor_tau_inc, j) : P.tau_inc_uniform;
      const float tdec = B.motor_tau_dec ? AGX_AT(B.motor_tau_dec, j) : P.tau_dec_uniform;
      u[j] = motor_update(P, ref, AGX_AT(B.motor_thrust, j), kT, tinc, tdec);
      AGX_AT(B.motor_thrust, j) = u[j];
    }
  }
  if (B.wrench_cmd) {
    AGX_AT(B.wrench_cmd, 0) = wc.f.x; AGX_AT(B.wrench_cmd, 1) = wc.f.y; AGX_AT(B.wrench_cmd, 2) = wc.f.z;
    AGX_AT(B.wrench_cmd, 3) = wc.t.x; AGX_AT(B.wrench_cmd, 4) = wc.t.y; AGX_AT(B.wrench_cmd, 5) = wc.t.z;
  }
  // call_controller (base_multirotor.py:246-258): output_forces / output_torques are zero outside the application mask
  float *F = R.force + (size_t)i * NB * 3, *T = R.torque + (size_t)i * NB * 3;
  for (int b = 0; b < NB * 3; ++b) { F[b] = 0.0f; T[b] = 0.0f; }
  if (P.root_link_mode) {  // control_allocation.py:67-79: output wrench = A u at the (single) masked body
    float w[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float acc = 0.0f;
      for (int j = 0; j < M; ++j) acc += P.alloc[M * r + j] * u[j];
      w[r] = acc;
    }
    const int b = R.body_of_motor[0];
    F[3 * b] = w[0]; F[3 * b + 1] = w[1]; F[3 * b + 2] = w[2];
    T[3 * b] = w[3]; T[3 * b + 1] = w[4]; T[3 * b + 2] = w[5];
  } else {  // control_allocation.py:103-114: force (0, 0, u), torque cq * force * (-dir) at every motor link, in the LINK's frame
    for (int j = 0; j < M; ++j) {
      const int b = R.body_of_motor[j];
      F[3 * b + 2] = u[j];
      T[3 * b] = (P.cq * 0.0f) * (-P.motor_dir[j]);
      T[3 * b + 1] = (P.cq * 0.0f) * (-P.motor_dir[j]);
      T[3 * b + 2] = (P.cq * u[j]) * (-P.motor_dir[j]);
    }
  }
  // simulate_drag (:260-285), then apply_disturbance (:213-234): both `+=` into body 0
  {
    const float vbn = norm(d.vbody);
    F[0] += (-P.lin_drag_linear[0] * d.vbody.x) + (-P.lin_drag_quadratic[0] * vbn * d.vbody.x);
    F[1] += (-P.lin_drag_linear[1] * d.vbody.y) + (-P.lin_drag_quadratic[1] * vbn * d.vbody.y);
    F[2] += (-P.lin_drag_linear[2] * d.vbody.z) + (-P.lin_drag_quadratic[2] * vbn * d.vbody.z);
    T[0] += (-P.ang_drag_linear[0] * d.wbody.x) + (-P.ang_drag_quadratic[0] * fabsf(d.wbody.x) * d.wbody.x);
    T[1] += (-P.ang_drag_linear[1] * d.wbody.y) + (-P.ang_drag_quadratic[1] * fabsf(d.wbody.y) * d.wbody.y);
    T[2] += (-P.ang_drag_linear[2] * d.wbody.z) + (-P.ang_drag_quadratic[2] * fabsf(d.wbody.z) * d.wbody.z);
  }
  float di[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  bool any = false;
  if (B.disturb) {  // draws supplied by the host ([k][7][N] rows of this sub-step)
    const float *dd = B.disturb + (size_t)R.substep * 7 * n + i;
    const float occ = dd[0];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float lo = -B.disturb_max[c], hi = B.disturb_max[c];
      di[c] = ((hi - lo) * dd[(size_t)(1 + c) * n] + lo) * occ;
    }
    any = true;
  } else if (B.disturb_prob > 0.0f) {  // the device stream of the fused step: same (env, step, sub-step) -> same draws
    float ud[7];
    rng_fill<7>(B.rng_seed, B.env_index_base + i, agx::step_index(B), RNG_DISTURB + R.substep, ud);
    const float occ = ud[0] < B.disturb_prob ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float lo = -B.disturb_max[c], hi = B.disturb_max[c];
      di[c] = ((hi - lo) * ud[1 + c] + lo) * occ;
    }
    any = true;
  }
  if (any) {
    F[0] += di[0]; F[1] += di[1]; F[2] += di[2];
    T[0] += di[3]; T[1] += di[4]; T[2] += di[5];
  }
}

// robot_force_tensor / robot_torque_tensor -> the net body-frame wrench on the rigid composite (agx_net_body_wrench)
__global__ void __launch_bounds__(256) k_net_body_wrench(int n, AgxLinkFrames L, const float *__restrict__ force,
                                                         const float *__restrict__ torque, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int NB = L.num_bodies;
  const float *F = force + (size_t)i * NB * 3, *T = torque + (size_t)i * NB * 3;
  float w[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int b = 0; b < NB; ++b) {
    const float *Rm = L.rot[b], *r = L.pos[b];
    const float f[3] = {F[3 * b], F[3 * b + 1], F[3 * b + 2]}, t[3] = {T[3 * b], T[3 * b + 1], T[3 * b + 2]};
    float fr[3], tr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      fr[c] = (Rm[3 * c] * f[0] + Rm[3 * c + 1] * f[1]) + Rm[3 * c + 2] * f[2];
      tr[c] = (Rm[3 * c] * t[0] + Rm[3 * c + 1] * t[1]) + Rm[3 * c + 2] * t[2];
    }
    const float cx = r[1] * fr[2] - r[2] * fr[1], cy = r[2] * fr[0] - r[0] * fr[2], cz = r[0] * fr[1] - r[1] * fr[0];
    w[0] += fr[0]; w[1] += fr[1]; w[2] += fr[2];
    w[3] += cx + tr[0]; w[4] += cy + tr[1]; w[5] += cz + tr[2];
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) out[(size_t)i * 6 + c] = w[c];
}

// ---------------------------------------------------------------------------------------
// Stand-alone task kernels (same device functions as the fused epilogue)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_reward_position(AgxEnvBuffers B, int n, const float *__restrict__ target, int episode_len,
                                                          int reset_on_collision, float *__restrict__ reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool reset = false;
  if (i < n) {
    EnvState s = load_state(B.state, n, i);
    Q4 qveh = Q4{AGX_AT(B.derived, 3), AGX_AT(B.derived, 4), AGX_AT(B.derived, 5), AGX_AT(B.derived, 6)};
    V3 wb = V3{AGX_AT(B.derived, 13), AGX_AT(B.derived, 14), AGX_AT(B.derived, 15)};
    V3 tgt = V3{AGX_AT(target, 0), AGX_AT(target, 1), AGX_AT(target, 2)};
    bool crash = B.crashes[i] != 0;
    reward[i] = reward_position(s, qveh, wb, tgt, crash);
    B.crashes[i] = crash ? 1 : 0;
    bool trunc = B.sim_steps[i] > episode_len;
    B.truncations[i] = trunc ? 1 : 0;
    reset = (crash && reset_on_collision) || trunc;
    B.reset_mask[i] = reset ? 1 : 0;
  }
  if (__ballot(reset) != 0ull && (threadIdx.x & 63) == 0) atomicOr(B.reset_flag + B.flag_parity, 1);
}

// position_setpoint_task.py:194-203, obs [N][13] row-major (what the policy network consumes)
// reward | terminated | truncated behind the observation in the exchange row (header: step_rows)
AGX_DEV void write_step_row_tail(const AgxEnvBuffers &B, int i, float *__restrict__ row, int obs_dim) {
  row_store(B, row + obs_dim, B.step_reward[i]);
  row_store(B, row + obs_dim + 1, B.crashes[i] ? 1.0f : 0.0f);
  row_store(B, row + obs_dim + 2, B.truncations[i] ? 1.0f : 0.0f);
}
AGX_DEV void write_obs_position(const AgxEnvBuffers &B, int n, int i, V3 tgt, float *__restrict__ obs, const EnvState &s,
                                const Derived &d) {
  float v[13] = {tgt.x - s.p.x, tgt.y - s.p.y, tgt.z - s.p.z, s.q.x, s.q.y, s.q.z, s.q.w,
                 d.vbody.x, d.vbody.y, d.vbody.z, d.wbody.x, d.wbody.y, d.wbody.z};
  float *o = obs + (size_t)i * 13;
#pragma unroll
  for (int c = 0; c < 13; ++c) o[c] = v[c];
  if (float *rows = B.step_rows[B.flag_parity]) {
    float *r = rows + (size_t)i * 16;
    if (B.push_world > 0) {  // the 64-byte row as four 16-byte stores per destination
      row_store4_push(B, r, v[0], v[1], v[2], v[3]);
      row_store4_push(B, r + 4, v[4], v[5], v[6], v[7]);
      row_store4_push(B, r + 8, v[8], v[9], v[10], v[11]);
      row_store4_push(B, r + 12, v[12], B.step_reward[i], B.crashes[i] ? 1.0f : 0.0f, B.truncations[i] ? 1.0f : 0.0f);
    } else {
#pragma unroll
      for (int c = 0; c < 13; ++c) row_store(B, r + c, v[c]);
      write_step_row_tail(B, i, r, 13);
    }
  }
}
__global__ void __launch_bounds__(256) k_obs_position(AgxEnvBuffers B, int n, const float *__restrict__ target, float *__restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  push_wait_for_slot(B);
  if (i < n)
    write_obs_position(B, n, i, V3{AGX_AT(target, 0), AGX_AT(target, 1), AGX_AT(target, 2)}, obs, load_state(B.state, n, i),
                       load_derived(B.derived, n, i));
  step_rows_signal(B);
}

struct NavParams {
  float rp[18];
};

// navigation_task.py:416-521 (+ :305-309 truncation)
__global__ void __launch_bounds__(256) k_reward_navigation(AgxEnvBuffers B, int n, const float *__restrict__ target, NavParams R,
                                                            float cpf, float *__restrict__ pos_err,
                                                            float *__restrict__ prev_pos_err, int episode_len,
                                                            int reset_on_collision, float *__restrict__ reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool reset = false;
  if (i < n) {
    V3 p = V3{AGX_AT(B.state, 0), AGX_AT(B.state, 1), AGX_AT(B.state, 2)};
    Q4 qveh = Q4{AGX_AT(B.derived, 3), AGX_AT(B.derived, 4), AGX_AT(B.derived, 5), AGX_AT(B.derived, 6)};
    V3 tgt = V3{AGX_AT(target, 0), AGX_AT(target, 1), AGX_AT(target, 2)};
    V3 ppe = V3{AGX_AT(pos_err, 0), AGX_AT(pos_err, 1), AGX_AT(pos_err, 2)};
    AGX_AT(prev_pos_err, 0) = ppe.x; AGX_AT(prev_pos_err, 1) = ppe.y; AGX_AT(prev_pos_err, 2) = ppe.z;
    V3 pe = quat_rotate_inverse(qveh, tgt - p);
    AGX_AT(pos_err, 0) = pe.x; AGX_AT(pos_err, 1) = pe.y; AGX_AT(pos_err, 2) = pe.z;
    bool crash = B.crashes[i] != 0;
    reward[i] = reward_navigation(R.rp, cpf, pe, ppe, AGX_AT(B.actions, 0), AGX_AT(B.actions, 2), AGX_AT(B.actions, 3),
                                  AGX_AT(B.prev_actions, 0), AGX_AT(B.prev_actions, 2), AGX_AT(B.prev_actions, 3), crash);
    bool trunc = B.sim_steps[i] > episode_len;
    B.truncations[i] = trunc ? 1 : 0;
    reset = (crash && reset_on_collision) || trunc;
    B.reset_mask[i] = reset ? 1 : 0;
  }
  if (__ballot(reset) != 0ull && (threadIdx.x & 63) == 0) atomicOr(B.reset_flag + B.flag_parity, 1);
}

// navigation_task.py:369-393; one wave per env so the depth min-pool is a coalesced sweep
// (part, nparts): the env's work split over `nparts` waves -- the state part goes to the last one, the cell rows cy = part,
// part + nparts, ... of the min-pool to each; the minimum over the image comes back as this wave's share (the caller reduces).
AGX_DEV float obs_navigation_env(const AgxEnvBuffers &B, int n, int i, const float *__restrict__ target,
                                 const float *__restrict__ u_vec, const float *__restrict__ u_euler,
                                 const float *__restrict__ pixels, int ns, int H, int W, int gh, int gw, int obs_dim,
                                 float *__restrict__ obs, float *__restrict__ min_pixel, int part = 0, int nparts = 1) {
  const int lane = threadIdx.x & 63;
  float *o = obs + (size_t)i * obs_dim;
  float *row = B.step_rows[B.flag_parity] ? B.step_rows[B.flag_parity] + (size_t)i * (obs_dim + 3) : nullptr;
  float imin = INFINITY;  // NavigationTask.post_image_reward_addition on the same sweep (min_pixel != NULL, ns == 1)
  if (lane == 0 && part == nparts - 1) {
    V3 p = V3{AGX_AT(B.state, 0), AGX_AT(B.state, 1), AGX_AT(B.state, 2)};
    Q4 qveh = Q4{AGX_AT(B.derived, 3), AGX_AT(B.derived, 4), AGX_AT(B.derived, 5), AGX_AT(B.derived, 6)};
    V3 tgt = V3{AGX_AT(target, 0), AGX_AT(target, 1), AGX_AT(target, 2)};
    V3 v = quat_rotate_inverse(qveh, tgt - p);
    float u6[6];
    if (u_vec) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { u6[c] = u_vec[(size_t)i * 3 + c]; u6[3 + c] = u_euler[(size_t)i * 3 + c]; }
    } else {
      rng_fill<6>(B.rng_seed, B.env_index_base + i, agx::step_index(B), RNG_OBS_NOISE, u6);
    }
    // 0.1 * 2 * rand_like(vec - 0.5): the -0.5 sits inside rand_like in the reference (:374)
    V3 pv = V3{v.x + 0.1f * 2.0f * u6[0], v.y + 0.1f * 2.0f * u6[1], v.z + 0.1f * 2.0f * u6[2]};
    float dist = norm(v);
    o[0] = pv.x / dist; o[1] = pv.y / dist; o[2] = pv.z / dist; o[3] = dist;
    float e0 = ssa(AGX_AT(B.derived, 0)), e1 = ssa(AGX_AT(B.derived, 1));
    o[4] = e0 + 0.1f * (u6[3] - 0.5f);
    o[5] = e1 + 0.1f * (u6[4] - 0.5f);
    o[6] = 0.0f;
    o[7] = AGX_AT(B.derived, 10); o[8] = AGX_AT(B.derived, 11); o[9] = AGX_AT(B.derived, 12);
    o[10] = AGX_AT(B.derived, 13); o[11] = AGX_AT(B.derived, 14); o[12] = AGX_AT(B.derived, 15);
    o[13] = AGX_AT(B.actions, 0); o[14] = AGX_AT(B.actions, 1); o[15] = AGX_AT(B.actions, 2); o[16] = AGX_AT(B.actions, 3);
    if (row) {
      for (int c = 0; c < 17 && c < obs_dim; ++c) row_store(B, row + c, o[c]);  // this lane's own stores
      write_step_row_tail(B, i, row, obs_dim);
    }
  }
  if (pixels) {
    // gh x gw min-pool of sensor 0's image as a COALESCED sweep: the wave reads 64 consecutive pixels of a row per load
    // (each lane keeps the minimum of its column over the rows of the cell row), then the columns of one cell are
    // reduced across lanes.  min is exact and order-free, so any arrangement gives the bits of the serial loop.
    const float *img = pixels + (size_t)i * ns * H * W;  // sensor 0
    const int ch = (H + gh - 1) / gh;
    // wide images (W >= 256, cells a multiple of 4 pixels wide: the 32 x 512 LiDAR): a lane takes 4 consecutive pixels per
    // load (1 KB per wave instruction instead of 256 B) and the sweep below runs over these groups of 4
    const bool vec4 = (W & 3) == 0 && W >= 256 && (((W + gw - 1) / gw) & 3) == 0 && ((size_t)img & 15) == 0;
    const int Wv = vec4 ? W >> 2 : W;                    // columns the sweep sees
    const int cw = ((W + gw - 1) / gw) >> (vec4 ? 2 : 0);  // cell width in such columns
    const bool pow2 = (cw & (cw - 1)) == 0 && cw < 64;
    for (int cy = part; cy < gh; cy += nparts) {
      const int y0 = cy * ch, y1 = min(y0 + ch, H);
      float cell = INFINITY;  // lane c < gw: cell (cy, c)
      for (int x0 = 0; x0 < Wv && y0 < y1; x0 += 64) {
        const int x = x0 + lane;
        float m = INFINITY;
        if (x < Wv) {
          // one wave per env: the rows of a cell are requested TOGETHER (batches of 8 loads in flight) -- issued one by one, the
          // 48 row loads of a 64 x 48 frame were 48 memory latencies in sequence and the whole kernel (min is exact: any order)
          for (int yb = y0; yb < y1; yb += 8) {
            if (vec4) {
              float4 v4[8];
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const int y = min(yb + r, y1 - 1);  // (a row read twice changes no minimum)
                v4[r] = *reinterpret_cast<const float4 *>(img + (size_t)y * W + 4 * x);
              }
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const float vv[4] = {v4[r].x, v4[r].y, v4[r].z, v4[r].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  m = fminf(m, vv[k]);
                  float v10 = 10.0f * vv[k];
                  if (v10 < 0.0f) v10 = 10.0f;
                  imin = fminf(imin, v10);
                }
              }
            } else {
              float vr[8];
#pragma unroll
              for (int r = 0; r < 8; ++r) vr[r] = img[(size_t)min(yb + r, y1 - 1) * W + x];
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                m = fminf(m, vr[r]);
                float v10 = 10.0f * vr[r];
                if (v10 < 0.0f) v10 = 10.0f;
                imin = fminf(imin, v10);
              }
            }
          }
        }
        if (pow2) {  // cells are aligned groups of cw lanes: butterfly inside the group, lane c fetches its group's value
          for (int sft = 1; sft < cw; sft <<= 1) m = fminf(m, __shfl_xor(m, sft));
          const int src = lane * cw - x0;
          const float t = __shfl(m, src & 63);
          if (src >= 0 && src < 64 && lane < gw) cell = fminf(cell, t);
        } else {
          const int c_lo = x0 / cw, c_hi = min(x0 + 63, Wv - 1) / cw;
          for (int c = c_lo; c <= c_hi; ++c) {  // wave-uniform: the cells this 64-column chunk touches
            float t = (x < Wv && x / cw == c) ? m : INFINITY;
            for (int off = 32; off > 0; off >>= 1) t = fminf(t, __shfl_xor(t, off));
            if (lane == c) cell = fminf(cell, t);
          }
        }
      }
      const int k = 17 + cy * gw + lane;
      if (lane < gw && k < obs_dim) {
        o[k] = cell;
        if (row) row_store(B, row + k, cell);
      }
    }
    if (min_pixel) {
      for (int off = 32; off > 0; off >>= 1) imin = fminf(imin, __shfl_xor(imin, off));
      if (lane == 0 && nparts == 1) min_pixel[i] = imin;
    }
  }
  return imin;
}
// Small batches (the 256 .. 2048 envs an RL run uses): one WORKGROUP per env, its four waves take every fourth cell row of the
// min-pool each and the last one the state part as well -- the one-wave-per-env form runs the eight cell rows' loads as eight
// memory round trips in sequence and the Philox draws of the state part in front of them (13 us at 256 envs; this one: 5).
// min is exact and order-free: the same bits.
__global__ void __launch_bounds__(256) k_obs_navigation_split(AgxEnvBuffers B, int n, const float *__restrict__ target,
                                                               const float *__restrict__ u_vec, const float *__restrict__ u_euler,
                                                               const float *__restrict__ pixels, int ns, int H, int W, int gh, int gw,
                                                               int obs_dim, float *__restrict__ obs, float *__restrict__ min_pixel) {
  __shared__ float wave_min[4];
  const int i = blockIdx.x, w = threadIdx.x >> 6;
  push_wait_for_slot(B);
  const float imin = obs_navigation_env(B, n, i, target, u_vec, u_euler, pixels, ns, H, W, gh, gw, obs_dim, obs, min_pixel, w, 4);
  if (min_pixel && pixels) {
    if ((threadIdx.x & 63) == 0) wave_min[w] = imin;
    __syncthreads();
    if (threadIdx.x == 0) min_pixel[i] = fminf(fminf(wave_min[0], wave_min[1]), fminf(wave_min[2], wave_min[3]));
  }
  step_rows_signal(B);
}
__global__ void __launch_bounds__(256) k_obs_navigation(AgxEnvBuffers B, int n, const float *__restrict__ target,
                                                         const float *__restrict__ u_vec, const float *__restrict__ u_euler,
                                                         const float *__restrict__ pixels, int ns, int H, int W, int gh, int gw,
                                                         int obs_dim, float *__restrict__ obs, float *__restrict__ min_pixel) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per env
  push_wait_for_slot(B);
  if (i < n) obs_navigation_env(B, n, i, target, u_vec, u_euler, pixels, ns, H, W, gh, gw, obs_dim, obs, min_pixel);
  step_rows_signal(B);
}

// ---------------------------------------------------------------------------------------
// Reset.  Uniform draws come either from tensors (host RNG, reference-faithful stream) or
// from Philox4x32-10 evaluated in place (sync-free mode).
// ---------------------------------------------------------------------------------------
// The uniform draws one env's reset consumes: env bounds (6), robot state (13), controller gains (12), and per motor
// (tau_inc, tau_dec, thrust, kT).
template <int M>
struct ResetDraws {
  float ub[6], us[13], ug[12], um[M][4];
};

// strict mode: the tensors torch drew (AoS, the reference's order)
template <int M>
AGX_DEV void host_reset_draws(const AgxRobotParams &P, const AgxResetArgs &R, int i, ResetDraws<M> &D) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    D.ub[c] = R.u_bounds_lo[(size_t)i * 3 + c];
    D.ub[3 + c] = R.u_bounds_hi[(size_t)i * 3 + c];
  }
#pragma unroll
  for (int c = 0; c < 13; ++c) D.us[c] = R.u_state[(size_t)i * 13 + c];
#pragma unroll
  for (int c = 0; c < 12; ++c) D.ug[c] = R.randomize_gains ? R.u_gains[(size_t)i * 12 + c] : 0.0f;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    size_t k = (size_t)i * M + j;
    D.um[j][0] = R.u_tau_inc[k];
    D.um[j][1] = R.u_tau_dec[k];
    D.um[j][2] = R.u_thrust[k];
    D.um[j][3] = P.use_rps ? R.u_kT[k] : 0.0f;
  }
}

// sync-free mode: the Philox blocks of a resetting env are evaluated by the WAVE, one block per lane (2 bounds + 4 state
// + 3 gains + M motor blocks of 4 draws), and handed to the env's own lane with v_readlane: a lane on its own would run
// the 9 + M blocks (10 rounds each) back to back, and with a few of 8192 envs resetting on almost every step that
// serial chain was the longest path of the reset / observation kernel.  Same (seed; env, episode, stream, block)
// coordinates, hence the same draws as rng_fill / rng_block in any other arrangement.  Must be called by all 64 lanes.
template <int M>
AGX_DEV void wave_reset_draws(const AgxResetArgs &R, int i, int ep, bool mine, ResetDraws<M> &D) {
  constexpr int NB = 9 + M;
  const int lane = threadIdx.x & 63;
  int stream = RNG_MOTOR, blk = lane - 9;
  if (lane < 2) { stream = RNG_BOUNDS; blk = lane; }
  else if (lane < 6) { stream = RNG_STATE; blk = lane - 2; }
  else if (lane < 9) { stream = RNG_GAINS; blk = lane - 6; }
  unsigned long long todo = __ballot(mine);
  if (__popcll(todo) > 8) {  // a full reset (task.reset(), short episodes): every lane for itself is the shorter path
    if (mine) {
      rng_fill<6>(R.seed, i, ep, RNG_BOUNDS, D.ub);
      rng_fill<13>(R.seed, i, ep, RNG_STATE, D.us);
      if (R.randomize_gains) rng_fill<12>(R.seed, i, ep, RNG_GAINS, D.ug);
#pragma unroll
      for (int j = 0; j < M; ++j) {
        F4 um = rng_block(R.seed, i, ep, RNG_MOTOR, j);
#pragma unroll
        for (int k = 0; k < 4; ++k) D.um[j][k] = um.v[k];
      }
    }
    return;
  }
  while (todo) {
    const int L = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int iL = __builtin_amdgcn_readlane(i, L), epL = __builtin_amdgcn_readlane(ep, L);
    F4 f{};
    if (lane < NB) f = rng_block(R.seed, iL, epL, stream, blk);
    const bool me = lane == L;
#define AGX_TAKE(dst, b, k)                                                                   \
  {                                                                                            \
    float v_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.v[k]), (b)));          \
    dst = me ? v_ : dst;                                                                       \
  }
#pragma unroll
    for (int c = 0; c < 6; ++c) AGX_TAKE(D.ub[c], c / 4, c % 4)
#pragma unroll
    for (int c = 0; c < 13; ++c) AGX_TAKE(D.us[c], 2 + c / 4, c % 4)
    if (R.randomize_gains) {
#pragma unroll
      for (int c = 0; c < 12; ++c) AGX_TAKE(D.ug[c], 6 + c / 4, c % 4)
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) AGX_TAKE(D.um[j][k], 9 + j, k)
    }
#undef AGX_TAKE
  }
}

// BaseMultirotor.reset_idx / MotorModel.reset_idx / IsaacGymEnv.reset_idx of ONE env from its draws; returns the new state
template <int M>
AGX_DEV EnvState reset_env(const AgxRobotParams &P, const AgxEnvBuffers &B, int n, const AgxResetArgs &R, int i, int ep,
                           const ResetDraws<M> &D) {
  EnvState s;
  // IsaacGymEnv.reset_idx: env bounds first, the robot spawn uses them
  float bmin[3], bmax[3];
  bounds_from_draws(R, D.ub, bmin, bmax);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    AGX_AT(B.bounds_min, c) = bmin[c];
    AGX_AT(B.bounds_max, c) = bmax[c];
  }
  float r[13];
#pragma unroll
  for (int c = 0; c < 13; ++c) r[c] = (R.max_state[c] - R.min_state[c]) * D.us[c] + R.min_state[c];
  s.p = V3{bmin[0] + (bmax[0] - bmin[0]) * r[0], bmin[1] + (bmax[1] - bmin[1]) * r[1], bmin[2] + (bmax[2] - bmin[2]) * r[2]};
  s.q = quat_from_euler(r[3], r[4], r[5]);
  s.v = V3{r[7], r[8], r[9]};
  s.w = V3{r[10], r[11], r[12]};
  store_state(B.state, n, i, s);
  if (R.randomize_gains) {
#pragma unroll
    for (int c = 0; c < 12; ++c) AGX_AT(B.gains, c) = (R.gains_max[c] - R.gains_min[c]) * D.ug[c] + R.gains_min[c];
  }
#pragma unroll
  for (int j = 0; j < M; ++j) {
    if (B.motor_tau_inc) AGX_AT(B.motor_tau_inc, j) = (R.tau_inc_max - R.tau_inc_min) * D.um[j][0] + R.tau_inc_min;
    if (B.motor_tau_dec) AGX_AT(B.motor_tau_dec, j) = (R.tau_dec_max - R.tau_dec_min) * D.um[j][1] + R.tau_dec_min;
    AGX_AT(B.motor_thrust, j) = (P.max_thrust - P.min_thrust) * D.um[j][2] + P.min_thrust;
    if (P.use_rps) AGX_AT(B.motor_kT, j) = (R.kT_max - R.kT_min) * D.um[j][3] + R.kT_min;
  }
  B.sim_steps[i] = 0;
  if (B.episode_count) B.episode_count[i] = ep + 1;
  return s;
}

// What follows the reset decision of one env step, for one env: the masked reset (base_multirotor.py:177-205,
// motor_model.py:140-154, env_manager.py:301) and, when WITH_OBS, the position task's observation of the post-reset state.
// `any`: some env of the batch resets (wave-uniform); s / d: the env's state and derived tensors as the step left them.
// Must be called by all 64 lanes (wave_reset_draws).
template <int M, bool WITH_OBS>
AGX_DEV void reset_and_observe(const AgxRobotParams &P, const AgxEnvBuffers &B, int n, const AgxResetArgs &R, int i, bool valid,
                               bool any, bool mine, int ep, V3 tgt, float *__restrict__ obs, const EnvState &s, const Derived &d) {
  if (!any) {  // nobody resets: the reference does not touch anything
    if (WITH_OBS && valid) write_obs_position(B, n, i, tgt, obs, s, d);
    return;
  }
  ResetDraws<M> D{};
  if (R.u_state) {
    if (mine) host_reset_draws<M>(P, R, i, D);
  } else {
    wave_reset_draws<M>(R, B.env_index_base + i, ep, mine, D);  // draws are keyed by the GLOBAL env index
  }
  if (valid) {
    EnvState s2 = mine ? reset_env<M>(P, B, n, R, i, ep, D) : s;
    // BaseMultirotor.reset_idx ends with an un-indexed update_states(): every env is refreshed
    // (lean: nobody reads the derived tensors before the next env step rewrites them; the observation reads the body velocities)
    const bool lean = (B.launch_flags & 4) != 0;
    Derived d2 = lean ? update_states_body(s2) : update_states(s2);
    if (!lean) store_derived(B.derived, n, i, d2);
    if (WITH_OBS) write_obs_position(B, n, i, tgt, obs, s2, d2);
  }
}

// The reset / observation half of the env step as its own launch.  Every load is issued before the flag is looked at (one
// memory round trip instead of flag -> mask -> state in sequence).
template <int M, bool WITH_OBS>
__global__ void __launch_bounds__(256) k_reset_masked(AgxRobotParams P, AgxEnvBuffers B, int n, AgxResetArgs R,
                                                      const float *__restrict__ target, float *__restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) B.reset_flag[B.flag_parity ^ 1] = 0;  // the NEXT step's flag; nobody reads or writes it now
  if (WITH_OBS) push_wait_for_slot(B);
  const bool valid = i < n;
  EnvState s{};
  Derived d{};
  V3 tgt{};
  int mask = 0, ep = 0;
  if (valid) {
    s = load_state(B.state, n, i);
    if (WITH_OBS) {
      d = load_derived(B.derived, n, i);
      tgt = V3{AGX_AT(target, 0), AGX_AT(target, 1), AGX_AT(target, 2)};
    }
    mask = B.reset_mask[i];  // compared below, behind the last load (see reset_masked_quad_obs_body)
    if (B.episode_count) ep = B.episode_count[i];
  }
  const int flag = B.reset_flag[B.flag_parity];  // (one word: the branch is taken by whole waves)
  const bool any = flag != 0, mine = mask != 0;
  reset_and_observe<M, WITH_OBS>(P, B, n, R, i, valid, any, mine && any, ep, tgt, obs, s, d);
  if (WITH_OBS) step_rows_signal(B);
}

// The robot side of a navigation step in one launch (agx_nav_robot_side): the masked robot reset, the sensor mounts and the
// target of the envs that reset, the world pose of every sensor -- four dependent launches of ~5 us each at RL batch sizes.
// Same device functions as the stand-alone kernels, same order; what the later parts read (episode count, bounds, state) was
// written by the SAME thread, so program order is all the ordering it takes.
template <int M>
__global__ void __launch_bounds__(256) k_nav_robot_side(AgxRobotParams P, AgxEnvBuffers B, int n, AgxResetArgs R, AgxNavRobotSideArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) B.reset_flag[B.flag_parity ^ 1] = 0;  // the NEXT step's flag; nobody reads or writes it now
  const bool valid = i < n;
  EnvState s{};
  int mask = 0, ep = 0;
  if (valid) {
    s = load_state(B.state, n, i);
    mask = B.reset_mask[i];  // compared below, behind the last load (see reset_masked_quad_obs_body)
    if (B.episode_count) ep = B.episode_count[i];
  }
  const int flag = B.reset_flag[B.flag_parity];  // (one word: the branch is taken by whole waves)
  const bool any = flag != 0, mine = mask != 0;
  reset_and_observe<M, false>(P, B, n, R, i, valid, any, mine && any, ep, V3{}, nullptr, s, Derived{});
  if (!valid) return;
  const int ns = A.num_sensors;
  if (any && mine) {
    if (A.randomize_mount) {
      Ratio3 Tr, Ro;
#pragma unroll
      for (int c = 0; c < 3; ++c) { Tr.lo[c] = A.mount_t_min[c]; Tr.hi[c] = A.mount_t_max[c]; Ro.lo[c] = A.mount_r_min[c]; Ro.hi[c] = A.mount_r_max[c]; }
      for (int q = 0; q < ns; ++q) sensor_mount_reset_env(B, i, q, i * ns + q, Tr, Ro, nullptr, nullptr, A.local_pos, A.local_quat);
    }
    if (A.reset_target) {
      Ratio3 Rt;
#pragma unroll
      for (int c = 0; c < 3; ++c) { Rt.lo[c] = A.target_ratio_min[c]; Rt.hi[c] = A.target_ratio_max[c]; }
      nav_target_reset_env(B, n, i, A.num_actions, Rt, nullptr, A.target, A.target_yaw, A.zero_prev_actions);
    }
  }
  const Q4 fq = Q4{A.frame_quat[0], A.frame_quat[1], A.frame_quat[2], A.frame_quat[3]};
  for (int q = 0; q < ns; ++q) sensor_pose_env(B, n, i, i * ns + q, A.local_pos, A.local_quat, fq, A.sensor_pos, A.sensor_quat);
}

// k_reset_masked<4, WITH_OBS> with four lanes per env (see k_env_step_quad_position): the refresh of every env's derived
// tensors and the observation are vector work; the reset of an env itself (rare: a few of 8192 per step) stays the scalar
// code, run by the first lane of the env's quad, which then hands the new state to the other three.
// HOST_DRAWS: the strict mode's uniforms come from tensors the host filled (R.u_state ...).  Its own instance, so that the
// kernel of the device-RNG mode holds none of those loads: at the join of the two paths the compiler otherwise waits
// (s_waitcnt vmcnt(N)) for loads that only the other path issued, and on gfx9 that counter also counts the STORES of a
// resetting env -- the slowest waves of the launch sat out their own stores' round trips twice.
template <bool HOST_DRAWS>
AGX_DEV void reset_masked_quad_obs_body(const AgxRobotParams &P, const AgxEnvBuffers &B, int n, const AgxResetArgs &R,
                                        const float *__restrict__ target, float *__restrict__ obs) {
  const int tid = threadIdx.x;
  const int l = tid & 3, l3 = l < 3 ? l : 2;
  const int i = blockIdx.x * 16 + (tid >> 2);
  const unsigned ol = ((unsigned)l * (unsigned)n + (unsigned)i) * 4u, ol3 = ((unsigned)l3 * (unsigned)n + (unsigned)i) * 4u;  // AGX_QAT
  if (blockIdx.x == 0 && tid == 0) B.reset_flag[B.flag_parity ^ 1] = 0;  // the NEXT step's flag; nobody reads or writes it now
  push_wait_for_slot(B);
  const bool valid = i < n;
  float p = 0.0f, q = 0.0f, v = 0.0f, w = 0.0f, vbody = 0.0f, wbody = 0.0f, tgt = 0.0f;
  int mask = 0, ep = 0, tail_crashed = 0, tail_truncated = 0;
  float tail_reward = 0.0f;
  float *const rows = B.step_rows[B.flag_parity];
  // every load is issued before ANY of them is looked at: one memory round trip.  (The mask is compared below, not here: a
  // compare inside this block made the compiler wait for the mask byte before it issued the episode count and the flag.)
  if (valid) {
    p = AGX_QAT(B.state, 0, ol3); q = AGX_QAT(B.state, 3, ol); v = AGX_QAT(B.state, 7, ol3); w = AGX_QAT(B.state, 10, ol3);
    vbody = AGX_QAT(B.derived, 10, ol3); wbody = AGX_QAT(B.derived, 13, ol3);
    tgt = AGX_QAT(target, 0, ol3);
    mask = B.reset_mask[i];
    if (B.episode_count) ep = B.episode_count[i];
    if (rows) {  // sharded run: reward | terminated | truncated ride behind the observation in the exchange row
      tail_reward = B.step_reward[i];
      tail_crashed = B.crashes[i];
      tail_truncated = B.truncations[i];
    }
  }
  const int flag = B.reset_flag[B.flag_parity];  // (one word: the branch is taken by whole waves)
  const bool any = flag != 0;
  const bool mine = mask != 0;
  if (any) {
    const bool lead = mine && l == 0;
    ResetDraws<4> D{};
    if (HOST_DRAWS) {
      if (lead) host_reset_draws<4>(P, R, i, D);
    } else {
      wave_reset_draws<4>(R, B.env_index_base + i, ep, lead, D);  // draws are keyed by the GLOBAL env index
    }
    if (__ballot(mine) != 0ull) {  // some env of this wave resets
      EnvState s{};
      if (lead) s = reset_env<4>(P, B, n, R, i, ep, D);
      // the quad takes the new state over from its first lane
      const float npv = q4::by_lane(l3, q4::bc<0>(s.p.x), q4::bc<0>(s.p.y), q4::bc<0>(s.p.z));
      const float nq = q4::by_lane(l, q4::bc<0>(s.q.x), q4::bc<0>(s.q.y), q4::bc<0>(s.q.z), q4::bc<0>(s.q.w));
      const float nv = q4::by_lane(l3, q4::bc<0>(s.v.x), q4::bc<0>(s.v.y), q4::bc<0>(s.v.z));
      const float nw = q4::by_lane(l3, q4::bc<0>(s.w.x), q4::bc<0>(s.w.y), q4::bc<0>(s.w.z));
      p = mine ? npv : p; q = mine ? nq : q; v = mine ? nv : v; w = mine ? nw : w;
    }
    // BaseMultirotor.reset_idx ends with an un-indexed update_states(): every env is refreshed
    const QuadDerived d = update_states_quad(q, v, w);
    if (valid) {
      if (l < 3) {
        AGX_QAT(B.derived, 0, ol) = d.euler;
        AGX_QAT(B.derived, 7, ol) = d.vveh;
        AGX_QAT(B.derived, 10, ol) = d.vbody;
        AGX_QAT(B.derived, 13, ol) = d.wbody;
      }
      AGX_QAT(B.derived, 3, ol) = d.qveh;
    }
    vbody = d.vbody;
    wbody = d.wbody;
  }
  if (valid) {  // position_setpoint_task.py:194-203: target - p | q | v_body | w_body
    float *o = obs + (size_t)i * 13;
    const float e = tgt - p;
    if (l < 3) { o[l] = e; o[7 + l] = vbody; o[10 + l] = wbody; }
    o[3 + l] = q;
    if (rows) {
      float *r = rows + (size_t)i * 16;
      if (B.push_world > 0) {
        // peer push: lane l stores elements 4 l .. 4 l + 3 of the row (e0 e1 e2 q0 | q1 q2 q3 vb0 | vb1 vb2 wb0 wb1 | wb2 reward
        // crashed truncated): the quad writes its env's 64-byte row as ONE line per destination, a wave 1 KB contiguous.
        // (the permutes are evaluated on the whole quad before the per-lane pick)
        const float e1 = q4::bc<1>(e), e2 = q4::bc<2>(e), q2 = q4::perm<0, 2, 2, 3>(q), q3 = q4::bc<3>(q);
        const float vb0 = q4::bc<0>(vbody), vb1 = q4::bc<1>(vbody), wb0 = q4::bc<0>(wbody), wb1 = q4::bc<1>(wbody), wb2 = q4::bc<2>(wbody);
        float t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        if (l == 3) {
          t1 = tail_reward;
          t2 = tail_crashed ? 1.0f : 0.0f;
          t3 = tail_truncated ? 1.0f : 0.0f;
        }
        const float x0 = q4::by_lane(l, e, q, vb1, wb2);      // e0 (own) | q1 (own) | vb1 | wb2
        const float x1 = q4::by_lane(l, e1, q2, vbody, t1);   // e1 | q2 | vb2 (own) | reward
        const float x2 = q4::by_lane(l, e2, q3, wb0, t2);     // e2 | q3 | wb0 | crashed
        const float x3 = q4::by_lane(l, q, vb0, wb1, t3);     // q0 (own) | vb0 | wb1 | truncated
        row_store4_push(B, r + 4 * l, x0, x1, x2, x3);
      } else {
        if (l < 3) { row_store(B, r + l, e); row_store(B, r + 7 + l, vbody); row_store(B, r + 10 + l, wbody); }
        row_store(B, r + 3 + l, q);
        if (l == 0) {  // write_step_row_tail on the values loaded at the top
          row_store(B, r + 13, tail_reward);
          row_store(B, r + 14, tail_crashed ? 1.0f : 0.0f);
          row_store(B, r + 15, tail_truncated ? 1.0f : 0.0f);
        }
      }
    }
  }
  step_rows_signal(B);
}
__global__ void __launch_bounds__(64, 1) k_reset_masked_quad_obs(AgxRobotParams P, AgxEnvBuffers B, int n, AgxResetArgs R,
                                                                 const float *__restrict__ target, float *__restrict__ obs) {
  reset_masked_quad_obs_body<false>(P, B, n, R, target, obs);
}
__global__ void __launch_bounds__(64, 1) k_reset_masked_quad_obs_host_draws(AgxRobotParams P, AgxEnvBuffers B, int n, AgxResetArgs R,
                                                                            const float *__restrict__ target,
                                                                            float *__restrict__ obs) {
  reset_masked_quad_obs_body<true>(P, B, n, R, target, obs);
}

// AssetManager.reset_idx (asset_manager.py:51-71) + the half-obstacle resample (env_manager.py:283-295)
__global__ void __launch_bounds__(256) k_reset_assets(AgxEnvBuffers B, int n, int K, AgxResetArgs R, const float *__restrict__ u1,
                                                       const float *__restrict__ u2, const float *__restrict__ u_sel,
                                                       const float *__restrict__ min_ratio, const float *__restrict__ max_ratio,
                                                       int num_obstacles, int nk, float *__restrict__ asset_state) {
  // env on grid.x (2^31 blocks), asset chunk on grid.y: HIP caps grid.y at 65535, the env count has no such bound
  const int env = blockIdx.x;
  const int a = blockIdx.y * blockDim.x + threadIdx.x;
  if (a >= K) return;
  if (B.reset_flag[B.flag_parity] == 0 || B.reset_mask[env] == 0) return;
  reset_asset_one(B, R, env, a, K, u1, u2, u_sel, min_ratio, max_ratio, num_obstacles, nk, asset_state);
}

}  // namespace agx

// =========================================================================================
// C ABI
// =========================================================================================
using namespace agx;

// instantiate the step kernel for every (motor count, controller) pair in use by the reference's
// multirotor configs: 4 (quad, lmf*, x500, magpie), 6, 8 (octarotor) motors
#define AGX_DISPATCH_M(M_, ...)                                       \
  switch (M_) {                                                       \
    case 4: { constexpr int kM = 4; __VA_ARGS__; } break;             \
    case 6: { constexpr int kM = 6; __VA_ARGS__; } break;             \
    case 8: { constexpr int kM = 8; __VA_ARGS__; } break;             \
    default: return fail(AGX_E_UNSUPPORTED, "num_motors %d: kernels are built for 4, 6 and 8 motors", M_); \
  }
#define AGX_DISPATCH_CTRL(C_, ...)                                                         \
  switch (C_) {                                                                            \
    case AGX_CTRL_NONE: { constexpr int kC = AGX_CTRL_NONE; __VA_ARGS__; } break;          \
    case AGX_CTRL_POSITION: { constexpr int kC = AGX_CTRL_POSITION; __VA_ARGS__; } break;  \
    case AGX_CTRL_VELOCITY: { constexpr int kC = AGX_CTRL_VELOCITY; __VA_ARGS__; } break;  \
    case AGX_CTRL_ATTITUDE: { constexpr int kC = AGX_CTRL_ATTITUDE; __VA_ARGS__; } break;  \
    case AGX_CTRL_RATES: { constexpr int kC = AGX_CTRL_RATES; __VA_ARGS__; } break;        \
    case AGX_CTRL_ACCELERATION: { constexpr int kC = AGX_CTRL_ACCELERATION; __VA_ARGS__; } break; \
    case AGX_CTRL_VEL_STEERING: { constexpr int kC = AGX_CTRL_VEL_STEERING; __VA_ARGS__; } break; \
    case AGX_CTRL_FULLY_ACTUATED: { constexpr int kC = AGX_CTRL_FULLY_ACTUATED; __VA_ARGS__; } break; \
    case AGX_CTRL_WRENCH: { constexpr int kC = AGX_CTRL_WRENCH; __VA_ARGS__; } break;      \
    default: return fail(AGX_E_ARG, "unknown controller id %d", C_);                       \
  }

static int check_common(const AgxRobotParams *P, const AgxEnvBuffers *B, int n) {
  AGX_REQUIRE(B != nullptr, "null buffers");
  AGX_REQUIRE(n > 0, "num_envs must be > 0 (got %d)", n);
  // (SoaRef: a tensor's [<= 16][N] floats are addressed with 32-bit byte offsets from its base)
  AGX_REQUIRE(n <= (1 << 26), "num_envs %d above 2^26 = 67 108 864 per GPU: shard the job (the SoA accesses use 32-bit byte offsets)", n);
  AGX_REQUIRE(B->flag_parity == 0 || B->flag_parity == 1, "flag_parity must be 0 or 1");
  AGX_REQUIRE((!B->step_rows[0] && !B->step_rows[1]) ||
                  (B->step_rows[0] && B->step_rows[1] && B->step_reward && B->crashes && B->truncations),
              "step_rows needs both parity buffers, step_reward, crashes and truncations");
  if (P) {
    AGX_REQUIRE(P->num_motors >= 1 && P->num_motors <= AGX_MAX_MOTORS, "num_motors out of range");
    AGX_REQUIRE(P->num_actions >= 1 && P->num_actions <= AGX_MAX_ACTIONS, "num_actions out of range");
    AGX_REQUIRE(P->controller >= 0 && P->controller <= AGX_CTRL_WRENCH, "unknown controller id %d", P->controller);
    AGX_REQUIRE(P->controller != AGX_CTRL_FULLY_ACTUATED || P->num_actions == 7, "fully actuated controller needs 7 actions");
    AGX_REQUIRE(P->controller != AGX_CTRL_NONE || P->num_actions == P->num_motors, "no_control needs num_actions == num_motors");
    AGX_REQUIRE(P->controller == AGX_CTRL_NONE || P->controller == AGX_CTRL_FULLY_ACTUATED || P->controller == AGX_CTRL_WRENCH ||
                    P->num_actions == 4,
                "Lee controllers take 4 actions");
  }
  return AGX_OK;
}

// The four-lanes-per-env kernel covers the plain quadrotor position step; agx_set_option("env_step_quad", 0) keeps k_env_step (A/B runs).
static bool quad_kernel_usable(const AgxRobotParams *P, const AgxEnvBuffers *B, const AgxTaskArgs *T) {
  if (!option_env_step_quad() || B->boxes || B->launch_flags != 0 || B->disturb || B->disturb_prob > 0.0f || P->num_actions != 4) return false;
  if (T->kind != AGX_TASK_NONE && T->kind != AGX_TASK_POSITION) return false;
  for (int c = 0; c < 3; ++c)
    if (P->lin_drag_linear[c] != 0.0f || P->lin_drag_quadratic[c] != 0.0f || P->ang_drag_linear[c] != 0.0f ||
        P->ang_drag_quadratic[c] != 0.0f)
      return false;
  return true;
}

// ... and the sub-step loop of the velocity / acceleration controlled quadrotors (navigation tasks)
static bool quad_loop_kernel_usable(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, int k) {
  if (!option_env_step_quad() || pick_block(n) != 64 || k < 1 || B->launch_flags != 0) return false;
  const bool lee_quad = P->num_motors == 4 && P->num_actions == 4 && P->controller >= AGX_CTRL_POSITION &&
                        P->controller <= AGX_CTRL_VEL_STEERING;
  const bool fa_octa = P->num_motors == 8 && P->num_actions == 7 && P->controller == AGX_CTRL_FULLY_ACTUATED;
  const bool lee_octa = P->num_motors == 8 && P->num_actions == 4 && (P->controller == AGX_CTRL_POSITION || P->controller == AGX_CTRL_VELOCITY);
  if (!lee_quad && !fa_octa && !lee_octa) return false;
  for (int c = 0; c < 3; ++c)
    if (P->lin_drag_linear[c] != 0.0f || P->lin_drag_quadratic[c] != 0.0f || P->ang_drag_linear[c] != 0.0f ||
        P->ang_drag_quadratic[c] != 0.0f)
      return false;
  return true;
}

template <int M, int CTRL, bool WIDE>
static void launch_env_step(int k, int n, int block, size_t lds, hipStream_t stream, const AgxRobotParams &P, const AgxEnvBuffers &B,
                            const float *actions_in, const AgxTaskArgs &T) {
  if (k == 1)
    hipLaunchKernelGGL((k_env_step<M, CTRL, true, WIDE>), dim3(blocks_for(n, block)), dim3(block), lds, stream, P, B, n, actions_in, k, T);
  else
    hipLaunchKernelGGL((k_env_step<M, CTRL, false, WIDE>), dim3(blocks_for(n, block)), dim3(block), lds, stream, P, B, n, actions_in, k, T);
}

extern "C" int agx_env_step(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const float *actions_in, int k,
                            const AgxTaskArgs *task, void *stream) {
  if (int e = check_common(P, B, n)) return e;
  AGX_REQUIRE(P != nullptr, "null params");
  AGX_REQUIRE(actions_in != nullptr, "null actions");
  AGX_REQUIRE(k >= 0 && k <= AGX_MAX_SUBSTEPS, "k_substeps out of range: %d", k);
  AGX_REQUIRE(P->controller != AGX_CTRL_WRENCH || k <= 1,
              "external controller (AGX_CTRL_WRENCH): one launch per physics sub-step, the host re-evaluates the controller in between");
  AGX_REQUIRE((B->launch_flags & ~0xFF0F) == 0, "launch_flags: bits 0-3 and the sub-step index in bits 8-15");
  AGX_REQUIRE((B->launch_flags & AGX_LAUNCH_BODY_WRENCH) == 0 || P->controller == AGX_CTRL_WRENCH,
              "AGX_LAUNCH_BODY_WRENCH (external robot) goes with AGX_CTRL_WRENCH: actions_in is a wrench [N][6]");
  AGX_REQUIRE((B->launch_flags & 4) == 0 || ((B->launch_flags & 3) == 0 && P->controller != AGX_CTRL_WRENCH &&
                                             (!task || task->kind != AGX_TASK_NAVIGATION)),
              "AGX_LAUNCH_LEAN needs the fused step of a built-in controller without the navigation reward (it reads the action history)");
  AGX_REQUIRE(B->state && B->derived && B->actions && B->prev_actions && B->motor_thrust && B->crashes && B->truncations &&
                  B->sim_steps,
              "null env buffer");
  AGX_REQUIRE(!P->use_rps || B->motor_kT, "null motor_kT with use_rps");
  AgxTaskArgs T{};
  if (task) T = *task;
  AGX_REQUIRE(T.kind >= AGX_TASK_NONE && T.kind <= AGX_TASK_NAVIGATION, "bad task kind %d", T.kind);
  if (T.kind != AGX_TASK_NONE) {
    AGX_REQUIRE(T.target && T.reward && B->reset_mask && B->reset_flag, "null task buffer");
    AGX_REQUIRE(T.kind != AGX_TASK_NAVIGATION || (T.pos_err && T.prev_pos_err && P->num_actions >= 4), "null navigation buffer");
    AGX_REQUIRE((!T.successes && !T.timeouts && !T.counters) || (T.successes && T.timeouts && T.counters && T.kind == AGX_TASK_NAVIGATION),
                "navigation bookkeeping in the epilogue needs successes, timeouts and counters, and the navigation task kind");
  }
  const int block = pick_block(n);
  const size_t lds = B->boxes ? (size_t)k * 3 * block * sizeof(float) : 0;
  if (k == 1 && block == 64 && P->num_motors == 4 && P->controller == AGX_CTRL_POSITION && quad_kernel_usable(P, B, &T)) {
    hipLaunchKernelGGL(k_env_step_quad_position, dim3(blocks_for(n, 16)), dim3(64), 0, (hipStream_t)stream, *P, *B, n, actions_in,
                       T);
    return check_launch("agx_env_step");
  }
  if (quad_loop_kernel_usable(P, B, n, k)) {
    const size_t lds4 = B->boxes ? (size_t)k * 3 * 16 * sizeof(float) : 0;
    if (P->num_motors == 8 && P->controller == AGX_CTRL_POSITION) {
      hipLaunchKernelGGL((k_env_step_quad_loop<8, AGX_CTRL_POSITION>), dim3(blocks_for(n, 16)), dim3(64), lds4, (hipStream_t)stream, *P, *B, n,
                         actions_in, k, T);
      return check_launch("agx_env_step");
    }
    if (P->num_motors == 8 && P->controller == AGX_CTRL_VELOCITY) {
      hipLaunchKernelGGL((k_env_step_quad_loop<8, AGX_CTRL_VELOCITY>), dim3(blocks_for(n, 16)), dim3(64), lds4, (hipStream_t)stream, *P, *B, n,
                         actions_in, k, T);
      return check_launch("agx_env_step");
    }
    switch (P->controller) {
#define AGX_QUAD_LOOP(C_)                                                                                                        \
  case C_:                                                                                                                       \
    hipLaunchKernelGGL((k_env_step_quad_loop<4, C_>), dim3(blocks_for(n, 16)), dim3(64), lds4, (hipStream_t)stream, *P, *B, n,     \
                       actions_in, k, T);                                                                                        \
    break;
      AGX_QUAD_LOOP(AGX_CTRL_POSITION)
      AGX_QUAD_LOOP(AGX_CTRL_VELOCITY)
      AGX_QUAD_LOOP(AGX_CTRL_ATTITUDE)
      AGX_QUAD_LOOP(AGX_CTRL_RATES)
      AGX_QUAD_LOOP(AGX_CTRL_ACCELERATION)
      AGX_QUAD_LOOP(AGX_CTRL_VEL_STEERING)
#undef AGX_QUAD_LOOP
      case AGX_CTRL_FULLY_ACTUATED:
        hipLaunchKernelGGL((k_env_step_quad_loop<8, AGX_CTRL_FULLY_ACTUATED>), dim3(blocks_for(n, 16)), dim3(64), lds4, (hipStream_t)stream,
                           *P, *B, n, actions_in, k, T);
        break;
      default: break;
    }
    return check_launch("agx_env_step");
  }
  AGX_DISPATCH_M(P->num_motors,
                 AGX_DISPATCH_CTRL(P->controller, {
                   if (block == 64)
                     launch_env_step<kM, kC, true>(k, n, block, lds, (hipStream_t)stream, *P, *B, actions_in, T);
                   else
                     launch_env_step<kM, kC, false>(k, n, block, lds, (hipStream_t)stream, *P, *B, actions_in, T);
                 }));
  return check_launch("agx_env_step");
}

extern "C" int agx_env_step_kernel(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, int k, const AgxTaskArgs *task, char *out,
                                   int cap) {
  AGX_REQUIRE(P && B && out && cap > 0 && n > 0, "bad arguments");
  AgxTaskArgs T{};
  if (task) T = *task;
  const int block = pick_block(n);
  if (k == 1 && block == 64 && P->num_motors == 4 && P->controller == AGX_CTRL_POSITION && quad_kernel_usable(P, B, &T))
    snprintf(out, (size_t)cap, "k_env_step_quad_position_%d", blocks_for(n, 16) * 64);
  else if (quad_loop_kernel_usable(P, B, n, k))
    snprintf(out, (size_t)cap, "k_env_step_quad_loop<%d,%d>_%d", P->num_motors, P->controller, blocks_for(n, 16) * 64);
  else
    snprintf(out, (size_t)cap, "k_env_step<%d,%d,%s,%s>_%d", P->num_motors, P->controller, k == 1 ? "true" : "false",
             block == 64 ? "true" : "false", blocks_for(n, block) * block);
  return AGX_OK;
}

extern "C" int agx_dynamics_substeps(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const float *actions_in,
                                     int k, void *stream) {
  return agx_env_step(P, B, n, actions_in, k, nullptr, stream);
}

extern "C" int agx_update_states(const AgxEnvBuffers *B, int n, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(B->state && B->derived, "null env buffer");
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_update_states, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *B, n);
  return check_launch("agx_update_states");
}

extern "C" int agx_collide_spheres_boxes(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, void *stream) {
  AGX_REQUIRE(P && B, "bad arguments");
  if (int e = check_common(P, B, n)) return e;  // the n <= 2^26 bound the 32-bit SoaRef offsets of the kernel depend on
  AGX_REQUIRE(B->state && B->crashes, "null env buffer");
  if (!B->boxes || B->num_boxes <= 0) return AGX_OK;  // no obstacles: nothing can be hit
  hipLaunchKernelGGL(k_collide_spheres_boxes, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, P->collision_radius);
  return check_launch("agx_collide_spheres_boxes");
}

extern "C" int agx_controller_wrench(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const float *action,
                                     void *stream) {
  if (int e = check_common(P, B, n)) return e;
  AGX_REQUIRE(P && P->controller != AGX_CTRL_NONE, "controller required");
  AGX_REQUIRE(action && B->state && B->derived && B->wrench_cmd, "null buffer");
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_controller_wrench, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *P, *B, n, action);
  return check_launch("agx_controller_wrench");
}

extern "C" int agx_robot_step(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const float *action, const AgxRobotStepArgs *R,
                              void *stream) {
  if (int e = check_common(P, B, n)) return e;
  AGX_REQUIRE(P && R && action, "null argument");
  AGX_REQUIRE(P->controller != AGX_CTRL_WRENCH, "agx_robot_step evaluates a BUILT-IN controller (an external controller class is called by the host)");
  AGX_REQUIRE(B->state && B->derived && B->motor_thrust && R->force && R->torque, "null buffer");
  AGX_REQUIRE(!P->use_rps || B->motor_kT, "null motor_kT with use_rps");
  AGX_REQUIRE(R->num_bodies >= 1 && R->num_bodies <= AGX_MAX_BODIES, "num_bodies %d outside [1, %d]", R->num_bodies, AGX_MAX_BODIES);
  AGX_REQUIRE(R->substep >= 0 && R->substep < AGX_MAX_SUBSTEPS, "substep out of range");
  AGX_REQUIRE((long long)n * R->num_bodies * 3 < (1ll << 31), "per-body tensors too large for this entry point");
  for (int j = 0; j < (P->root_link_mode ? 1 : P->num_motors); ++j)
    AGX_REQUIRE(R->body_of_motor[j] >= 0 && R->body_of_motor[j] < R->num_bodies, "application mask entry %d = %d outside [0, %d)", j,
                R->body_of_motor[j], R->num_bodies);
  hipLaunchKernelGGL(k_robot_step, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *P, *B, n, action, *R);
  return check_launch("agx_robot_step");
}

extern "C" int agx_net_body_wrench(int n, const AgxLinkFrames *L, const float *force, const float *torque, float *out, void *stream) {
  AGX_REQUIRE(n > 0 && L && force && torque && out, "bad arguments");
  AGX_REQUIRE(L->num_bodies >= 1 && L->num_bodies <= AGX_MAX_BODIES, "num_bodies %d outside [1, %d]", L->num_bodies, AGX_MAX_BODIES);
  hipLaunchKernelGGL(k_net_body_wrench, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, n, *L, force, torque, out);
  return check_launch("agx_net_body_wrench");
}

extern "C" int agx_reward_position(const AgxEnvBuffers *B, int n, const float *target, int episode_len,
                                   int reset_on_collision, float *reward, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(target && reward && B->reset_flag && B->reset_mask && B->state && B->derived && B->crashes && B->truncations &&
                  B->sim_steps,
              "null buffer");
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_reward_position, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *B, n, target,
                     episode_len, reset_on_collision, reward);
  return check_launch("agx_reward_position");
}

extern "C" int agx_obs_position(const AgxEnvBuffers *B, int n, const float *target, float *obs, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(target && obs && B->state && B->derived, "null buffer");
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_obs_position, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *B, n, target, obs);
  return check_launch("agx_obs_position");
}

extern "C" int agx_reward_navigation(const AgxEnvBuffers *B, int n, const float *target, const float *rp, float cpf,
                                     float *pos_err, float *prev_pos_err, int episode_len, int reset_on_collision,
                                     float *reward, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(target && rp && pos_err && prev_pos_err && reward && B->reset_flag && B->reset_mask, "null buffer");
  AGX_REQUIRE(B->state && B->derived && B->actions && B->prev_actions && B->crashes && B->truncations && B->sim_steps,
              "null env buffer");
  NavParams R;
  for (int c = 0; c < 18; ++c) R.rp[c] = rp[c];  // rp is a HOST pointer (18 config scalars)
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_reward_navigation, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *B, n, target, R, cpf,
                     pos_err, prev_pos_err, episode_len, reset_on_collision, reward);
  return check_launch("agx_reward_navigation");
}

extern "C" int agx_obs_navigation(const AgxEnvBuffers *B, int n, const float *target, const float *u_vec,
                                  const float *u_euler, const float *pixels, int ns, int H, int W, int gh, int gw,
                                  int obs_dim, float *obs, float *min_pixel, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(!min_pixel || (pixels && ns == 1), "min_pixel: needs the image, and covers it only with one sensor");
  AGX_REQUIRE(target && obs && B->state && B->derived && B->actions, "null buffer");
  AGX_REQUIRE((u_vec == nullptr) == (u_euler == nullptr), "u_vec and u_euler: both tensors or both NULL (device generator)");
  AGX_REQUIRE(obs_dim >= 17, "obs_dim must be >= 17");
  AGX_REQUIRE(!pixels || (ns > 0 && H > 0 && W > 0 && gh > 0 && gw > 0), "bad image sizes");
  if (pixels && n <= 2048 && gh >= 4)
    hipLaunchKernelGGL(k_obs_navigation_split, dim3(n), dim3(256), 0, (hipStream_t)stream, *B, n, target, u_vec, u_euler, pixels, ns, H,
                       W, gh, gw, obs_dim, obs, min_pixel);
  else
    hipLaunchKernelGGL(k_obs_navigation, dim3(blocks_for(n, 4)), dim3(256), 0, (hipStream_t)stream, *B, n, target, u_vec,
                       u_euler, pixels, ns, H, W, gh, gw, obs_dim, obs, min_pixel);
  return check_launch("agx_obs_navigation");
}

static int check_reset(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const AgxResetArgs *R) {
  if (int e = check_common(P, B, n)) return e;
  AGX_REQUIRE(P && R && B->reset_flag && B->reset_mask && B->bounds_min && B->bounds_max, "null argument");
  if (R->u_state) {
    AGX_REQUIRE(R->u_bounds_lo && R->u_bounds_hi && R->u_tau_inc && R->u_tau_dec && R->u_thrust, "null reset input");
    AGX_REQUIRE(!P->use_rps || R->u_kT, "null u_kT with use_rps");
    AGX_REQUIRE(!R->randomize_gains || R->u_gains, "null u_gains with randomize_gains");
  } else {
    AGX_REQUIRE(B->episode_count, "device RNG needs buf->episode_count");
  }
  AGX_REQUIRE(!R->randomize_gains || B->gains, "null gains with randomize_gains");
  return AGX_OK;
}

extern "C" int agx_reset_masked(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const AgxResetArgs *R,
                                void *stream) {
  if (int e = check_reset(P, B, n, R)) return e;
  const int block = pick_block(n);
  AGX_DISPATCH_M(P->num_motors, hipLaunchKernelGGL((k_reset_masked<kM, false>), dim3(blocks_for(n, block)), dim3(block), 0,
                                                   (hipStream_t)stream, *P, *B, n, *R, nullptr, nullptr));
  return check_launch("agx_reset_masked");
}

extern "C" int agx_nav_robot_side(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const AgxResetArgs *R,
                                  const AgxNavRobotSideArgs *A, void *stream) {
  if (int e = check_reset(P, B, n, R)) return e;
  AGX_REQUIRE(A, "null AgxNavRobotSideArgs");
  AGX_REQUIRE(R->u_state == nullptr, "agx_nav_robot_side draws with the device generator only (sync-free mode)");
  AGX_REQUIRE(A->num_sensors >= 0 && (A->num_sensors == 0 || (A->local_pos && A->local_quat && A->sensor_pos && A->sensor_quat)),
              "sensor buffers missing");
  AGX_REQUIRE(!A->reset_target || (A->target && B->bounds_min && B->bounds_max), "target part needs target and the env bounds");
  AGX_REQUIRE((!A->reset_target && !(A->num_sensors && A->randomize_mount)) || B->episode_count, "device RNG needs buf->episode_count");
  AGX_REQUIRE(!A->zero_prev_actions || B->prev_actions, "zero_prev_actions needs buf->prev_actions");
  AGX_DISPATCH_M(P->num_motors, hipLaunchKernelGGL((k_nav_robot_side<kM>), dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                                                   *P, *B, n, *R, *A));
  return check_launch("agx_nav_robot_side");
}

extern "C" int agx_post_step_position(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const AgxResetArgs *R,
                                      const float *target, float *obs, void *stream) {
  if (int e = check_reset(P, B, n, R)) return e;
  AGX_REQUIRE(target && obs && B->state && B->derived, "null buffer");
  const int block = pick_block(n);
  if (block == 64 && P->num_motors == 4 && option_env_step_quad()) {
    if (R->u_state)
      hipLaunchKernelGGL(k_reset_masked_quad_obs_host_draws, dim3(blocks_for(n, 16)), dim3(64), 0, (hipStream_t)stream, *P, *B, n, *R,
                         target, obs);
    else
      hipLaunchKernelGGL(k_reset_masked_quad_obs, dim3(blocks_for(n, 16)), dim3(64), 0, (hipStream_t)stream, *P, *B, n, *R, target, obs);
    return check_launch("agx_post_step_position");
  }
  AGX_DISPATCH_M(P->num_motors, hipLaunchKernelGGL((k_reset_masked<kM, true>), dim3(blocks_for(n, block)), dim3(block), 0,
                                                   (hipStream_t)stream, *P, *B, n, *R, target, obs));
  return check_launch("agx_post_step_position");
}

extern "C" int agx_push_advance(AgxEnvBuffers *b) {
  AGX_REQUIRE(b && b->push_world > 0 && b->push_world <= 8 && b->push_slots >= 5 && b->push_base && b->push_slice_bytes > 0,
              "agx_push_advance: peer push is not bound (or fewer than 5 receive slots)");
  b->push_pub_seq = b->push_seq;  // the rows of the step before are announced by the first kernel of this one
  b->push_pub_index = b->push_flag_index;
  const uint32_t seq = ++b->push_seq;
  const int slot = (int)((seq - 1u) % (uint32_t)b->push_slots);
  b->push_flag_index = slot * b->push_world + b->push_rank;
  b->step_rows[0] = b->step_rows[1] = (float *)(b->push_base + ((size_t)slot * b->push_world + b->push_rank) * (size_t)b->push_slice_bytes);
  if (seq > 2u) {
    b->push_wait_seq = seq - 2u;
    b->push_wait_index = (int)((seq - 3u) % (uint32_t)b->push_slots) * b->push_world;
  } else {
    b->push_wait_seq = 0u;
  }
  return AGX_OK;
}

extern "C" int agx_position_task_step(const AgxPositionStepPlan *plan, const float *actions_in, void *stream) {
  AGX_REQUIRE(plan && plan->params && plan->buf && plan->task && plan->reset, "null plan member");
  plan->buf->flag_parity ^= 1;  // new env step: the flag the previous step's reset kernel cleared
  if (plan->buf->push_world > 0)
    if (int e = agx_push_advance(plan->buf)) return e;
  if (int e = agx_env_step(plan->params, plan->buf, plan->num_envs, actions_in, plan->k_substeps, plan->task, stream)) return e;
  // peer push: the env-step kernel has waited (one wave) until the slot of this step's rows was vacated; the observation
  // kernel behind it need not look again
  const uint32_t wait_seq = plan->buf->push_wait_seq;
  plan->buf->push_wait_seq = 0;
  const int rc = agx_post_step_position(plan->params, plan->buf, plan->num_envs, plan->reset, plan->target, plan->obs, stream);
  plan->buf->push_wait_seq = wait_seq;
  return rc;
}

extern "C" int agx_reset_assets(const AgxEnvBuffers *B, int n, int K, const AgxResetArgs *R, const float *u1, const float *u2,
                                const float *u_sel, const float *min_ratio, const float *max_ratio, int num_obstacles,
                                int num_keep, float *asset_state, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(K > 0 && R && min_ratio && max_ratio && asset_state && B->reset_flag && B->reset_mask, "bad arguments");
  AGX_REQUIRE((u1 && u2 && u_sel && R->u_state) || (!u1 && !u2 && !u_sel && !R->u_state),
              "asset draws and robot draws must both come from tensors or both from the device generator");
  AGX_REQUIRE(u1 || B->episode_count, "device RNG needs buf->episode_count");
  AGX_REQUIRE(blocks_for(K, 64) <= 65535, "too many assets per env");
  dim3 grid(n, blocks_for(K, 64));
  hipLaunchKernelGGL(k_reset_assets, grid, dim3(64), 0, (hipStream_t)stream, *B, n, K, *R, u1, u2, u_sel, min_ratio, max_ratio,
                     num_obstacles, num_keep, asset_state);
  return check_launch("agx_reset_assets");
}
