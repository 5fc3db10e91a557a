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
#include "agx_device_math.h"

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

AGX_DEV EnvState load_state(const float *__restrict__ s, int n, int i) {
  EnvState e;
  e.p = V3{s[0 * n + i], s[1 * n + i], s[2 * n + i]};
  e.q = Q4{s[3 * n + i], s[4 * n + i], s[5 * n + i], s[6 * n + i]};
  e.v = V3{s[7 * n + i], s[8 * n + i], s[9 * n + i]};
  e.w = V3{s[10 * n + i], s[11 * n + i], s[12 * n + i]};
  return e;
}
AGX_DEV void store_state(float *__restrict__ s, int n, int i, const EnvState &e) {
  s[0 * n + i] = e.p.x; s[1 * n + i] = e.p.y; s[2 * n + i] = e.p.z;
  s[3 * n + i] = e.q.x; s[4 * n + i] = e.q.y; s[5 * n + i] = e.q.z; s[6 * n + i] = e.q.w;
  s[7 * n + i] = e.v.x; s[8 * n + i] = e.v.y; s[9 * n + i] = e.v.z;
  s[10 * n + i] = e.w.x; s[11 * n + i] = e.w.y; s[12 * n + i] = e.w.z;
}
AGX_DEV void store_derived(float *__restrict__ d, int n, int i, const Derived &x) {
  d[0 * n + i] = x.euler.x; d[1 * n + i] = x.euler.y; d[2 * n + i] = x.euler.z;
  d[3 * n + i] = x.qveh.x; d[4 * n + i] = x.qveh.y; d[5 * n + i] = x.qveh.z; d[6 * n + i] = x.qveh.w;
  d[7 * n + i] = x.vveh.x; d[8 * n + i] = x.vveh.y; d[9 * n + i] = x.vveh.z;
  d[10 * n + i] = x.vbody.x; d[11 * n + i] = x.vbody.y; d[12 * n + i] = x.vbody.z;
  d[13 * n + i] = x.wbody.x; d[14 * n + i] = x.wbody.y; d[15 * n + i] = x.wbody.z;
}
AGX_DEV Derived load_derived(const float *__restrict__ d, int n, int i) {
  Derived x;
  x.euler = V3{d[0 * n + i], d[1 * n + i], d[2 * n + i]};
  x.qveh = Q4{d[3 * n + i], d[4 * n + i], d[5 * n + i], d[6 * n + i]};
  x.vveh = V3{d[7 * n + i], d[8 * n + i], d[9 * n + i]};
  x.vbody = V3{d[10 * n + i], d[11 * n + i], d[12 * n + i]};
  x.wbody = V3{d[13 * n + i], d[14 * n + i], d[15 * n + i]};
  return x;
}
AGX_DEV Gains load_gains(const float *__restrict__ g, int n, int i) {
  Gains k;
  k.kp = V3{g[0 * n + i], g[1 * n + i], g[2 * n + i]};
  k.kv = V3{g[3 * n + i], g[4 * n + i], g[5 * n + i]};
  k.kr = V3{g[6 * n + i], g[7 * n + i], g[8 * n + i]};
  k.kw = V3{g[9 * n + i], g[10 * n + i], g[11 * n + i]};
  return k;
}

// BaseMultirotor.update_states, base_multirotor.py:287-294
AGX_DEV Derived update_states(const EnvState &s) {
  Derived d;
  V3 e = euler_xyz_0_2pi(s.q);
  d.euler = V3{ssa(e.x), ssa(e.y), ssa(e.z)};
  // vehicle_frame_quat_from_quat: euler * [0, 0, 1] (utils/math.py:176-180)
  d.qveh = quat_from_euler(e.x * 0.0f, e.y * 0.0f, e.z * 1.0f);
  d.vveh = quat_rotate_inverse(d.qveh, s.v);
  d.vbody = quat_rotate_inverse(s.q, s.v);
  d.wbody = quat_rotate_inverse(s.q, s.w);
  return d;
}

// base_lee_controller.py:120-134
AGX_DEV V3 compute_acceleration(const EnvState &s, Q4 qveh, V3 sp_pos, V3 sp_vel, const Gains &g) {
  V3 sp_vel_w = quat_rotate(qveh, sp_vel);
  V3 pe = sp_pos - s.p;
  V3 ve = sp_vel_w - s.v;
  return V3{g.kp.x * pe.x + g.kv.x * ve.x, g.kp.y * pe.y + g.kv.y * ve.y, g.kp.z * pe.z + g.kv.z * ve.z};
}

// base_lee_controller.py:136-154 (sp_w.z is clamped in place by the caller-visible ref)
AGX_DEV V3 compute_body_torque(const AgxRobotParams &P, Q4 q, V3 wb, Q4 qd, V3 &sp_w, const Gains &g) {
  sp_w.z = fminf(fmaxf(sp_w.z, -P.max_yaw_rate), P.max_yaw_rate);
  Q4 qe = quat_mul(conj(q), qd);
  M33 R = quat_to_rotmat(qe);
  V3 rot_err = V3{0.5f * (-(R.m21 - R.m12)), 0.5f * (R.m20 - R.m02), 0.5f * (-(R.m10 - R.m01))};
  V3 wsp_b = quat_rotate(qe, sp_w);
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
  V3 b3 = f / norm(f);
  V3 tmp = V3{cosf(yaw), sinf(yaw), 0.0f};
  V3 b2 = cross(b3, tmp);
  b2 = b2 / norm(b2);
  V3 b1 = cross(b2, b3);
  M33 R{b1.x, b2.x, b3.x, b1.y, b2.y, b3.y, b1.z, b2.z, b3.z};
  return rotmat_to_quat(R);
}

// base_lee_controller.py:158-169
AGX_DEV Q4 desired_orientation_forces_yaw(V3 f, float yaw) {
  float pitch = atan2f(f.x, f.z);
  float roll = atan2f(-f.y, sqrtf(f.z * f.z + f.x * f.x));
  return quat_from_euler(roll, pitch, yaw);
}

// base_lee_controller.py:201-215 (stale matrix entries only ever multiply zero rates)
AGX_DEV V3 euler_rates_to_body_rates(V3 euler, V3 r) {
  float sp = sinf(euler.y), cp = cosf(euler.y);
  float sr = sinf(euler.x), cr = cosf(euler.x);
  return V3{1.0f * r.x + 0.0f * r.y + (-sp) * r.z, 0.0f * r.x + cr * r.y + (sr * cp) * r.z,
            0.0f * r.x + (-sr) * r.y + (cr * cp) * r.z};
}

// One env's controller (control/controllers/*.py).  a[] holds the +-10 clipped action and
// is mutated where the reference mutates it.
AGX_DEV Wrench run_controller(const AgxRobotParams &P, const EnvState &s, const Derived &d, float (&a)[AGX_MAX_ACTIONS],
                              const Gains &g) {
  Wrench w{V3{0, 0, 0}, V3{0, 0, 0}};
  const V3 grav = V3{P.gravity[0], P.gravity[1], P.gravity[2]};
  const float m = P.mass;
  const V3 zero = V3{0, 0, 0};
  switch (P.controller) {
    case AGX_CTRL_POSITION: {  // position_control.py:20-51
      V3 acc = compute_acceleration(s, d.qveh, V3{a[0], a[1], a[2]}, zero, g);
      V3 f = (acc - grav) * m;
      M33 R = quat_to_rotmat(s.q);
      w.f.z = f.x * R.m02 + f.y * R.m12 + f.z * R.m22;
      Q4 qd = desired_orientation_pos_vel(f, a[3]);
      V3 wsp = zero;
      w.t = compute_body_torque(P, s.q, d.wbody, qd, wsp, g);
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
      V3 acc = compute_acceleration(s, d.qveh, V3{a[0], a[1], a[2]}, zero, g);
      V3 f = (acc - grav) * m;
      w.f = quat_rotate_inverse(s.q, f);
      V3 wsp = zero;
      w.t = compute_body_torque(P, s.q, d.wbody, Q4{a[3], a[4], a[5], a[6]}, wsp, g);
    } break;
    default: break;
  }
  return w;
}

// control/motor_model.py:88-250
AGX_DEV float clamp_minmax(float x, float lo, float hi) { return fmaxf(fminf(x, hi), lo); }
AGX_DEV float sgnf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
AGX_DEV float motor_rate(float err, float mix, float max_rate) { return clamp_minmax(mix * err, -max_rate, max_rate); }
AGX_DEV float rk4_delta(float ref, float cur, float mix, float max_rate, float dt) {
  float k1 = motor_rate(ref - cur, mix, max_rate);
  float k2 = motor_rate(ref - (cur + 0.5f * dt * k1), mix, max_rate);
  float k3 = motor_rate(ref - (cur + 0.5f * dt * k2), mix, max_rate);
  float k4 = motor_rate(ref - (cur + dt * k3), mix, max_rate);
  return (dt / 6.0f) * (k1 + 2.0f * k2 + 2.0f * k3 + k4);
}
AGX_DEV float motor_update(const AgxRobotParams &P, float ref, float cur, float kT, float tau_inc, float tau_dec) {
  const float dt = P.dt;
  ref = fminf(fmaxf(ref, P.min_thrust), P.max_thrust);
  float err = ref - cur;
  float tc = (sgnf(cur) * sgnf(err) < 0.0f) ? tau_dec : tau_inc;
  float mix = P.use_discrete_approximation ? 1.0f / (dt + tc) : 1.0f / tc;
  if (P.use_rps) {
    float cur_rpm = sqrtf(cur / kT);
    float des_rpm = sqrtf(ref / kT);
    if (P.integration_rk4)
      cur_rpm += rk4_delta(des_rpm, cur_rpm, mix, P.max_rate, dt);
    else
      cur_rpm += motor_rate(des_rpm - cur_rpm, mix, P.max_rate) * dt;
    return kT * (cur_rpm * cur_rpm);
  }
  if (P.integration_rk4) return cur + rk4_delta(ref, cur, mix, P.max_rate, dt);
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
  V3 v_new = V3{s.v.x + dt * (Fw.x / P.mass), s.v.y + dt * (Fw.y / P.mass), s.v.z + dt * (Fw.z / P.mass)};
  v_new = V3{v_new.x + P.gravity[0] * dt, v_new.y + P.gravity[1] * dt, v_new.z + P.gravity[2] * dt};
  float ml = fmaxf(1.0f - P.linear_damping * dt, 0.0f);
  float ma = fmaxf(1.0f - P.angular_damping * dt, 0.0f);
  v_new = v_new * ml;
  w_new = w_new * ma;
  float v2 = dot(v_new, v_new), w2 = dot(w_new, w_new);
  if (v2 > P.max_linear_velocity * P.max_linear_velocity) v_new = v_new * (P.max_linear_velocity / sqrtf(v2));
  if (w2 > P.max_angular_velocity * P.max_angular_velocity) w_new = w_new * (P.max_angular_velocity / sqrtf(w2));
  s.p = V3{s.p.x + v_new.x * dt, s.p.y + v_new.y * dt, s.p.z + v_new.z * dt};
  float wm2 = dot(w_new, w_new);
  if (wm2 != 0.0f) {
    float wm = sqrtf(wm2);
    float half = dt * wm * 0.5f;
    float sn = sinf(half), cs = cosf(half);
    float sc = sn / wm;
    float x1 = w_new.x * sc, y1 = w_new.y * sc, z1 = w_new.z * sc;
    Q4 q = s.q;
    float rx = x1 * q.w + y1 * q.z - z1 * q.y;
    float ry = y1 * q.w + z1 * q.x - x1 * q.z;
    float rz = z1 * q.w + x1 * q.y - y1 * q.x;
    float rw = -(x1 * q.x) - y1 * q.y - z1 * q.z;
    rx += q.x * cs; ry += q.y * cs; rz += q.z * cs; rw += q.w * cs;
    float nn = sqrtf(rx * rx + ry * ry + rz * rz + rw * rw);
    s.q = Q4{rx / nn, ry / nn, rz / nn, rw / nn};
  }
  s.v = v_new;
  s.w = w_new;
}

// sphere (robot collision sphere, quad.urdf:16) vs obstacle OBBs; replaces the PhysX
// contact-force test of env_manager.py:358-362
AGX_DEV bool collide_boxes(const float *__restrict__ boxes, int nb, int n, int i, V3 p, float r) {
  bool hit = false;
  const float r2 = r * r;
  for (int b = 0; b < nb; ++b) {
    const float *bx = boxes + (size_t)b * 10 * n + i;
    V3 c = V3{bx[0 * (size_t)n], bx[1 * (size_t)n], bx[2 * (size_t)n]};
    Q4 q = Q4{bx[3 * (size_t)n], bx[4 * (size_t)n], bx[5 * (size_t)n], bx[6 * (size_t)n]};
    V3 h = V3{bx[7 * (size_t)n], bx[8 * (size_t)n], bx[9 * (size_t)n]};
    V3 l = quat_rotate_inverse(q, p - c);
    float ex = fabsf(l.x) - h.x, ey = fabsf(l.y) - h.y, ez = fabsf(l.z) - h.z;
    float d2 = 0.0f;
    if (ex > 0.0f) d2 += ex * ex;
    if (ey > 0.0f) d2 += ey * ey;
    if (ez > 0.0f) d2 += ez * ez;
    hit = hit || (d2 < r2);
  }
  return hit;
}

template <int M>
__global__ void __launch_bounds__(256) k_dynamics_substeps(AgxRobotParams P, AgxEnvBuffers B, int n, const float *__restrict__ actions_in, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int A = P.num_actions;
  EnvState s = load_state(B.state, n, i);
  float u[M], kT[M], tinc[M], tdec[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    u[j] = B.motor_thrust[j * n + i];
    kT[j] = P.use_rps ? B.motor_kT[j * n + i] : 1.0f;
    tinc[j] = B.motor_tau_inc[j * n + i];
    tdec[j] = B.motor_tau_dec[j * n + i];
  }
  Gains g{};
  if (P.controller != AGX_CTRL_NONE) g = load_gains(B.gains, n, i);
  float a_in[AGX_MAX_ACTIONS], a_old[AGX_MAX_ACTIONS];
#pragma unroll
  for (int c = 0; c < AGX_MAX_ACTIONS; ++c) {
    a_in[c] = (c < A) ? actions_in[(size_t)i * A + c] : 0.0f;
    a_old[c] = (c < A) ? B.actions[c * n + i] : 0.0f;
  }
  // EnvManager.reset_tensors (env_manager.py:342-344)
  bool crashed = false;
  Derived d{};
  Wrench wc{V3{0, 0, 0}, V3{0, 0, 0}};
  if (i == 0) *B.reset_flag = 0;
  const bool root_link = P.root_link_mode != 0;
  for (int sub = 0; sub < k; ++sub) {
    d = update_states(s);
    float a[AGX_MAX_ACTIONS];
#pragma unroll
    for (int c = 0; c < AGX_MAX_ACTIONS; ++c) a[c] = clamp_minmax(a_in[c], -10.0f, 10.0f);  // clip_actions
    if (P.controller == AGX_CTRL_NONE) {
#pragma unroll
      for (int j = 0; j < M; ++j) u[j] = motor_update(P, a[j], u[j], kT[j], tinc[j], tdec[j]);
    } else {
      wc = run_controller(P, s, d, a, g);
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
      bw[r] = acc;
    }
    // simulate_drag (base_multirotor.py:260-285), pre-physics body velocities
    {
      float vbn = norm(d.vbody);
      bw[0] += (-P.lin_drag_linear[0] * d.vbody.x) + (-P.lin_drag_quadratic[0] * vbn * d.vbody.x);
      bw[1] += (-P.lin_drag_linear[1] * d.vbody.y) + (-P.lin_drag_quadratic[1] * vbn * d.vbody.y);
      bw[2] += (-P.lin_drag_linear[2] * d.vbody.z) + (-P.lin_drag_quadratic[2] * vbn * d.vbody.z);
      bw[3] += (-P.ang_drag_linear[0] * d.wbody.x) + (-P.ang_drag_quadratic[0] * fabsf(d.wbody.x) * d.wbody.x);
      bw[4] += (-P.ang_drag_linear[1] * d.wbody.y) + (-P.ang_drag_quadratic[1] * fabsf(d.wbody.y) * d.wbody.y);
      bw[5] += (-P.ang_drag_linear[2] * d.wbody.z) + (-P.ang_drag_quadratic[2] * fabsf(d.wbody.z) * d.wbody.z);
    }
    if (B.disturb) {  // apply_disturbance (base_multirotor.py:213-234)
      const float *dd = B.disturb + (size_t)sub * 7 * n + i;
      float occ = dd[0];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float lo = -B.disturb_max[c], hi = B.disturb_max[c];
        bw[c] += ((hi - lo) * dd[(size_t)(1 + c) * n] + lo) * occ;
      }
    }
    integrate(P, s, V3{bw[0], bw[1], bw[2]}, V3{bw[3], bw[4], bw[5]});
    if (B.boxes) crashed = crashed || collide_boxes(B.boxes, B.num_boxes, n, i, s.p, P.collision_radius);
  }
  store_state(B.state, n, i, s);
  if (k > 0) {
    store_derived(B.derived, n, i, d);
#pragma unroll
    for (int j = 0; j < M; ++j) B.motor_thrust[j * n + i] = u[j];
    // RobotManagerIGE.pre_physics_step runs every sub-step: prev <- cur, cur <- action
#pragma unroll
    for (int c = 0; c < AGX_MAX_ACTIONS; ++c)
      if (c < A) {
        B.prev_actions[c * n + i] = (k >= 2) ? a_in[c] : a_old[c];
        B.actions[c * n + i] = a_in[c];
      }
    if (B.wrench_cmd) {
      B.wrench_cmd[0 * n + i] = wc.f.x; B.wrench_cmd[1 * n + i] = wc.f.y; B.wrench_cmd[2 * n + i] = wc.f.z;
      B.wrench_cmd[3 * n + i] = wc.t.x; B.wrench_cmd[4 * n + i] = wc.t.y; B.wrench_cmd[5 * n + i] = wc.t.z;
    }
  }
  B.crashes[i] = crashed ? 1 : 0;
  B.truncations[i] = 0;
  B.sim_steps[i] = B.sim_steps[i] + 1;
}

__global__ void __launch_bounds__(256) k_update_states(AgxEnvBuffers B, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  EnvState s = load_state(B.state, n, i);
  store_derived(B.derived, n, i, update_states(s));
}

__global__ void __launch_bounds__(256) k_controller_wrench(AgxRobotParams P, AgxEnvBuffers B, int n, const float *__restrict__ action) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  EnvState s = load_state(B.state, n, i);
  Derived d = load_derived(B.derived, n, i);
  Gains g = load_gains(B.gains, n, i);
  float a[AGX_MAX_ACTIONS];
#pragma unroll
  for (int c = 0; c < AGX_MAX_ACTIONS; ++c)
    a[c] = (c < P.num_actions) ? clamp_minmax(action[(size_t)i * P.num_actions + c], -10.0f, 10.0f) : 0.0f;
  Wrench wc = run_controller(P, s, d, a, g);
  B.wrench_cmd[0 * n + i] = wc.f.x; B.wrench_cmd[1 * n + i] = wc.f.y; B.wrench_cmd[2 * n + i] = wc.f.z;
  B.wrench_cmd[3 * n + i] = wc.t.x; B.wrench_cmd[4 * n + i] = wc.t.y; B.wrench_cmd[5 * n + i] = wc.t.z;
}

// ---------------------------------------------------------------------------------------
// Tasks
// ---------------------------------------------------------------------------------------

// position_setpoint_task.py:205-229, 245-282 + truncation (:172-174) + reset set (env_manager.py:364-371)
__global__ void __launch_bounds__(256) k_reward_position(AgxEnvBuffers B, int n, const float *__restrict__ target, int episode_len,
                                  int reset_on_collision, float *__restrict__ reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool reset = false;
  if (i < n) {
    EnvState s = load_state(B.state, n, i);
    Q4 qveh = Q4{B.derived[3 * n + i], B.derived[4 * n + i], B.derived[5 * n + i], B.derived[6 * n + i]};
    V3 wb = V3{B.derived[13 * n + i], B.derived[14 * n + i], B.derived[15 * n + i]};
    V3 tgt = V3{target[0 * n + i], target[1 * n + i], target[2 * n + i]};
    V3 pe = quat_apply(conj(qveh), tgt - s.p);  // quat_apply_inverse
    float dist = norm(pe);
    float pos_reward = 3.0f * expf(-8.0f * dist * dist) + 2.0f * expf(-4.0f * dist * dist);
    float dist_reward = (20.0f - dist) / 40.0f;
    V3 up = quat_rotate(s.q, V3{0.0f, 0.0f, 1.0f});  // quat_axis(q, 2)
    float tilt = fabsf(1.0f - up.z);
    float up_reward = 0.2f / (0.1f + tilt * tilt);
    float spin = norm(wb);
    float ang_reward = (1.0f / (1.0f + spin * spin)) * 3.0f;
    float total = pos_reward + dist_reward + pos_reward * (up_reward + ang_reward);
    total = 1.0f * total;
    bool crash = B.crashes[i] != 0;
    if (dist > 8.0f) crash = true;
    if (crash) total = -20.0f;
    reward[i] = total;
    B.crashes[i] = crash ? 1 : 0;
    bool trunc = B.sim_steps[i] > episode_len;
    B.truncations[i] = trunc ? 1 : 0;
    reset = (crash && reset_on_collision) || trunc;
    B.reset_mask[i] = reset ? 1 : 0;
  }
  if (__ballot(reset) != 0ull && (threadIdx.x & 63) == 0) atomicOr(B.reset_flag, 1);
}

// position_setpoint_task.py:194-203, obs [N][13] row-major (what the policy network consumes)
__global__ void __launch_bounds__(256) k_obs_position(AgxEnvBuffers B, int n, const float *__restrict__ target, float *__restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float *o = obs + (size_t)i * 13;
  o[0] = target[0 * n + i] - B.state[0 * n + i];
  o[1] = target[1 * n + i] - B.state[1 * n + i];
  o[2] = target[2 * n + i] - B.state[2 * n + i];
  o[3] = B.state[3 * n + i]; o[4] = B.state[4 * n + i]; o[5] = B.state[5 * n + i]; o[6] = B.state[6 * n + i];
  o[7] = B.derived[10 * n + i]; o[8] = B.derived[11 * n + i]; o[9] = B.derived[12 * n + i];
  o[10] = B.derived[13 * n + i]; o[11] = B.derived[14 * n + i]; o[12] = B.derived[15 * n + i];
}

AGX_DEV float exp_reward(float mag, float ex, float v) { return mag * expf(-(v * v) * ex); }
AGX_DEV float exp_penalty(float mag, float ex, float v) { return mag * (expf(-(v * v) * ex) - 1.0f); }

struct NavParams {
  float rp[18];
};

// navigation_task.py:416-521 (+ :305-309 truncation)
__global__ void __launch_bounds__(256) k_reward_navigation(AgxEnvBuffers B, int n, int A, const float *__restrict__ target, NavParams R,
                                    float cpf, float *__restrict__ pos_err, float *__restrict__ prev_pos_err,
                                    int episode_len, int reset_on_collision, float *__restrict__ reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool reset = false;
  if (i < n) {
    const float *rp = R.rp;
    float mult = 1.0f + 2.0f * cpf;
    V3 p = V3{B.state[0 * n + i], B.state[1 * n + i], B.state[2 * n + i]};
    Q4 qveh = Q4{B.derived[3 * n + i], B.derived[4 * n + i], B.derived[5 * n + i], B.derived[6 * n + i]};
    V3 tgt = V3{target[0 * n + i], target[1 * n + i], target[2 * n + i]};
    V3 ppe = V3{pos_err[0 * n + i], pos_err[1 * n + i], pos_err[2 * n + i]};
    prev_pos_err[0 * n + i] = ppe.x; prev_pos_err[1 * n + i] = ppe.y; prev_pos_err[2 * n + i] = ppe.z;
    V3 pe = quat_rotate_inverse(qveh, tgt - p);
    pos_err[0 * n + i] = pe.x; pos_err[1 * n + i] = pe.y; pos_err[2 * n + i] = pe.z;
    float dist = norm(pe), prev_dist = norm(ppe);
    float pos_reward = exp_reward(rp[0], rp[1], dist);
    float close_reward = exp_reward(rp[2], rp[3], dist);
    float closer = prev_dist - dist;
    float closer_reward = (closer > 0.0f) ? rp[4] * closer : 2.0f * rp[4] * closer;
    float dist_reward = (20.0f - dist) / 20.0f;
    float a0 = B.actions[0 * n + i], a2 = B.actions[2 * n + i], a3 = B.actions[3 * n + i];
    float dx = a0 - B.prev_actions[0 * n + i], dz = a2 - B.prev_actions[2 * n + i], dyaw = a3 - B.prev_actions[3 * n + i];
    float diff_pen = exp_penalty(rp[5], rp[6], dx) + exp_penalty(rp[7], rp[8], dz) + exp_penalty(rp[9], rp[10], dyaw);
    float abs_pen = cpf * exp_penalty(rp[11], rp[12], a0) + cpf * exp_penalty(rp[13], rp[14], a2) +
                    cpf * exp_penalty(rp[15], rp[16], a3);
    float total_pen = diff_pen + abs_pen;
    float r = mult * (pos_reward + close_reward + closer_reward + dist_reward) + total_pen;
    bool crash = B.crashes[i] != 0;
    if (crash) r = rp[17];
    reward[i] = r;
    bool trunc = B.sim_steps[i] > episode_len;
    B.truncations[i] = trunc ? 1 : 0;
    reset = (crash && reset_on_collision) || trunc;
    B.reset_mask[i] = reset ? 1 : 0;
  }
  if (__ballot(reset) != 0ull && (threadIdx.x & 63) == 0) atomicOr(B.reset_flag, 1);
}

// navigation_task.py:369-393; one wave per env so the depth min-pool is a coalesced sweep
__global__ void __launch_bounds__(256) k_obs_navigation(AgxEnvBuffers B, int n, const float *__restrict__ target,
                                                         const float *__restrict__ u_vec, const float *__restrict__ u_euler,
                                                         const float *__restrict__ pixels, int ns, int H, int W, int gh, int gw,
                                                         int obs_dim, float *__restrict__ obs) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  float *o = obs + (size_t)i * obs_dim;
  if (lane == 0) {
    V3 p = V3{B.state[0 * n + i], B.state[1 * n + i], B.state[2 * n + i]};
    Q4 qveh = Q4{B.derived[3 * n + i], B.derived[4 * n + i], B.derived[5 * n + i], B.derived[6 * n + i]};
    V3 tgt = V3{target[0 * n + i], target[1 * n + i], target[2 * n + i]};
    V3 v = quat_rotate_inverse(qveh, tgt - p);
    // 0.1 * 2 * rand_like(vec - 0.5): the -0.5 sits inside rand_like in the reference (:374)
    V3 pv = V3{v.x + 0.1f * 2.0f * u_vec[(size_t)i * 3], v.y + 0.1f * 2.0f * u_vec[(size_t)i * 3 + 1],
               v.z + 0.1f * 2.0f * u_vec[(size_t)i * 3 + 2]};
    float dist = norm(v);
    o[0] = pv.x / dist; o[1] = pv.y / dist; o[2] = pv.z / dist; o[3] = dist;
    float e0 = ssa(B.derived[0 * n + i]), e1 = ssa(B.derived[1 * n + i]);
    o[4] = e0 + 0.1f * (u_euler[(size_t)i * 3] - 0.5f);
    o[5] = e1 + 0.1f * (u_euler[(size_t)i * 3 + 1] - 0.5f);
    o[6] = 0.0f;
    o[7] = B.derived[10 * n + i]; o[8] = B.derived[11 * n + i]; o[9] = B.derived[12 * n + i];
    o[10] = B.derived[13 * n + i]; o[11] = B.derived[14 * n + i]; o[12] = B.derived[15 * n + i];
    o[13] = B.actions[0 * n + i]; o[14] = B.actions[1 * n + i]; o[15] = B.actions[2 * n + i]; o[16] = B.actions[3 * n + i];
  }
  if (pixels) {
    const float *img = pixels + (size_t)i * ns * H * W;  // sensor 0
    const int ch = (H + gh - 1) / gh, cw = (W + gw - 1) / gw;
    for (int cell = lane; cell < gh * gw; cell += 64) {
      int cy = cell / gw, cx = cell % gw;
      float m = INFINITY;
      for (int y = cy * ch; y < min((cy + 1) * ch, H); ++y)
        for (int x = cx * cw; x < min((cx + 1) * cw, W); ++x) m = fminf(m, img[(size_t)y * W + x]);
      if (17 + cell < obs_dim) o[17 + cell] = m;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Reset (base_multirotor.py:177-205, motor_model.py:140-154, env_manager.py:301)
// ---------------------------------------------------------------------------------------
template <int M>
__global__ void __launch_bounds__(256) k_reset_masked(AgxRobotParams P, AgxEnvBuffers B, int n, AgxResetArgs R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (*B.reset_flag == 0) return;  // nobody resets: the reference does not touch anything
  bool reset = B.reset_mask[i] != 0;
  EnvState s;
  if (reset) {
    // IsaacGymEnv.reset_idx: env bounds first, the robot spawn uses them
    float bmin[3], bmax[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bmin[c] = (R.lower_bound_max[c] - R.lower_bound_min[c]) * R.u_bounds_lo[(size_t)i * 3 + c] + R.lower_bound_min[c];
      bmax[c] = (R.upper_bound_max[c] - R.upper_bound_min[c]) * R.u_bounds_hi[(size_t)i * 3 + c] + R.upper_bound_min[c];
      B.bounds_min[c * n + i] = bmin[c];
      B.bounds_max[c * n + i] = bmax[c];
    }
    float r[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) r[c] = (R.max_state[c] - R.min_state[c]) * R.u_state[(size_t)i * 13 + c] + R.min_state[c];
    s.p = V3{bmin[0] + (bmax[0] - bmin[0]) * r[0], bmin[1] + (bmax[1] - bmin[1]) * r[1], bmin[2] + (bmax[2] - bmin[2]) * r[2]};
    s.q = quat_from_euler(r[3], r[4], r[5]);
    s.v = V3{r[7], r[8], r[9]};
    s.w = V3{r[10], r[11], r[12]};
    store_state(B.state, n, i, s);
    if (R.u_gains) {
#pragma unroll
      for (int c = 0; c < 12; ++c)
        B.gains[c * n + i] = (R.gains_max[c] - R.gains_min[c]) * R.u_gains[(size_t)i * 12 + c] + R.gains_min[c];
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      size_t k = (size_t)i * M + j;
      B.motor_tau_inc[j * n + i] = (R.tau_inc_max - R.tau_inc_min) * R.u_tau_inc[k] + R.tau_inc_min;
      B.motor_tau_dec[j * n + i] = (R.tau_dec_max - R.tau_dec_min) * R.u_tau_dec[k] + R.tau_dec_min;
      B.motor_thrust[j * n + i] = (P.max_thrust - P.min_thrust) * R.u_thrust[k] + P.min_thrust;
      if (P.use_rps) B.motor_kT[j * n + i] = (R.kT_max - R.kT_min) * R.u_kT[k] + R.kT_min;
    }
    B.sim_steps[i] = 0;
  } else {
    s = load_state(B.state, n, i);
  }
  // BaseMultirotor.reset_idx ends with an un-indexed update_states(): every env is refreshed
  store_derived(B.derived, n, i, update_states(s));
}

}  // namespace agx

// =========================================================================================
// C ABI
// =========================================================================================
using namespace agx;

#define AGX_DISPATCH_M(M_, ...)                          \
  switch (M_) {                                          \
    case 1: { constexpr int kM = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int kM = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int kM = 3; __VA_ARGS__; } break; \
    case 4: { constexpr int kM = 4; __VA_ARGS__; } break; \
    case 5: { constexpr int kM = 5; __VA_ARGS__; } break; \
    case 6: { constexpr int kM = 6; __VA_ARGS__; } break; \
    case 7: { constexpr int kM = 7; __VA_ARGS__; } break; \
    case 8: { constexpr int kM = 8; __VA_ARGS__; } break; \
    default: return fail(AGX_E_ARG, "num_motors %d not in 1..8", M_); \
  }

static int check_common(const AgxRobotParams *P, const AgxEnvBuffers *B, int n) {
  AGX_REQUIRE(B != nullptr, "null buffers");
  AGX_REQUIRE(n > 0, "num_envs must be > 0 (got %d)", n);
  if (P) {
    AGX_REQUIRE(P->num_motors >= 1 && P->num_motors <= AGX_MAX_MOTORS, "num_motors out of range");
    AGX_REQUIRE(P->num_actions >= 1 && P->num_actions <= AGX_MAX_ACTIONS, "num_actions out of range");
    AGX_REQUIRE(P->controller >= 0 && P->controller <= AGX_CTRL_FULLY_ACTUATED, "unknown controller id %d", P->controller);
    AGX_REQUIRE(P->controller != AGX_CTRL_FULLY_ACTUATED || P->num_actions == 7, "fully actuated controller needs 7 actions");
    AGX_REQUIRE(P->controller != AGX_CTRL_NONE || P->num_actions == P->num_motors, "no_control needs num_actions == num_motors");
    AGX_REQUIRE(P->controller == AGX_CTRL_NONE || P->controller == AGX_CTRL_FULLY_ACTUATED || P->num_actions == 4,
                "Lee controllers take 4 actions");
  }
  return AGX_OK;
}

extern "C" int agx_dynamics_substeps(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const float *actions_in,
                                     int k, void *stream) {
  if (int e = check_common(P, B, n)) return e;
  AGX_REQUIRE(P != nullptr, "null params");
  AGX_REQUIRE(actions_in != nullptr, "null actions");
  AGX_REQUIRE(k >= 0 && k <= AGX_MAX_SUBSTEPS, "k_substeps out of range: %d", k);
  AGX_REQUIRE(B->state && B->derived && B->actions && B->prev_actions && B->motor_thrust && B->motor_tau_inc &&
                  B->motor_tau_dec && B->crashes && B->truncations && B->sim_steps && B->reset_flag,
              "null env buffer");
  AGX_REQUIRE(P->controller == AGX_CTRL_NONE || B->gains, "null gains");
  AGX_REQUIRE(!P->use_rps || B->motor_kT, "null motor_kT with use_rps");
  const int block = pick_block(n);
  AGX_DISPATCH_M(P->num_motors, hipLaunchKernelGGL(k_dynamics_substeps<kM>, dim3(blocks_for(n, block)), dim3(block), 0,
                                                   (hipStream_t)stream, *P, *B, n, actions_in, k));
  return check_launch("agx_dynamics_substeps");
}

extern "C" int agx_update_states(const AgxEnvBuffers *B, int n, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(B->state && B->derived, "null env buffer");
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_update_states, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *B, n);
  return check_launch("agx_update_states");
}

extern "C" int agx_controller_wrench(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const float *action,
                                     void *stream) {
  if (int e = check_common(P, B, n)) return e;
  AGX_REQUIRE(P && P->controller != AGX_CTRL_NONE, "controller required");
  AGX_REQUIRE(action && B->state && B->derived && B->gains && B->wrench_cmd, "null buffer");
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_controller_wrench, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *P, *B, n, action);
  return check_launch("agx_controller_wrench");
}

extern "C" int agx_reward_position(const AgxEnvBuffers *B, int n, const float *target, int episode_len,
                                   int reset_on_collision, float *reward, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(target && reward && B->reset_flag && B->reset_mask && B->state && B->derived && B->crashes && B->truncations && B->sim_steps,
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
  hipLaunchKernelGGL(k_reward_navigation, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *B, n, 4, target,
                     R, cpf, pos_err, prev_pos_err, episode_len, reset_on_collision, reward);
  return check_launch("agx_reward_navigation");
}

extern "C" int agx_obs_navigation(const AgxEnvBuffers *B, int n, const float *target, const float *u_vec,
                                  const float *u_euler, const float *pixels, int ns, int H, int W, int gh, int gw,
                                  int obs_dim, float *obs, void *stream) {
  if (int e = check_common(nullptr, B, n)) return e;
  AGX_REQUIRE(target && u_vec && u_euler && obs && B->state && B->derived && B->actions, "null buffer");
  AGX_REQUIRE(obs_dim >= 17, "obs_dim must be >= 17");
  AGX_REQUIRE(!pixels || (ns > 0 && H > 0 && W > 0 && gh > 0 && gw > 0), "bad image sizes");
  hipLaunchKernelGGL(k_obs_navigation, dim3(blocks_for(n, 4)), dim3(256), 0, (hipStream_t)stream, *B, n, target, u_vec,
                     u_euler, pixels, ns, H, W, gh, gw, obs_dim, obs);
  return check_launch("agx_obs_navigation");
}

extern "C" int agx_reset_masked(const AgxRobotParams *P, const AgxEnvBuffers *B, int n, const AgxResetArgs *R,
                                void *stream) {
  if (int e = check_common(P, B, n)) return e;
  AGX_REQUIRE(P && R && B->reset_flag && B->reset_mask, "null argument");
  AGX_REQUIRE(R->u_bounds_lo && R->u_bounds_hi && R->u_state && R->u_tau_inc && R->u_tau_dec && R->u_thrust &&
                  B->bounds_min && B->bounds_max,
              "null reset input");
  AGX_REQUIRE(!P->use_rps || R->u_kT, "null u_kT with use_rps");
  AGX_REQUIRE(!R->u_gains || B->gains, "null gains with u_gains");
  const int block = pick_block(n);
  AGX_DISPATCH_M(P->num_motors, hipLaunchKernelGGL(k_reset_masked<kM>, dim3(blocks_for(n, block)), dim3(block), 0,
                                                   (hipStream_t)stream, *P, *B, n, *R));
  return check_launch("agx_reset_masked");
}
