/*
 * TEST INFRASTRUCTURE -- CPU oracle, dynamics half.
 *
 * A restatement, in scalar C, of what the reference computes per env and per
 * physics sub-step.  Everything that exists as source in the reference
 * (controllers, allocation, motor model, drag, reward, obs) follows the cited
 * torch code line by line and is PINNED against golden vectors produced by the
 * reference's own code (tests/golden/, oracle/gen_golden.py).
 *
 * The rigid-body integrator (orc_integrate) and the collision flag
 * (orc_collide_sphere_boxes) replace the closed-source Isaac Gym / PhysX binary
 * (IGE_env_manager.py:444-449,477,486-495).  No reference source or test pins
 * their arithmetic: ** parity unpinned ** for those two functions.  The form
 * chosen is PhysX's published rigid-body update (PhysX 4.1, open source:
 * DyBodyCoreIntegrator.h bodyCoreComputeUnconstrainedVelocity / integrateCore)
 * plus the gyroscopic term of the articulation solver; see DESIGN.md.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no -ffast-math).
 */
#include "oracle_types.h"
#include "oracle_math.h" /* the elementary functions the HIP kernels evaluate too, bit for bit */

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI_F 3.14159274101257324f /* float(torch.pi) */
#define ORC_2PI_F 6.28318548202514648f

/* ------------------------------------------------------------------ */
/* utils/math.py restatements (quaternions are xyzw)                   */
/* ------------------------------------------------------------------ */

/* python-style modulo for a positive divisor (torch `%` == remainder). utils/math.py:144-152 */
static float py_mod(float a, float m) {
  float r = fmodf(a, m);
  if (r != 0.0f && ((r < 0.0f) != (m < 0.0f))) r += m;
  return r;
}

/* utils/math.py:150-152 */
static float ssa(float a) { return py_mod(a + ORC_PI_F, ORC_2PI_F) - ORC_PI_F; }

/* torch.cross: ATen's kernel evaluates a_j b_k - a_k b_j inside ONE compiled expression, which the compiler contracts
 * into a fused multiply-subtract -- fma(a_j, b_k, -(a_k b_j)): the second product is rounded, the first is not (the
 * same form on the CPU, where the goldens are made -- checked on 10^5 random vectors, 100 % bit-equal -- and in nvcc's
 * default fmad mode).  Separate torch ops (a * b + c written as two calls) round separately, and that is how every
 * other expression of this file is written. */
static void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
  o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
  o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

/* torch.norm(x, dim) of a 3-vector: the reduction kernel accumulates acc = fma(x_k, x_k, acc) starting from x_0^2
 * (bit-equal on 10^5 random vectors); 4-vectors take the kernel's unrolled path, ((x0^2 + x1^2) + x2^2) + x3^2 with
 * every product rounded (utils/math.py:296-298 `normalize`). */
static float norm3(const float x[3]) { return sqrtf(fmaf(x[2], x[2], fmaf(x[1], x[1], x[0] * x[0]))); }

static float dot3(const float a[3], const float b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* utils/math.py:329-336: a + b + c with a = v(2w^2-1), b = 2w(q x v), c = 2q(q.v) */
static void quat_rotate(const float q[4], const float v[3], float o[3]) {
  float w = q[3];
  float s = 2.0f * (w * w) - 1.0f;
  float cr[3];
  cross3(q, v, cr);
  float d = dot3(q, v);
  for (int i = 0; i < 3; ++i) {
    float a = v[i] * s;
    float b = cr[i] * w * 2.0f;
    float c = q[i] * d * 2.0f;
    o[i] = a + b + c;
  }
}

/* utils/math.py:340-347: a - b + c */
static void quat_rotate_inverse(const float q[4], const float v[3], float o[3]) {
  float w = q[3];
  float s = 2.0f * (w * w) - 1.0f;
  float cr[3];
  cross3(q, v, cr);
  float d = dot3(q, v);
  for (int i = 0; i < 3; ++i) {
    float a = v[i] * s;
    float b = cr[i] * w * 2.0f;
    float c = q[i] * d * 2.0f;
    o[i] = a - b + c;
  }
}

/* utils/math.py:314-320: t = 2 (q x v); v + w t + q x t */
static void quat_apply(const float q[4], const float v[3], float o[3]) {
  float t[3], u[3];
  cross3(q, v, t);
  t[0] *= 2.0f; t[1] *= 2.0f; t[2] *= 2.0f;
  cross3(q, t, u);
  for (int i = 0; i < 3; ++i) o[i] = v[i] + q[3] * t[i] + u[i];
}

static void quat_conj(const float q[4], float o[4]) {
  o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3];
}

/* utils/math.py:243-263 (Isaac Gym 9-product form) */
static void quat_mul(const float a[4], const float b[4], float o[4]) {
  float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
  float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
  float ww = (z1 + x1) * (x2 + y2);
  float yy = (w1 - y1) * (w2 + z2);
  float zz = (w1 + y1) * (w2 - z2);
  float xx = ww + yy + zz;
  float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[3] = qq - ww + (z1 - y1) * (y2 - z2);
  o[0] = qq - xx + (x1 + w1) * (x2 + w2);
  o[1] = qq - yy + (w1 - x1) * (y2 + z2);
  o[2] = qq - zz + (z1 + y1) * (w2 - x2);
}

/* utils/math.py:267-293, row-major 3x3 */
static void quat_to_rotmat(const float q[4], float m[9]) {
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float xx = x * x, xy = x * y, xz = x * z, xw = x * w;
  float yy = y * y, yz = y * z, yw = y * w;
  float zz = z * z, zw = z * w;
  m[0] = 1.0f - 2.0f * (yy + zz);
  m[1] = 2.0f * (xy - zw);
  m[2] = 2.0f * (xz + yw);
  m[3] = 2.0f * (xy + zw);
  m[4] = 1.0f - 2.0f * (xx + zz);
  m[5] = 2.0f * (yz - xw);
  m[6] = 2.0f * (xz - yw);
  m[7] = 2.0f * (yz + xw);
  m[8] = 1.0f - 2.0f * (xx + yy);
}

/* utils/math.py:156-172 */
static void quat_from_euler_xyz(float roll, float pitch, float yaw, float q[4]) {
  float cy, sy, cr, sr, cp, sp;
  om_sincosf(yaw * 0.5f, &sy, &cy);
  om_sincosf(roll * 0.5f, &sr, &cr);
  om_sincosf(pitch * 0.5f, &sp, &cp);
  q[3] = cy * cr * cp + sy * sr * sp;
  q[0] = cy * sr * cp - sy * cr * sp;
  q[1] = cy * cr * sp + sy * sr * cp;
  q[2] = sy * cr * cp - cy * sr * sp;
}

/* utils/math.py:124-146: angles in [0, 2pi) */
static void get_euler_xyz(const float q[4], float e[3]) {
  float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
  float sinr_cosp = 2.0f * (qw * qx + qy * qz);
  float cosr_cosp = qw * qw - qx * qx - qy * qy + qz * qz;
  float roll = om_atan2f(sinr_cosp, cosr_cosp);
  float sinp = 2.0f * (qw * qy - qz * qx);
  float pitch;
  if (fabsf(sinp) >= 1.0f) {
    /* copysign(pi/2, sinp) = |pi/2| * sign(sinp)  (utils/math.py:93-96) */
    float sg = (sinp > 0.0f) ? 1.0f : ((sinp < 0.0f) ? -1.0f : 0.0f);
    pitch = (ORC_PI_F / 2.0f) * sg;
  } else {
    pitch = om_asinf(sinp);
  }
  float siny_cosp = 2.0f * (qw * qz + qx * qy);
  float cosy_cosp = qw * qw + qx * qx - qy * qy - qz * qz;
  float yaw = om_atan2f(siny_cosp, cosy_cosp);
  e[0] = py_mod(roll, ORC_2PI_F);
  e[1] = py_mod(pitch, ORC_2PI_F);
  e[2] = py_mod(yaw, ORC_2PI_F);
}

/* pytorch3d.transforms.matrix_to_quaternion (argmax branch, no sign
 * standardisation) + the wxyz->xyzw reorder of base_lee_controller.py:188-189.
 * m is row-major.  Third-party, un-pinned in the reference (setup.py:17):
 * restated from the published algorithm.                                   */
static void rotmat_to_quat_xyzw(const float m[9], float q[4]) {
  float m00 = m[0], m01 = m[1], m02 = m[2];
  float m10 = m[3], m11 = m[4], m12 = m[5];
  float m20 = m[6], m21 = m[7], m22 = m[8];
  float t[4];
  t[0] = 1.0f + m00 + m11 + m22;
  t[1] = 1.0f + m00 - m11 - m22;
  t[2] = 1.0f - m00 + m11 - m22;
  t[3] = 1.0f - m00 - m11 + m22;
  float qa[4];
  for (int i = 0; i < 4; ++i) qa[i] = (t[i] > 0.0f) ? sqrtf(t[i]) : 0.0f;
  int best = 0; /* torch.argmax returns the first maximal index */
  for (int i = 1; i < 4; ++i)
    if (qa[i] > qa[best]) best = i;
  float c[4]; /* candidate in (r, i, j, k) order */
  switch (best) {
    case 0: c[0] = qa[0] * qa[0]; c[1] = m21 - m12; c[2] = m02 - m20; c[3] = m10 - m01; break;
    case 1: c[0] = m21 - m12; c[1] = qa[1] * qa[1]; c[2] = m10 + m01; c[3] = m02 + m20; break;
    case 2: c[0] = m02 - m20; c[1] = m10 + m01; c[2] = qa[2] * qa[2]; c[3] = m12 + m21; break;
    default: c[0] = m10 - m01; c[1] = m20 + m02; c[2] = m21 + m12; c[3] = qa[3] * qa[3]; break;
  }
  float den = 2.0f * ((qa[best] > 0.1f) ? qa[best] : 0.1f);
  q[3] = c[0] / den; /* w */
  q[0] = c[1] / den;
  q[1] = c[2] / den;
  q[2] = c[3] / den;
}

/* test access to the restated third-party routine (tests/test_oracle_physics_kat.py pins it against an
 * independent implementation, scipy.spatial.transform): m [n][9] row-major -> q [n][4] xyzw               */
void orc_rotmat_to_quat(int n, const float *m, float *q) {
  for (int i = 0; i < n; ++i) rotmat_to_quat_xyzw(m + 9 * i, q + 4 * i);
}

/* ------------------------------------------------------------------ */
/* a1: BaseMultirotor.update_states, base_multirotor.py:287-294         */
/* ------------------------------------------------------------------ */
void orc_update_states(int n, const float *state, float *euler, float *qveh, float *vveh,
                       float *vbody, float *wbody) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float *s = state + 13 * i;
    const float *q = s + 3, *v = s + 7, *w = s + 10;
    float e[3];
    get_euler_xyz(q, e);
    for (int k = 0; k < 3; ++k) euler[3 * i + k] = ssa(e[k]);
    /* vehicle_frame_quat_from_quat: euler * [0,0,1] -> quat (utils/math.py:176-180) */
    quat_from_euler_xyz(e[0] * 0.0f, e[1] * 0.0f, e[2] * 1.0f, qveh + 4 * i);
    quat_rotate_inverse(qveh + 4 * i, v, vveh + 3 * i);
    quat_rotate_inverse(q, v, vbody + 3 * i);
    quat_rotate_inverse(q, w, wbody + 3 * i);
  }
}

/* ------------------------------------------------------------------ */
/* a3-a10: Lee controllers                                             */
/* ------------------------------------------------------------------ */

/* base_lee_controller.py:120-134 */
static void compute_acceleration(const float p[3], const float v[3], const float qveh[4],
                                 const float sp_pos[3], const float sp_vel[3], const float Kp[3],
                                 const float Kv[3], float acc[3]) {
  float sp_vel_w[3];
  quat_rotate(qveh, sp_vel, sp_vel_w);
  for (int k = 0; k < 3; ++k) {
    float pe = sp_pos[k] - p[k];
    float ve = sp_vel_w[k] - v[k];
    acc[k] = Kp[k] * pe + Kv[k] * ve;
  }
}

/* base_lee_controller.py:136-154.  sp_angvel[2] is clamped IN PLACE. */
static void compute_body_torque(const OrcRobotParams *P, const float q[4], const float wb[3],
                                const float q_des[4], float sp_angvel[3], const float KR[3],
                                const float Kw[3], float tau[3]) {
  if (sp_angvel[2] < -P->max_yaw_rate) sp_angvel[2] = -P->max_yaw_rate;
  if (sp_angvel[2] > P->max_yaw_rate) sp_angvel[2] = P->max_yaw_rate;
  float qi[4], qe[4], R[9];
  quat_conj(q, qi);
  quat_mul(qi, q_des, qe);
  quat_to_rotmat(qe, R);
  /* 0.5 * vee(R^T - R), utils/math.py:35-42 */
  float rot_err[3];
  rot_err[0] = 0.5f * (-(R[7] - R[5]));
  rot_err[1] = 0.5f * (R[6] - R[2]);
  rot_err[2] = 0.5f * (-(R[3] - R[1]));
  float wsp_b[3];
  quat_rotate(qe, sp_angvel, wsp_b);
  float Jw[3];
  for (int r = 0; r < 3; ++r)
    Jw[r] = P->inertia[3 * r] * wb[0] + P->inertia[3 * r + 1] * wb[1] + P->inertia[3 * r + 2] * wb[2];
  float ff[3];
  cross3(wb, Jw, ff);
  for (int k = 0; k < 3; ++k) {
    float we = wb[k] - wsp_b[k];
    tau[k] = -KR[k] * rot_err[k] - Kw[k] * we + ff[k];
  }
}

/* base_lee_controller.py:173-194 */
static void desired_orientation_pos_vel(const float f[3], float yaw, float qd[4]) {
  float nf = norm3(f);
  float b3[3] = {f[0] / nf, f[1] / nf, f[2] / nf};
  float sy_, cy_;
  om_sincosf(yaw, &sy_, &cy_);
  float tmp[3] = {cy_, sy_, 0.0f};
  float b2[3], b1[3];
  cross3(b3, tmp, b2);
  float n2 = norm3(b2);
  b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
  cross3(b2, b3, b1);
  float R[9] = {b1[0], b2[0], b3[0], b1[1], b2[1], b3[1], b1[2], b2[2], b3[2]};
  rotmat_to_quat_xyzw(R, qd);
}

/* base_lee_controller.py:158-169 */
static void desired_orientation_forces_yaw(const float f[3], float yaw, float qd[4]) {
  float c_phi_s_theta = f[0];
  float s_phi = -f[1];
  float c_phi_c_theta = f[2];
  float pitch = om_atan2f(c_phi_s_theta, c_phi_c_theta);
  float roll = om_atan2f(s_phi, sqrtf(c_phi_c_theta * c_phi_c_theta + c_phi_s_theta * c_phi_s_theta));
  quat_from_euler_xyz(roll, pitch, yaw, qd);
}

/* base_lee_controller.py:201-215 with euler rates (0, 0, yaw_rate) */
static void euler_rates_to_body_rates(const float euler[3], const float rates[3], float out[3]) {
  float s_pitch, c_pitch, s_roll, c_roll;
  om_sincosf(euler[1], &s_pitch, &c_pitch);
  om_sincosf(euler[0], &s_roll, &c_roll);
  /* rows of the matrix; entries the reference leaves untouched only ever
     multiply the zero roll/pitch rates            */
  out[0] = 1.0f * rates[0] + 0.0f * rates[1] + (-s_pitch) * rates[2];
  out[1] = 0.0f * rates[0] + c_roll * rates[1] + (s_roll * c_pitch) * rates[2];
  out[2] = 0.0f * rates[0] + (-s_roll) * rates[1] + (c_roll * c_pitch) * rates[2];
}

/* One env's controller.  `a` is the (already +-10 clipped) action row and may
 * be mutated exactly where the reference mutates it (rates: yaw-rate clamp;
 * fully-actuated: quaternion normalisation).  wrench = [fx fy fz tx ty tz]. */
static void controller_one(const OrcRobotParams *P, const float *s, const float euler[3],
                           const float qveh[4], const float wb[3], float *a, const float Kp[3],
                           const float Kv[3], const float KR[3], const float Kw[3],
                           float wrench[6]) {
  const float *p = s, *q = s + 3, *v = s + 7;
  const float *g = P->gravity;
  float m = P->mass;
  for (int k = 0; k < 6; ++k) wrench[k] = 0.0f;
  float R[9];
  float zero3[3] = {0.0f, 0.0f, 0.0f};
  switch (P->controller) {
    case ORC_CTRL_POSITION: { /* position_control.py:20-51 */
      float acc[3], f[3], qd[4], wsp[3] = {0, 0, 0};
      compute_acceleration(p, v, qveh, a, zero3, Kp, Kv, acc);
      for (int k = 0; k < 3; ++k) f[k] = (acc[k] - g[k]) * m;
      quat_to_rotmat(q, R);
      wrench[2] = f[0] * R[2] + f[1] * R[5] + f[2] * R[8];
      desired_orientation_pos_vel(f, a[3], qd);
      compute_body_torque(P, q, wb, qd, wsp, KR, Kw, wrench + 3);
    } break;
    case ORC_CTRL_VELOCITY: { /* velocity_control.py:18-51 */
      float acc[3], f[3], qd[4], rates[3] = {0, 0, a[3]}, wsp[3];
      compute_acceleration(p, v, qveh, p, a, Kp, Kv, acc);
      for (int k = 0; k < 3; ++k) f[k] = (acc[k] - g[k]) * m;
      quat_to_rotmat(q, R);
      wrench[2] = f[0] * R[2] + f[1] * R[5] + f[2] * R[8];
      desired_orientation_pos_vel(f, euler[2], qd);
      euler_rates_to_body_rates(euler, rates, wsp);
      compute_body_torque(P, q, wb, qd, wsp, KR, Kw, wrench + 3);
    } break;
    case ORC_CTRL_ATTITUDE: { /* attitude_control.py:16-43 */
      float gn = norm3(g);
      wrench[2] = (a[0] + 1.0f) * m * gn;
      float rates[3] = {0, 0, a[3]}, wsp[3], qd[4];
      euler_rates_to_body_rates(euler, rates, wsp);
      quat_from_euler_xyz(a[1], a[2], euler[2], qd);
      compute_body_torque(P, q, wb, qd, wsp, KR, Kw, wrench + 3);
    } break;
    case ORC_CTRL_RATES: { /* rates_control.py:16-30; line 25's [N]-[N,3] broadcast bug is
                              restated as the z component                     */
      wrench[2] = (a[0] - g[2]) * m;
      compute_body_torque(P, q, wb, q, a + 1, KR, Kw, wrench + 3);
    } break;
    case ORC_CTRL_ACCELERATION: { /* acceleration_control.py:16-45 */
      float f[3], qd[4], rates[3] = {0, 0, a[3]}, wsp[3];
      for (int k = 0; k < 3; ++k) f[k] = m * (a[k] - g[k]);
      quat_to_rotmat(q, R);
      wrench[2] = f[0] * R[2] + f[1] * R[5] + f[2] * R[8];
      desired_orientation_forces_yaw(f, euler[2], qd);
      euler_rates_to_body_rates(euler, rates, wsp);
      compute_body_torque(P, q, wb, qd, wsp, KR, Kw, wrench + 3);
    } break;
    case ORC_CTRL_VEL_STEERING: { /* velocity_steeing_angle_controller.py:15-45 */
      float acc[3], f[3], qd[4], rates[3] = {0, 0, 0}, wsp[3];
      compute_acceleration(p, v, qveh, p, a, Kp, Kv, acc);
      for (int k = 0; k < 3; ++k) f[k] = (acc[k] - g[k]) * m;
      quat_to_rotmat(q, R);
      wrench[2] = f[0] * R[2] + f[1] * R[5] + f[2] * R[8];
      desired_orientation_pos_vel(f, a[3], qd);
      euler_rates_to_body_rates(euler, rates, wsp);
      compute_body_torque(P, q, wb, qd, wsp, KR, Kw, wrench + 3);
    } break;
    case ORC_CTRL_FULLY_ACTUATED: { /* fully_actuated_control.py:14-32 */
      float nq = sqrtf(a[3] * a[3] + a[4] * a[4] + a[5] * a[5] + a[6] * a[6]);
      if (nq < 1e-9f) nq = 1e-9f; /* utils/math.py:296-298 */
      for (int k = 3; k < 7; ++k) a[k] = a[k] / nq;
      float acc[3], f[3], wsp[3] = {0, 0, 0};
      compute_acceleration(p, v, qveh, a, zero3, Kp, Kv, acc);
      for (int k = 0; k < 3; ++k) f[k] = m * (acc[k] - g[k]);
      quat_rotate_inverse(q, f, wrench);
      compute_body_torque(P, q, wb, a + 3, wsp, KR, Kw, wrench + 3);
    } break;
    default: break;
  }
}

/* ------------------------------------------------------------------ */
/* a12: MotorModel.update_motor_thrusts, control/motor_model.py:88-250  */
/* ------------------------------------------------------------------ */
static float clampf(float x, float lo, float hi) {
  /* tensor_clamp = max(min(t, max_t), min_t), utils/math.py:219-221 */
  float y = (x < hi) ? x : hi;
  return (y > lo) ? y : lo;
}

static float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

static float motor_rate(float err, float mix, float max_rate) {
  return clampf(mix * err, -max_rate, max_rate);
}

/* motor_model.py:166-198 */
static float rk4_delta(float ref, float cur, float mix, float max_rate, float dt, float dt_over_6) {
  float k1 = motor_rate(ref - cur, mix, max_rate);
  float k2 = motor_rate(ref - (cur + 0.5f * dt * k1), mix, max_rate);
  float k3 = motor_rate(ref - (cur + 0.5f * dt * k2), mix, max_rate);
  float k4 = motor_rate(ref - (cur + dt * k3), mix, max_rate);
  return dt_over_6 * (k1 + 2.0f * k2 + 2.0f * k3 + k4);
}

static float motor_update_one(const OrcRobotParams *P, float ref, float cur, float kT,
                              float tau_inc, float tau_dec) {
  float dt = P->dt;
  ref = (ref < P->min_thrust) ? P->min_thrust : ((ref > P->max_thrust) ? P->max_thrust : ref);
  float err = ref - cur;
  float tc = (sgn(cur) * sgn(err) < 0.0f) ? tau_dec : tau_inc;
  float mix = P->use_discrete_approximation ? 1.0f / (dt + tc) : 1.0f / tc;
  if (P->use_rps) {
    float cur_rpm = sqrtf(cur / kT);
    float des_rpm = sqrtf(ref / kT);
    if (P->integration_rk4)
      cur_rpm += rk4_delta(des_rpm, cur_rpm, mix, P->max_rate, dt, P->dt_over_6);
    else
      cur_rpm += motor_rate(des_rpm - cur_rpm, mix, P->max_rate) * dt;
    return kT * (cur_rpm * cur_rpm);
  }
  if (P->integration_rk4) return cur + rk4_delta(ref, cur, mix, P->max_rate, dt, P->dt_over_6);
  return cur + motor_rate(err, mix, P->max_rate) * dt;
}

/* ------------------------------------------------------------------ */
/* a15: rigid-body integration -- REPLACES PhysX, ** parity unpinned ** */
/* ------------------------------------------------------------------ */
static void integrate_one(const OrcRobotParams *P, float *s, const float Fb[3], const float Tb[3]) {
  float *p = s, *q = s + 3, *v = s + 7, *w = s + 10;
  float dt = P->dt;
  /* LOCAL_SPACE wrench -> world (IGE_env_manager.py:444-449) */
  float Fw[3];
  quat_rotate(q, Fb, Fw);
  /* angular: Euler's equation in the body frame (articulation solver keeps the
     gyroscopic term), J constant in body frame                               */
  float wb[3];
  quat_rotate_inverse(q, w, wb);
  float Jw[3], gyro[3], rhs[3], dwb[3], wb_new[3], w_new[3];
  for (int r = 0; r < 3; ++r)
    Jw[r] = P->inertia[3 * r] * wb[0] + P->inertia[3 * r + 1] * wb[1] + P->inertia[3 * r + 2] * wb[2];
  cross3(wb, Jw, gyro);
  for (int k = 0; k < 3; ++k) rhs[k] = Tb[k] - gyro[k];
  for (int r = 0; r < 3; ++r)
    dwb[r] = P->inertia_inv[3 * r] * rhs[0] + P->inertia_inv[3 * r + 1] * rhs[1] +
             P->inertia_inv[3 * r + 2] * rhs[2];
  for (int k = 0; k < 3; ++k) wb_new[k] = wb[k] + dt * dwb[k];
  quat_rotate(q, wb_new, w_new);
  /* linear: external force then gravity (bodyCoreComputeUnconstrainedVelocity) */
  float v_new[3];
  for (int k = 0; k < 3; ++k) {
    v_new[k] = v[k] + dt * (Fw[k] / P->mass);
    v_new[k] = v_new[k] + P->gravity[k] * dt;
  }
  /* damping: v *= max(0, 1 - c dt) */
  float ml = 1.0f - P->linear_damping * dt, ma = 1.0f - P->angular_damping * dt;
  if (ml < 0.0f) ml = 0.0f;
  if (ma < 0.0f) ma = 0.0f;
  for (int k = 0; k < 3; ++k) { v_new[k] *= ml; w_new[k] *= ma; }
  /* velocity clamps */
  float v2 = dot3(v_new, v_new), w2 = dot3(w_new, w_new);
  if (v2 > P->max_linear_velocity * P->max_linear_velocity) {
    float sc = P->max_linear_velocity / sqrtf(v2);
    for (int k = 0; k < 3; ++k) v_new[k] *= sc;
  }
  if (w2 > P->max_angular_velocity * P->max_angular_velocity) {
    float sc = P->max_angular_velocity / sqrtf(w2);
    for (int k = 0; k < 3; ++k) w_new[k] *= sc;
  }
  /* pose: semi-implicit (new velocities), integrateCore's exponential map */
  for (int k = 0; k < 3; ++k) p[k] = p[k] + v_new[k] * dt;
  float wm2 = dot3(w_new, w_new);
  if (wm2 != 0.0f) {
    float wm = sqrtf(wm2);
    float half = dt * wm * 0.5f;
    float sn, cs;
    om_sincosf(half, &sn, &cs);
    float sc = sn / wm;
    float qv[4] = {w_new[0] * sc, w_new[1] * sc, w_new[2] * sc, 0.0f};
    /* result = quatVel * q + q * cos ; Hamilton product written out */
    float x1 = qv[0], y1 = qv[1], z1 = qv[2];
    float x2 = q[0], y2 = q[1], z2 = q[2], w2q = q[3];
    float rx = x1 * w2q + y1 * z2 - z1 * y2;
    float ry = y1 * w2q + z1 * x2 - x1 * z2;
    float rz = z1 * w2q + x1 * y2 - y1 * x2;
    float rw = -(x1 * x2) - y1 * y2 - z1 * z2;
    rx += x2 * cs; ry += y2 * cs; rz += z2 * cs; rw += w2q * cs;
    float nn = sqrtf(rx * rx + ry * ry + rz * rz + rw * rw);
    q[0] = rx / nn; q[1] = ry / nn; q[2] = rz / nn; q[3] = rw / nn;
  }
  for (int k = 0; k < 3; ++k) { v[k] = v_new[k]; w[k] = w_new[k]; }
}

/* ------------------------------------------------------------------ */
/* One physics sub-step for all envs: BaseMultirotor.step             */
/* (base_multirotor.py:296-307) + integration.                         */
/*                                                                    */
/*  state   [N,13]  in/out   p q(xyzw) v_world w_world                 */
/*  action  [N,A]   in (clipped copy is written to action_clipped)     */
/*  thrust  [N,M]   in/out   MotorModel.current_motor_thrust           */
/*  kT, tau_inc, tau_dec [N,M]                                         */
/*  Kp,Kv,KR,Kw [N,3]                                                  */
/*  disturb [N,7] or NULL: (bernoulli, 3 force u01, 3 torque u01)      */
/*  outputs: euler[N,3] qveh[N,4] vveh[N,3] vbody[N,3] wbody[N,3],     */
/*           wrench_cmd[N,6] (controller output), body_wrench[N,6]     */
/*           (net LOCAL-frame wrench handed to the integrator)         */
/* ------------------------------------------------------------------ */
void orc_substep(const OrcRobotParams *P, int n, float *state, const float *action,
                 float *action_clipped, float *thrust, const float *kT, const float *tau_inc,
                 const float *tau_dec, const float *Kp, const float *Kv, const float *KR,
                 const float *Kw, const float *disturb, const float *disturb_max, float *euler,
                 float *qveh, float *vveh, float *vbody, float *wbody, float *wrench_cmd,
                 float *body_wrench, int do_integrate) {
  const int M = P->num_motors, A = P->num_actions;
  orc_update_states(n, state, euler, qveh, vveh, vbody, wbody);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    float *s = state + 13 * i;
    float *a = action_clipped + A * i;
    /* clip_actions, base_multirotor.py:207-211 */
    for (int k = 0; k < A; ++k) a[k] = clampf(action[A * i + k], -10.0f, 10.0f);
    float u[ORC_MAX_MOTORS];
    float wr[6] = {0, 0, 0, 0, 0, 0};
    if (P->controller == ORC_CTRL_NONE) {
      for (int j = 0; j < M; ++j)
        u[j] = motor_update_one(P, a[j], thrust[M * i + j], kT[M * i + j], tau_inc[M * i + j],
                                tau_dec[M * i + j]);
    } else {
      controller_one(P, s, euler + 3 * i, qveh + 4 * i, wbody + 3 * i, a, Kp + 3 * i, Kv + 3 * i,
                     KR + 3 * i, Kw + 3 * i, wr);
      /* control_allocation.py:85-93: u_ref = A+ w */
      for (int j = 0; j < M; ++j) {
        float r = 0.0f;
        for (int k = 0; k < 6; ++k) r += P->alloc_pinv[6 * j + k] * wr[k];
        u[j] = motor_update_one(P, r, thrust[M * i + j], kT[M * i + j], tau_inc[M * i + j],
                                tau_dec[M * i + j]);
      }
    }
    for (int k = 0; k < 6; ++k) wrench_cmd[6 * i + k] = wr[k];
    for (int j = 0; j < M; ++j) thrust[M * i + j] = u[j];
    /* net LOCAL-frame wrench on the rigid composite.
       motor_link: F_i=[0,0,u_i], tau_i=-cq dir_i F_i at each motor link
       (control_allocation.py:103-114) -> folded into wrench_map.
       root_link: w_out = A u (control_allocation.py:67-79).               */
    const float *W = P->root_link_mode ? P->alloc : P->wrench_map;
    float link[6];
    for (int r = 0; r < 6; ++r) {
      float acc = 0.0f;
      for (int j = 0; j < M; ++j) acc += W[M * r + j] * u[j];
      link[r] = acc;
    }
    /* the ROOT link's entry of robot_force / robot_torque_tensor: the allocator's wrench in root_link mode, else 0
       (base_multirotor.py:246-258); simulate_drag (:260-285, pre-physics body velocities) and apply_disturbance
       (:213-234; u01 drawn by the caller) accumulate into it with `+=`, in that order.  The net wrench on the rigid
       composite is the motor links' sum plus the root link's entry. */
    float root[6];
    for (int k = 0; k < 6; ++k) root[k] = P->root_link_mode ? link[k] : 0.0f;
    const float *vb = vbody + 3 * i, *wb = wbody + 3 * i;
    float vbn = norm3(vb);
    for (int k = 0; k < 3; ++k) {
      float dl = -P->lin_drag_linear[k] * vb[k];
      float dq = -P->lin_drag_quadratic[k] * vbn * vb[k];
      root[k] += dl + dq;
      float al = -P->ang_drag_linear[k] * wb[k];
      float aq = -P->ang_drag_quadratic[k] * fabsf(wb[k]) * wb[k];
      root[3 + k] += al + aq;
    }
    if (disturb) {
      const float *d = disturb + 7 * i;
      for (int k = 0; k < 6; ++k) {
        float lo = -disturb_max[k], hi = disturb_max[k];
        root[k] += ((hi - lo) * d[1 + k] + lo) * d[0];
      }
    }
    float bw[6];
    for (int k = 0; k < 6; ++k) bw[k] = P->root_link_mode ? root[k] : link[k] + root[k];
    for (int k = 0; k < 6; ++k) body_wrench[6 * i + k] = bw[k];
    if (do_integrate) integrate_one(P, s, bw, bw + 3);
  }
}

/* ------------------------------------------------------------------ */
/* Robot plug-in path.  BaseMultirotor.step(action) with    */
/* its OUTPUT as the reference leaves it: the per-body tensors           */
/* robot_force_tensor / robot_torque_tensor [N, num_bodies, 3], each     */
/* body's wrench in that body's own frame (base_multirotor.py:246-258   */
/* call_controller, :260-285 simulate_drag, :213-234 apply_disturbance;  */
/* control_allocation.py:52-114).  No integration.                       */
/* ------------------------------------------------------------------ */
void orc_robot_step(const OrcRobotParams *P, int n, const float *state, const float *action,
                    float *action_clipped, float *thrust, const float *kT, const float *tau_inc,
                    const float *tau_dec, const float *Kp, const float *Kv, const float *KR,
                    const float *Kw, const float *disturb, const float *disturb_max, float *euler,
                    float *qveh, float *vveh, float *vbody, float *wbody, float *wrench_cmd,
                    int num_bodies, const int *body_of_motor, float *force, float *torque) {
  const int M = P->num_motors;
  float *bw = (float *)malloc(sizeof(float) * 6 * (size_t)n);
  /* controller, allocation, motor model: orc_substep without its integration (state is not written) */
  orc_substep(P, n, (float *)state, action, action_clipped, thrust, kT, tau_inc, tau_dec, Kp, Kv, KR, Kw,
              NULL, NULL, euler, qveh, vveh, vbody, wbody, wrench_cmd, bw, 0);
  free(bw);
  for (int i = 0; i < n; ++i) {
    float *F = force + (size_t)i * num_bodies * 3, *T = torque + (size_t)i * num_bodies * 3;
    const float *u = thrust + M * i;
    for (int b = 0; b < num_bodies * 3; ++b) { F[b] = 0.0f; T[b] = 0.0f; }
    if (P->root_link_mode) { /* control_allocation.py:67-79: output_wrench = A u at the masked (root) body */
      const int b = body_of_motor[0];
      for (int r = 0; r < 6; ++r) {
        float acc = 0.0f;
        for (int j = 0; j < M; ++j) acc += P->alloc[M * r + j] * u[j];
        if (r < 3) F[3 * b + r] = acc; else T[3 * b + r - 3] = acc;
      }
    } else { /* :103-114: motor_forces = (0, 0, u); motor_torques = cq * motor_forces * (-dir) */
      for (int j = 0; j < M; ++j) {
        const int b = body_of_motor[j];
        F[3 * b + 2] = u[j];
        T[3 * b] = (P->cq * 0.0f) * (-P->motor_dir[j]);
        T[3 * b + 1] = (P->cq * 0.0f) * (-P->motor_dir[j]);
        T[3 * b + 2] = (P->cq * u[j]) * (-P->motor_dir[j]);
      }
    }
    const float *vb = vbody + 3 * i, *wb = wbody + 3 * i;
    const float vbn = norm3(vb);
    for (int k = 0; k < 3; ++k) {
      F[k] += (-P->lin_drag_linear[k] * vb[k]) + (-P->lin_drag_quadratic[k] * vbn * vb[k]);
      T[k] += (-P->ang_drag_linear[k] * wb[k]) + (-P->ang_drag_quadratic[k] * fabsf(wb[k]) * wb[k]);
    }
    if (disturb) {
      const float *d = disturb + 7 * i;
      for (int k = 0; k < 6; ++k) {
        const float lo = -disturb_max[k], hi = disturb_max[k];
        const float v = ((hi - lo) * d[1 + k] + lo) * d[0];
        if (k < 3) F[k] += v; else T[k - 3] += v;
      }
    }
  }
}

/* What PhysX does with per-body wrenches given in each body's own frame (gym.apply_rigid_body_force_tensors(...,   */
/* LOCAL_SPACE), IGE_env_manager.py:444-449) on a composite of rigidly connected bodies: the net wrench about the    */
/* centre of mass, F = sum R_b f_b, T = sum (r_b x R_b f_b + R_b t_b).  rot [B,9] row-major, pos [B,3]: body poses   */
/* in the root-link frame (URDF joint origins).                                                                      */
void orc_net_body_wrench(int n, int num_bodies, const float *rot, const float *pos, const float *force,
                         const float *torque, float *out) {
  for (int i = 0; i < n; ++i) {
    const float *F = force + (size_t)i * num_bodies * 3, *T = torque + (size_t)i * num_bodies * 3;
    float w[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < num_bodies; ++b) {
      const float *R = rot + 9 * b, *r = pos + 3 * b;
      float fr[3], tr[3];
      for (int c = 0; c < 3; ++c) {
        fr[c] = (R[3 * c] * F[3 * b] + R[3 * c + 1] * F[3 * b + 1]) + R[3 * c + 2] * F[3 * b + 2];
        tr[c] = (R[3 * c] * T[3 * b] + R[3 * c + 1] * T[3 * b + 1]) + R[3 * c + 2] * T[3 * b + 2];
      }
      const float cx = r[1] * fr[2] - r[2] * fr[1], cy = r[2] * fr[0] - r[0] * fr[2], cz = r[0] * fr[1] - r[1] * fr[0];
      w[0] += fr[0]; w[1] += fr[1]; w[2] += fr[2];
      w[3] += cx + tr[0]; w[4] += cy + tr[1]; w[5] += cz + tr[2];
    }
    for (int c = 0; c < 6; ++c) out[6 * i + c] = w[c];
  }
}

/* integrator alone (used to build hybrid "reference control + our physics" traces) */
void orc_integrate(const OrcRobotParams *P, int n, float *state, const float *body_wrench) {
  for (int i = 0; i < n; ++i) integrate_one(P, state + 13 * i, body_wrench + 6 * i, body_wrench + 6 * i + 3);
}

/* ------------------------------------------------------------------ */
/* a16: collision flag -- REPLACES PhysX contacts, ** parity unpinned **/
/* crash |= sphere(center = robot position, r) overlaps any box.       */
/* boxes: [N,B,10] = centre(3) quat xyzw(4) half-extents(3)            */
/* ------------------------------------------------------------------ */
void orc_collide_sphere_boxes(int n, int nb, float radius, const float *state, const float *boxes,
                              uint8_t *crashes) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float *p = state + 13 * i;
    uint8_t hit = 0;
    for (int b = 0; b < nb; ++b) {
      const float *bx = boxes + (size_t)(i * nb + b) * 10;
      float d[3] = {p[0] - bx[0], p[1] - bx[1], p[2] - bx[2]};
      float l[3];
      quat_rotate_inverse(bx + 3, d, l);
      float dist2 = 0.0f;
      for (int k = 0; k < 3; ++k) {
        float e = fabsf(l[k]) - bx[7 + k];
        if (e > 0.0f) dist2 += e * e;
      }
      if (dist2 < radius * radius) hit = 1;
    }
    crashes[i] = crashes[i] | hit;
  }
}

/* ------------------------------------------------------------------ */
/* a17: position-setpoint task reward / crash / obs                    */
/* position_setpoint_task.py:205-229, 245-282, 194-203                 */
/* ------------------------------------------------------------------ */
void orc_reward_position(int n, const float *state, const float *qveh, const float *wbody,
                         const float *target, uint8_t *crashes, float *reward) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float *p = state + 13 * i, *q = p + 3;
    float d[3] = {target[3 * i] - p[0], target[3 * i + 1] - p[1], target[3 * i + 2] - p[2]};
    float qi[4], pe[3];
    quat_conj(qveh + 4 * i, qi);
    quat_apply(qi, d, pe); /* quat_apply_inverse */
    float dist = norm3(pe);
    float pos_reward = 3.0f * om_expf(-8.0f * dist * dist) + 2.0f * om_expf(-4.0f * dist * dist);
    float dist_reward = (20.0f - dist) / 40.0f;
    float ez[3] = {0.0f, 0.0f, 1.0f}, up[3];
    quat_rotate(q, ez, up); /* quat_axis(q, 2) */
    float tilt = fabsf(1.0f - up[2]);
    /* `0.2 / tensor`: torch (eager and TorchScript alike) evaluates scalar / tensor as tensor.reciprocal() * scalar */
    float up_reward = (1.0f / (0.1f + tilt * tilt)) * 0.2f;
    const float *w = wbody + 3 * i;
    float spin = norm3(w);
    float ang_reward = (1.0f / (1.0f + spin * spin)) * 3.0f;
    float total = pos_reward + dist_reward + pos_reward * (up_reward + ang_reward);
    total = 1.0f * total;
    if (dist > 8.0f) crashes[i] = 1;
    if (crashes[i]) total = -20.0f;
    reward[i] = total;
  }
}

void orc_obs_position(int n, const float *state, const float *vbody, const float *wbody,
                      const float *target, float *obs) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const float *s = state + 13 * i;
    float *o = obs + 13 * i;
    for (int k = 0; k < 3; ++k) o[k] = target[3 * i + k] - s[k];
    for (int k = 0; k < 4; ++k) o[3 + k] = s[3 + k];
    for (int k = 0; k < 3; ++k) o[7 + k] = vbody[3 * i + k];
    for (int k = 0; k < 3; ++k) o[10 + k] = wbody[3 * i + k];
  }
}

/* ------------------------------------------------------------------ */
/* a18: navigation task reward, navigation_task.py:416-521             */
/* rp: 17 reward parameters in the order of navigation_task_config.py  */
/* pos_err / prev_pos_err [N,3] are in/out (prev <- cur, cur <- new)   */
/* ------------------------------------------------------------------ */
static float exp_reward(float mag, float ex, float v) { return mag * om_expf(-(v * v) * ex); }
static float exp_penalty(float mag, float ex, float v) { return mag * (om_expf(-(v * v) * ex) - 1.0f); }

void orc_reward_navigation(int n, const float *state, const float *qveh, const float *target,
                           const float *action, const float *prev_action, int num_actions,
                           float curriculum_progress, const float *rp, float *pos_err,
                           float *prev_pos_err, const uint8_t *crashes, float *reward) {
  float mult = 1.0f + 2.0f * curriculum_progress;
  for (int i = 0; i < n; ++i) {
    const float *p = state + 13 * i;
    float d[3] = {target[3 * i] - p[0], target[3 * i + 1] - p[1], target[3 * i + 2] - p[2]};
    for (int k = 0; k < 3; ++k) prev_pos_err[3 * i + k] = pos_err[3 * i + k];
    quat_rotate_inverse(qveh + 4 * i, d, pos_err + 3 * i);
    const float *pe = pos_err + 3 * i, *ppe = prev_pos_err + 3 * i;
    float dist = norm3(pe);
    float prev_dist = norm3(ppe);
    float pos_reward = exp_reward(rp[0], rp[1], dist);
    float close_reward = exp_reward(rp[2], rp[3], dist);
    float closer = prev_dist - dist;
    float closer_reward = (closer > 0.0f) ? rp[4] * closer : 2.0f * rp[4] * closer;
    float dist_reward = (20.0f - dist) / 20.0f;
    const float *a = action + num_actions * i, *pa = prev_action + num_actions * i;
    float dx = a[0] - pa[0], dz = a[2] - pa[2], dyaw = a[3] - pa[3];
    float diff_pen = exp_penalty(rp[5], rp[6], dx) + exp_penalty(rp[7], rp[8], dz) +
                     exp_penalty(rp[9], rp[10], dyaw);
    float abs_pen = curriculum_progress * exp_penalty(rp[11], rp[12], a[0]) +
                    curriculum_progress * exp_penalty(rp[13], rp[14], a[2]) +
                    curriculum_progress * exp_penalty(rp[15], rp[16], a[3]);
    float total_pen = diff_pen + abs_pen;
    float r = mult * (pos_reward + close_reward + closer_reward + dist_reward) + total_pen;
    if (crashes[i]) r = rp[17];
    reward[i] = r;
  }
}

/* ------------------------------------------------------------------ */
/* a20: reset helpers (the u01 draws are inputs; torch owns the RNG)   */
/* ------------------------------------------------------------------ */

/* BaseMultirotor.reset_idx, base_multirotor.py:177-205.  u01 [N,13];
 * mask[N] selects the envs to overwrite.                              */
void orc_reset_robot_state(int n, const uint8_t *mask, const float *u01, const float *min_state,
                           const float *max_state, const float *bounds_min, const float *bounds_max,
                           float *state) {
  for (int i = 0; i < n; ++i) {
    if (!mask[i]) continue;
    float r[13];
    for (int k = 0; k < 13; ++k) r[k] = (max_state[k] - min_state[k]) * u01[13 * i + k] + min_state[k];
    float *s = state + 13 * i;
    for (int k = 0; k < 3; ++k)
      s[k] = bounds_min[3 * i + k] + (bounds_max[3 * i + k] - bounds_min[3 * i + k]) * r[k];
    quat_from_euler_xyz(r[3], r[4], r[5], s + 3);
    for (int k = 0; k < 6; ++k) s[7 + k] = r[7 + k];
  }
}

/* quaternion helper exports for unit tests */
void orc_quat_from_euler(int n, const float *e, float *q) {
  for (int i = 0; i < n; ++i) quat_from_euler_xyz(e[3 * i], e[3 * i + 1], e[3 * i + 2], q + 4 * i);
}
void orc_quat_mul(int n, const float *a, const float *b, float *o) {
  for (int i = 0; i < n; ++i) quat_mul(a + 4 * i, b + 4 * i, o + 4 * i);
}
void orc_tf_apply(int n, const float *q, const float *t, const float *v, float *o) {
  for (int i = 0; i < n; ++i) {
    quat_apply(q + 4 * i, v + 3 * i, o + 3 * i);
    for (int k = 0; k < 3; ++k) o[3 * i + k] += t[3 * i + k];
  }
}

/* ------------------------------------------------------------------ */
/* Counter-based generator of the product's sync-free mode, restated:  */
/* Philox4x32-10 (Salmon et al. 2011), counter = (env, episode,        */
/* stream, j / 4), key = seed; uniform = top 24 bits / 2^24.           */
/* ------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

/* out[i][j] = j-th uniform of `stream` for env i in its episodes[i]-th reset */
void orc_rng_fill(uint64_t seed, int n, const int32_t *episodes, int stream, int count, float *out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < count; ++j) {
      uint32_t c[4] = {(uint32_t)i, (uint32_t)episodes[i], (uint32_t)stream, (uint32_t)(j >> 2)};
      philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
      out[(size_t)i * count + j] = (float)(c[j & 3] >> 8) * (1.0f / 16777216.0f);
    }
}

/* ------------------------------------------------------------------ */
/* AssetManager.reset_idx (asset_manager.py:51-71) with the half-       */
/* obstacle resample of EnvManager.reset_idx (env_manager.py:283-295).  */
/* u [N,K,13] is the draw each env actually uses, sel[N] the bernoulli  */
/* outcome; bounds [N,3] are the env's (already resampled) bounds.      */
/* ------------------------------------------------------------------ */
void orc_reset_assets(int n, int K, const uint8_t *mask, const float *u, const uint8_t *sel, const float *min_ratio,
                      const float *max_ratio, const float *bounds_min, const float *bounds_max, int num_obstacles,
                      int num_keep, float *asset_state) {
  for (int i = 0; i < n; ++i) {
    if (!mask[i]) continue;
    int n_full = num_obstacles > num_keep ? num_obstacles : num_keep;
    int half_o = num_obstacles / 2, half_k = num_keep / 2;
    int n_half = half_o > half_k ? half_o : half_k;
    int n_active = sel[i] ? n_half : n_full;
    for (int a = 0; a < K; ++a) {
      size_t b = ((size_t)i * K + a) * 13;
      float r[6];
      for (int c = 0; c < 6; ++c) r[c] = (max_ratio[b + c] - min_ratio[b + c]) * u[b + c] + min_ratio[b + c];
      float *st = asset_state + b;
      for (int c = 0; c < 3; ++c)
        st[c] = (a >= n_active) ? -1000.0f : bounds_min[3 * i + c] + (bounds_max[3 * i + c] - bounds_min[3 * i + c]) * r[c];
      quat_from_euler_xyz(r[3], r[4], r[5], st + 3);
    }
  }
}

/* ------------------------------------------------------------------ */
/* Navigation task observation, navigation_task.py:369-393.            */
/* u_vec / u_euler [N,3] are the two rand_like draws.  Slots 17.. hold  */
/* the gh x gw min-pooled depth image of sensor 0 (stands in for the    */
/* VAE latents, which are outside the simulation path).                 */
/* ------------------------------------------------------------------ */
void orc_obs_navigation(int n, const float *state, const float *euler, const float *qveh, const float *vbody,
                        const float *wbody, const float *actions, int num_actions, const float *target,
                        const float *u_vec, const float *u_euler, const float *pixels, int ns, int H, int W, int gh,
                        int gw, int obs_dim, float *obs) {
  for (int i = 0; i < n; ++i) {
    const float *p = state + 13 * i;
    float d[3] = {target[3 * i] - p[0], target[3 * i + 1] - p[1], target[3 * i + 2] - p[2]};
    float v[3];
    quat_rotate_inverse(qveh + 4 * i, d, v);
    float dist = norm3(v);
    float *o = obs + (size_t)i * obs_dim;
    for (int k = 0; k < 3; ++k) o[k] = (v[k] + 0.1f * 2.0f * u_vec[3 * i + k]) / dist;
    o[3] = dist;
    o[4] = ssa(euler[3 * i]) + 0.1f * (u_euler[3 * i] - 0.5f);
    o[5] = ssa(euler[3 * i + 1]) + 0.1f * (u_euler[3 * i + 1] - 0.5f);
    o[6] = 0.0f;
    for (int k = 0; k < 3; ++k) { o[7 + k] = vbody[3 * i + k]; o[10 + k] = wbody[3 * i + k]; }
    for (int k = 0; k < 4; ++k) o[13 + k] = actions[num_actions * i + k];
    if (pixels) {
      const float *img = pixels + (size_t)i * ns * H * W;
      int ch = (H + gh - 1) / gh, cw = (W + gw - 1) / gw;
      for (int cy = 0; cy < gh; ++cy)
        for (int cx = 0; cx < gw; ++cx) {
          float m = INFINITY;
          for (int y = cy * ch; y < (cy + 1) * ch && y < H; ++y)
            for (int x = cx * cw; x < (cx + 1) * cw && x < W; ++x)
              if (img[y * W + x] < m) m = img[y * W + x];
          int cell = cy * gw + cx;
          if (17 + cell < obs_dim) o[17 + cell] = m;
        }
    }
  }
}

/* ------------------------------------------------------------------ */
/* f2: LiDAR navigation task, task/lidar_navigation_task/              */
/*     lidar_navigation_task.py                                        */
/* ------------------------------------------------------------------ */

/* compute_reward (:554-719).  rp: 22 parameters in the order of
 * lidar_navigation_task_config.py:30-53; all arrays row-major [N,*].   */
void orc_reward_lidar_navigation(int n, const float *pos_err, const float *vveh, const float *wbody,
                                 const float *yaw_error, const uint8_t *crashes, const float *action,
                                 const float *prev_action, const float *ttc, float curriculum_progress,
                                 const float *rp, float *reward) {
  const float mult = (float)(1.0 + 2.0 * (double)curriculum_progress);
  const float cpf = curriculum_progress;
  for (int i = 0; i < n; ++i) {
    const float *pe = pos_err + 3 * i, *v = vveh + 3 * i, *a = action + 4 * i, *pa = prev_action + 4 * i;
    float dist = norm3(pe);
    float pos_reward = exp_reward(rp[0], rp[1], dist);
    float very_close = exp_reward(rp[2], rp[3], dist);
    float vel_norm = norm3(v);
    float vd[3], ug[3];
    for (int k = 0; k < 3; ++k) { vd[k] = v[k] / (vel_norm + 1e-6f); ug[k] = pe[k] / (dist + 1e-6f); }
    float reasonable_vel = exp_reward(2.0f, 2.0f, vel_norm - 2.0f);
    float vdc = vd[0] * ug[0] + vd[1] * ug[1] + vd[2] * ug[2];
    float near = dist / 3.0f;
    if (near > 1.0f) near = 1.0f;
    float vdc_reward = ((vdc > 0.0f) ? rp[4] * vdc * reasonable_vel : -0.2f) * near;
    float over = vel_norm - 3.0f;
    if (over < 0.0f) over = 0.0f;
    float vel_mag_pen = exp_penalty(2.0f, 2.0f, over);
    float close_to_goal = 1.0f - exp_reward(1.0f, 2.0f, dist);
    float vx = v[0] < 0.0f ? 0.0f : v[0];
    float neg_x_pen = exp_penalty(2.0f, 8.0f, vx) * close_to_goal;
    float vel_pen = vel_mag_pen + neg_x_pen;
    float low_vel = exp_reward(1.5f, 10.0f, vel_norm) + exp_reward(1.5f, 0.5f, vel_norm);
    float ye = yaw_error[i];
    float correct_yaw = exp_reward(2.0f, 0.2f, ye) + exp_reward(4.0f, 15.0f, ye);
    float alignment = exp_reward(1.0f, 2.0f, ye);
    float low_angvel = exp_reward(1.5f, 5.0f, wbody[3 * i + 2]) * alignment;
    float stable = (dist < 1.0f) ? (low_vel + correct_yaw + low_angvel) : 0.0f;
    float dist_reward = (20.0f - dist) / 20.0f;
    float diff_pen = exp_penalty(rp[5], rp[6], a[0] - pa[0]) + exp_penalty(rp[7], rp[8], a[1] - pa[1]) +
                     exp_penalty(rp[9], rp[10], a[2] - pa[2]) + exp_penalty(rp[11], rp[12], a[3] - pa[3]);
    float abs_pen = cpf * exp_penalty(rp[13], rp[14], a[0]) + cpf * exp_penalty(rp[17], rp[18], a[2]) +
                    cpf * exp_penalty(rp[19], rp[20], a[3]) + cpf * exp_penalty(rp[15], rp[16], a[1]);
    float total_pen = diff_pen + abs_pen;
    float t2 = ttc[i] * ttc[i];
    float ttc_pen = exp_reward(-3.0f, 2.0f, t2);
    float r = mult * (pos_reward + very_close * alignment + vdc_reward + dist_reward + stable + vel_pen + total_pen + ttc_pen);
    if (crashes[i]) r = rp[21];
    reward[i] = r;
  }
}

/* process_image_observation (:313-363) + add_noise_to_downsampled_lidar_data (:281-310).
 * pointcloud [N][H][W][3] world frame; ranges clipped to 10 outside [0.2, 10]; time to collision
 * from the velocity component along each ray; ph x pw min-pooling; optional noise tensors
 * ([N][H/ph][W/pw] each; low_* only used for rows >= low_row0, like ds[:, 10:]); inverse range. */
void orc_lidar_image_obs(int n, int H, int W, int ph, int pw, int low_row0, const float *pointcloud,
                         const float *robot_pos, const float *robot_linvel, const float *noise_mask,
                         const float *noise_val, const float *max_mask, const float *low_mask,
                         const float *low_val, float *ttc_out, float *ds_out) {
  const int oh = H / ph, ow = W / pw;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    float *rng = (float *)malloc(sizeof(float) * (size_t)H * W);
    const float *p = robot_pos + 3 * i, *lv = robot_linvel + 3 * i;
    float tmin = INFINITY;
    for (int j = 0; j < H * W; ++j) {
      const float *pc = pointcloud + ((size_t)i * H * W + j) * 3;
      float d[3] = {pc[0] - p[0], pc[1] - p[1], pc[2] - p[2]};
      float r = norm3(d);
      float u[3] = {d[0] / (r + 1e-6f), d[1] / (r + 1e-6f), d[2] / (r + 1e-6f)};
      float rc = r;
      if (rc > 10.0f) rc = 10.0f;
      if (rc < 0.2f) rc = 10.0f;
      rng[j] = rc;
      float vc = lv[0] * u[0] + lv[1] * u[1] + lv[2] * u[2];
      float t = (vc > 0.0f) ? rc / (vc + 1e-6f) : 10.0f;
      if (t < tmin) tmin = t;
    }
    ttc_out[i] = tmin < 0.0f ? 0.0f : (tmin > 10.0f ? 10.0f : tmin);
    for (int cy = 0; cy < oh; ++cy)
      for (int cx = 0; cx < ow; ++cx) {
        float m = INFINITY;
        for (int y = cy * ph; y < (cy + 1) * ph; ++y)
          for (int x = cx * pw; x < (cx + 1) * pw; ++x)
            if (rng[y * W + x] < m) m = rng[y * W + x];
        size_t c = (size_t)i * oh * ow + (size_t)cy * ow + cx;
        if (noise_mask && noise_mask[c] == 1.0f) m += noise_val[c];
        if (max_mask && max_mask[c] == 1.0f) m = 10.0f;
        if (low_mask && cy >= low_row0 && low_mask[c] == 1.0f) m = low_val[c];
        ds_out[c] = 1.0f / m;
      }
    free(rng);
  }
}

/* process_obs_for_task (:440-470): obs [N][17 + cells] */
void orc_obs_lidar_navigation(int n, const float *state, const float *euler, const float *qveh, const float *vbody,
                              const float *wbody, const float *actions, int num_actions, const float *target,
                              const float *target_yaw, const float *u_vec, const float *u_euler,
                              const float *downsampled, int cells, float *obs) {
  for (int i = 0; i < n; ++i) {
    const float *p = state + 13 * i;
    float d[3] = {target[3 * i] - p[0], target[3 * i + 1] - p[1], target[3 * i + 2] - p[2]};
    float v[3];
    quat_rotate_inverse(qveh + 4 * i, d, v);
    float dist = norm3(v);
    float *o = obs + (size_t)i * (17 + cells);
    for (int k = 0; k < 3; ++k) o[k] = (v[k] + 0.2f * (u_vec[3 * i + k] - 0.5f)) / dist;  /* 0.1 * 2 * (rand - 0.5) */
    o[3] = dist;
    float e0 = ssa(euler[3 * i]), e1 = ssa(euler[3 * i + 1]), e2 = ssa(euler[3 * i + 2]);
    o[4] = e0 + 0.1f * (u_euler[3 * i] - 0.5f);
    o[5] = e1 + 0.1f * (u_euler[3 * i + 1] - 0.5f);
    o[6] = ssa(target_yaw[i] - e2);
    for (int k = 0; k < 3; ++k) { o[7 + k] = vbody[3 * i + k]; o[10 + k] = wbody[3 * i + k]; }
    for (int k = 0; k < 4; ++k) o[13 + k] = actions[num_actions * i + k];
    for (int k = 0; k < cells; ++k) o[17 + k] = downsampled[(size_t)i * cells + k];
  }
}

/* ------------------------------------------------------------------ */
/* f4: IMU, aerial_gym/sensors/imu_sensor.py:74-131.  force [N,3] is    */
/* what the force sensor reports on the base link (body frame, total    */
/* force incl. gravity); one call = one physics sub-step.               */
/* ------------------------------------------------------------------ */
void orc_imu_update(int n, float mass, const float *g_world, float sqrt_dt, int world_frame, int enable_noise,
                    int enable_bias, const float *bias_std, const float *noise_std, const float *max_value,
                    const float *force, const float *quat, const float *wbody, const float *sensor_quat,
                    const float *z_noise, const float *z_bias, float *bias, float *meas) {
  for (int i = 0; i < n; ++i) {
    float at[3] = {force[3 * i] / mass, force[3 * i + 1] / mass, force[3 * i + 2] / mass};
    float q2[4], acc[3], ang[3], tmp[3];
    quat_mul(quat + 4 * i, sensor_quat + 4 * i, q2);
    if (world_frame) {
      float d[3] = {at[0] - g_world[0], at[1] - g_world[1], at[2] - g_world[2]};
      quat_rotate_inverse(q2, d, acc);
      quat_rotate_inverse(q2, wbody + 3 * i, ang);
    } else {
      quat_rotate_inverse(sensor_quat + 4 * i, at, acc);
      quat_rotate_inverse(q2, g_world, tmp);
      for (int c = 0; c < 3; ++c) acc[c] = acc[c] - tmp[c];
      quat_rotate_inverse(sensor_quat + 4 * i, wbody + 3 * i, ang);
    }
    for (int c = 0; c < 6; ++c) {
      float noise = z_noise[6 * i + c] * noise_std[c] / sqrt_dt;
      bias[6 * i + c] += z_bias[6 * i + c] * bias_std[c] * sqrt_dt;
      float v = (c < 3 ? acc[c] : ang[c - 3]) + (float)enable_bias * bias[6 * i + c] + (float)enable_noise * noise;
      float mx = max_value[c];
      v = v < mx ? v : mx;   /* tensor_clamp: max(min(x, hi), lo) */
      v = v > -mx ? v : -mx;
      meas[6 * i + c] = v;
    }
  }
}

/* IMUSensor.reset_idx (:144-153): bias = max_init * (2 (u - 0.5)); sensor quat from U(min, max) euler angles */
void orc_imu_reset(int n, const uint8_t *mask, const float *u_bias, const float *u_rot, const float *max_bias_init,
                   const float *min_rot, const float *max_rot, float *bias, float *sensor_quat) {
  for (int i = 0; i < n; ++i) {
    if (!mask[i]) continue;
    for (int c = 0; c < 6; ++c) bias[6 * i + c] = max_bias_init[c] * (2.0f * (u_bias[6 * i + c] - 0.5f));
    float e[3];
    for (int c = 0; c < 3; ++c) e[c] = (max_rot[c] - min_rot[c]) * u_rot[3 * i + c] + min_rot[c];
    quat_from_euler_xyz(e[0], e[1], e[2], sensor_quat + 4 * i);
  }
}

/* f4: kinematic obstacles -- twist [N*K,6] (world frame) written into the state, pose advanced by k
 * sub-steps with the integrator's rule (orc_integrate's pose update, constant velocities)           */
void orc_assets_integrate(int count, float *asset_state, const float *twist, float dt, int k) {
  for (int a = 0; a < count; ++a) {
    float *st = asset_state + (size_t)a * 13;
    const float *tw = twist + (size_t)a * 6;
    float wm2 = tw[3] * tw[3] + tw[4] * tw[4] + tw[5] * tw[5];
    float x1 = 0.0f, y1 = 0.0f, z1 = 0.0f, cs = 1.0f;
    if (wm2 != 0.0f) {
      float wm = sqrtf(wm2);
      float half = dt * wm * 0.5f;
      if (half > 60.0f) half = 60.0f;
      float sn;
      om_sincosf(half, &sn, &cs);
      float sc = sn / wm;
      x1 = tw[3] * sc; y1 = tw[4] * sc; z1 = tw[5] * sc;
    }
    for (int s = 0; s < k; ++s) {
      for (int c = 0; c < 3; ++c) st[c] = st[c] + tw[c] * dt;
      if (wm2 != 0.0f) {
        float *q = st + 3;
        float rx = x1 * q[3] + y1 * q[2] - z1 * q[1];
        float ry = y1 * q[3] + z1 * q[0] - x1 * q[2];
        float rz = z1 * q[3] + x1 * q[1] - y1 * q[0];
        float rw = -(x1 * q[0]) - y1 * q[1] - z1 * q[2];
        rx += q[0] * cs; ry += q[1] * cs; rz += q[2] * cs; rw += q[3] * cs;
        float nn = sqrtf(rx * rx + ry * ry + rz * rz + rw * rw);
        q[0] = rx / nn; q[1] = ry / nn; q[2] = rz / nn; q[3] = rw / nn;
      }
    }
    for (int c = 0; c < 6; ++c) st[7 + c] = tw[c];
  }
}

/* test access to oracle_math.h (tests/test_oracle_math.py pins the kernels against libm in double):
 * which = 0 sin, 1 cos, 2 atan2(x, y), 3 asin, 4 exp                                                  */
void orc_math_eval(int which, int n, const float *x, const float *y, float *out) {
  for (int i = 0; i < n; ++i) {
    switch (which) {
      case 0: out[i] = om_sinf(x[i]); break;
      case 1: out[i] = om_cosf(x[i]); break;
      case 2: out[i] = om_atan2f(x[i], y[i]); break;
      case 3: out[i] = om_asinf(x[i]); break;
      default: out[i] = om_expf(x[i]); break;
    }
  }
}
