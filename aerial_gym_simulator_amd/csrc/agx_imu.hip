// IMU sensor (aerial_gym/sensors/imu_sensor.py): specific force + body rates with a bias random walk,
// white noise, mount misalignment and saturation.  One lane per env; the k physics sub-steps of the env
// step are folded into one launch (k bias increments, last sub-step's noise and measurement).
#include "agx_common.h"
#include "agx_device_math.h"
#include "agx_rng.h"
#include "agx_step_signal.h"

namespace agx {

// two standard normals from two uniforms (Box-Muller); 1 - u keeps the log argument in (0, 1]
AGX_DEV void normal_pair(float u1, float u2, float &z0, float &z1) {
  float r = sqrtf(-2.0f * logf(1.0f - u1));
  float sn, cs;
  sincos_bounded(kTwoPi * u2, sn, cs);
  z0 = r * cs;
  z1 = r * sn;
}
// 6 normals of (env, step, sub-step, which): Philox blocks 3*slot .. 3*slot+2 -> 12 uniforms -> 6 normals
AGX_DEV void normals6(uint64_t seed, int env, int step, int slot, float z[6]) {
  float u[12];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    F4 f = rng_block(seed, env, step, RNG_IMU, 3 * slot + b);
#pragma unroll
    for (int l = 0; l < 4; ++l) u[4 * b + l] = f.v[l];
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) normal_pair(u[4 * j], u[4 * j + 1], z[2 * j], z[2 * j + 1]);
}

__global__ void __launch_bounds__(256) k_imu_update(AgxEnvBuffers B, int n, int k, AgxImuArgs A, const float *__restrict__ sensor_quat,
                                                     const float *__restrict__ z_noise, const float *__restrict__ z_bias,
                                                     float *__restrict__ bias, float *__restrict__ meas) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Q4 q = Q4{B.state[3 * n + i], B.state[4 * n + i], B.state[5 * n + i], B.state[6 * n + i]};
  const V3 wb = V3{B.derived[13 * n + i], B.derived[14 * n + i], B.derived[15 * n + i]};  // robot_body_angvel
  const V3 g = V3{A.g_world[0], A.g_world[1], A.g_world[2]};
  // force sensor on the base link: total force incl. gravity, body frame
  const V3 fb = V3{B.body_force[0 * n + i], B.body_force[1 * n + i], B.body_force[2 * n + i]};
  const V3 gb = quat_rotate_inverse(q, g);
  const V3 at = V3{(fb.x + A.mass * gb.x) / A.mass, (fb.y + A.mass * gb.y) / A.mass, (fb.z + A.mass * gb.z) / A.mass};
  const float *sq = sensor_quat + (size_t)i * 4;
  const Q4 qs = Q4{sq[0], sq[1], sq[2], sq[3]};
  const Q4 q2 = quat_mul(q, qs);
  V3 acc, ang;
  if (A.world_frame) {
    acc = quat_rotate_inverse(q2, at - g);
    ang = quat_rotate_inverse(q2, wb);
  } else {
    acc = quat_rotate_inverse(qs, at) - quat_rotate_inverse(q2, g);
    ang = quat_rotate_inverse(qs, wb);
  }
  float b[6], zn[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) b[c] = bias[(size_t)i * 6 + c];
  for (int s = 0; s < k; ++s) {  // update_bias once per physics sub-step
    float zb[6];
    if (z_bias) {
#pragma unroll
      for (int c = 0; c < 6; ++c) zb[c] = z_bias[((size_t)s * n + i) * 6 + c];
    } else {
      normals6(B.rng_seed, B.env_index_base + i, agx::step_index(B), 2 * s + 1, zb);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) b[c] += zb[c] * A.bias_std[c] * A.sqrt_dt;
  }
  if (z_noise) {
#pragma unroll
    for (int c = 0; c < 6; ++c) zn[c] = z_noise[(size_t)i * 6 + c];
  } else {
    normals6(B.rng_seed, B.env_index_base + i, agx::step_index(B), 2 * (k > 0 ? k - 1 : 0), zn);
  }
  const float v[6] = {acc.x, acc.y, acc.z, ang.x, ang.y, ang.z};
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    bias[(size_t)i * 6 + c] = b[c];
    float noise = zn[c] * A.noise_std[c] / A.sqrt_dt;
    float m = v[c] + (float)A.enable_bias * b[c] + (float)A.enable_noise * noise;
    m = fminf(m, A.max_value[c]);   // tensor_clamp = max(min(x, hi), lo)
    m = fmaxf(m, -A.max_value[c]);
    meas[(size_t)i * 6 + c] = m;
  }
}

__global__ void __launch_bounds__(256) k_imu_reset(AgxEnvBuffers B, int n, AgxImuArgs A, const float *__restrict__ u_bias,
                                                    const float *__restrict__ u_rot, float *__restrict__ bias,
                                                    float *__restrict__ sensor_quat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || B.reset_flag[B.flag_parity] == 0 || B.reset_mask[i] == 0) return;
  float u[9];
  if (u_bias) {
#pragma unroll
    for (int c = 0; c < 6; ++c) u[c] = u_bias[(size_t)i * 6 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) u[6 + c] = u_rot[(size_t)i * 3 + c];
  } else {
    rng_fill<9>(B.rng_seed, B.env_index_base + i, B.episode_count ? B.episode_count[i] : 0, RNG_IMU_RESET, u);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) bias[(size_t)i * 6 + c] = A.max_bias_init[c] * (2.0f * (u[c] - 0.5f));
  float e[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) e[c] = (A.max_rot[c] - A.min_rot[c]) * u[6 + c] + A.min_rot[c];
  Q4 q = quat_from_euler(e[0], e[1], e[2]);
  float *o = sensor_quat + (size_t)i * 4;
  o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
}

}  // namespace agx

using namespace agx;

extern "C" int agx_imu_update(const AgxEnvBuffers *B, int n, int k, const AgxImuArgs *A, const float *sensor_quat,
                              const float *z_noise, const float *z_bias, float *bias, float *imu_meas, void *stream) {
  AGX_REQUIRE(B && A && n > 0 && k >= 0 && k <= AGX_MAX_SUBSTEPS, "bad arguments");
  AGX_REQUIRE(B->state && B->derived && B->body_force, "agx_imu_update needs state, derived and body_force");
  AGX_REQUIRE(sensor_quat && bias && imu_meas, "null buffer");
  AGX_REQUIRE((z_noise == nullptr) == (z_bias == nullptr) || k == 0, "z_noise and z_bias: both tensors or both NULL");
  AGX_REQUIRE(A->mass > 0.0f && A->sqrt_dt > 0.0f, "mass and sqrt_dt must be positive");
  hipLaunchKernelGGL(k_imu_update, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, k, *A, sensor_quat, z_noise,
                     z_bias, bias, imu_meas);
  return check_launch("agx_imu_update");
}

extern "C" int agx_imu_reset(const AgxEnvBuffers *B, int n, const AgxImuArgs *A, const float *u_bias, const float *u_rot,
                             float *bias, float *sensor_quat, void *stream) {
  AGX_REQUIRE(B && A && n > 0, "bad arguments");
  AGX_REQUIRE(B->flag_parity == 0 || B->flag_parity == 1, "flag_parity must be 0 or 1");
  AGX_REQUIRE(B->reset_mask && B->reset_flag && bias && sensor_quat, "null buffer");
  AGX_REQUIRE((u_bias == nullptr) == (u_rot == nullptr), "u_bias and u_rot: both tensors or both NULL");
  hipLaunchKernelGGL(k_imu_reset, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, *A, u_bias, u_rot, bias,
                     sensor_quat);
  return check_launch("agx_imu_reset");
}
