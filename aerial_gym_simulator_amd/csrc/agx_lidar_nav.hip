// LiDAR navigation task kernels (aerial_gym/task/lidar_navigation_task/lidar_navigation_task.py):
// the reference's flagship training recipe -- `magpie` + `magpie_acceleration_control` + a 48 x 120
// world-frame point-cloud dome LiDAR -> 337-D observation.
//
//   k_lidar_image_obs           process_image_observation (:313-363) + add_noise_to_downsampled_lidar_data (:281-310)
//   k_reward_lidar_navigation   compute_rewards_and_crashes + compute_reward (:472-719) + truncation (:399-403)
//   k_obs_lidar_navigation      process_obs_for_task (:440-470)
//
// All three are epilogue work on data the env-step and ray-cast kernels left in HBM: one pass each,
// HBM bound (the image kernel reads the 69 KB point cloud of an env exactly once).
#include "agx_common.h"
#include "agx_device_math.h"
#include "agx_rng.h"
#include "agx_step_signal.h"

namespace agx {

AGX_DEV float exp_reward(float mag, float ex, float v) { return mag * exp_cw(-(v * v) * ex); }
AGX_DEV float exp_penalty(float mag, float ex, float v) { return mag * (exp_cw(-(v * v) * ex) - 1.0f); }

struct LidarNavParams {
  float rp[22];
};

// One workgroup per env.  Ranges of the H x W rays go to LDS (clipped), the time to collision is a
// workgroup min-reduction, the ph x pw min-pool reads LDS.
__global__ void __launch_bounds__(256) k_lidar_image_obs(AgxEnvBuffers B, int n, int H, int W, int ph, int pw, int low_row0,
                                                          const float *__restrict__ pointcloud,
                                                          const float *__restrict__ noise_mask,
                                                          const float *__restrict__ noise_val,
                                                          const float *__restrict__ max_mask,
                                                          const float *__restrict__ low_mask,
                                                          const float *__restrict__ low_val, int device_noise, int vec4,
                                                          float *__restrict__ ttc_out, float *__restrict__ ds_out) {
  extern __shared__ float lds[];  // [H * W] clipped ranges + [4] wave minima
  const int i = blockIdx.x, tid = threadIdx.x;
  const int npts = H * W;
  float *rng = lds, *wmin = lds + npts;
  const V3 p = V3{B.state[0 * n + i], B.state[1 * n + i], B.state[2 * n + i]};
  const V3 lv = V3{B.state[7 * n + i], B.state[8 * n + i], B.state[9 * n + i]};
  const float *pc = pointcloud + (size_t)i * npts * 3;
  float tmin = INFINITY;
  // one point: its clipped range to LDS, its time to collision into the thread's minimum
  auto point = [&](int j, float x, float y, float z) {
    V3 d = V3{x - p.x, y - p.y, z - p.z};
    float r = norm(d);  // torch.norm(world_dir_vectors, dim=-1)
    float den = r + 1e-6f;
    V3 u = V3{d.x / den, d.y / den, d.z / den};
    float rc = r;
    if (rc > 10.0f) rc = 10.0f;
    if (rc < 0.2f) rc = 10.0f;
    rng[j] = rc;
    float vc = lv.x * u.x + lv.y * u.y + lv.z * u.z;
    float t = (vc > 0.0f) ? rc / (vc + 1e-6f) : 10.0f;
    tmin = fminf(tmin, t);
  };
  if (vec4) {
    // The env's point cloud as float4: a thread takes FOUR points = three 16-byte loads (a wave 3 KB contiguous), two such groups
    // per trip, all six loads requested before the first is used (4-byte loads at a 12-byte stride: 142 -> 135 us at 8192 envs x
    // 48 x 120; the kernel reads 566 MB the ray-cast has just written and is bound by that).  A group read twice at the tail
    // changes neither a range nor the minimum.
    const float4 *pc4 = reinterpret_cast<const float4 *>(pc);
    const int groups = npts >> 2;
    for (int g0 = tid; g0 < groups; g0 += 2 * blockDim.x) {
      float4 v[2][3];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int g = min(g0 + q * (int)blockDim.x, groups - 1);
        v[q][0] = pc4[3 * g]; v[q][1] = pc4[3 * g + 1]; v[q][2] = pc4[3 * g + 2];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = 4 * min(g0 + q * (int)blockDim.x, groups - 1);
        point(j, v[q][0].x, v[q][0].y, v[q][0].z);
        point(j + 1, v[q][0].w, v[q][1].x, v[q][1].y);
        point(j + 2, v[q][1].z, v[q][1].w, v[q][2].x);
        point(j + 3, v[q][2].y, v[q][2].z, v[q][2].w);
      }
    }
  } else {
    // four points per trip, their twelve loads requested before the first is used (the loop was one memory latency per point)
    for (int j0 = tid; j0 < npts; j0 += 4 * blockDim.x) {
      float raw[4][3];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = min(j0 + q * (int)blockDim.x, npts - 1);  // (a point read twice changes neither its range nor the minimum)
        raw[q][0] = pc[3 * j]; raw[q][1] = pc[3 * j + 1]; raw[q][2] = pc[3 * j + 2];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) point(min(j0 + q * (int)blockDim.x, npts - 1), raw[q][0], raw[q][1], raw[q][2]);
    }
  }
  for (int off = 32; off > 0; off >>= 1) tmin = fminf(tmin, __shfl_xor(tmin, off));
  if ((tid & 63) == 0) wmin[tid >> 6] = tmin;
  __syncthreads();
  if (tid == 0) {
    float t = wmin[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fminf(t, wmin[w]);
    ttc_out[i] = t < 0.0f ? 0.0f : (t > 10.0f ? 10.0f : t);
  }
  const int oh = H / ph, ow = W / pw, cells = oh * ow;
  for (int c = tid; c < cells; c += blockDim.x) {
    const int cy = c / ow, cx = c % ow;
    float m = INFINITY;
    for (int y = cy * ph; y < (cy + 1) * ph; ++y)
      for (int x = cx * pw; x < (cx + 1) * pw; ++x) m = fminf(m, rng[y * W + x]);
    const size_t g = (size_t)i * cells + c;
    if (device_noise) {
      // u0 < 0.03: += U(0.2, 10); u2 < 0.02: = 10; rows >= low_row0 and u3 < 0.02: = U(0.2, 1)
      F4 a = rng_block(B.rng_seed, B.env_index_base + i, agx::step_index(B), RNG_LIDAR_NOISE, 2 * c);
      F4 b = rng_block(B.rng_seed, B.env_index_base + i, agx::step_index(B), RNG_LIDAR_NOISE, 2 * c + 1);
      if (a.v[0] < 0.03f) m += (10.0f - 0.2f) * a.v[1] + 0.2f;
      if (a.v[2] < 0.02f) m = 10.0f;
      if (cy >= low_row0 && a.v[3] < 0.02f) m = (1.0f - 0.2f) * b.v[0] + 0.2f;
    } else {
      if (noise_mask && noise_mask[g] == 1.0f) m += noise_val[g];
      if (max_mask && max_mask[g] == 1.0f) m = 10.0f;
      if (low_mask && cy >= low_row0 && low_mask[g] == 1.0f) m = low_val[g];
    }
    ds_out[g] = 1.0f / m;
  }
}

__global__ void __launch_bounds__(256) k_reward_lidar_navigation(AgxEnvBuffers B, int n, const float *__restrict__ target,
                                                                  const float *__restrict__ target_yaw,
                                                                  const float *__restrict__ action,
                                                                  const float *__restrict__ prev_action,
                                                                  const float *__restrict__ ttc, LidarNavParams R, float cpf,
                                                                  float mult, float *__restrict__ pos_err,
                                                                  float *__restrict__ prev_pos_err, int episode_len,
                                                                  int reset_on_collision, float *__restrict__ reward) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool reset = false;
  if (i < n) {
    const float *rp = R.rp;
    V3 p = V3{B.state[0 * n + i], B.state[1 * n + i], B.state[2 * n + i]};
    // derived tensors: euler(3) qveh(4) vveh(3) vbody(3) wbody(3)
    Q4 qveh = Q4{B.derived[3 * n + i], B.derived[4 * n + i], B.derived[5 * n + i], B.derived[6 * n + i]};
    V3 v = V3{B.derived[7 * n + i], B.derived[8 * n + i], B.derived[9 * n + i]};
    float wz = B.derived[15 * n + i];
    V3 tgt = V3{target[0 * n + i], target[1 * n + i], target[2 * n + i]};
    V3 ppe = V3{pos_err[0 * n + i], pos_err[1 * n + i], pos_err[2 * n + i]};
    prev_pos_err[0 * n + i] = ppe.x; prev_pos_err[1 * n + i] = ppe.y; prev_pos_err[2 * n + i] = ppe.z;
    V3 pe = quat_rotate_inverse(qveh, tgt - p);
    pos_err[0 * n + i] = pe.x; pos_err[1 * n + i] = pe.y; pos_err[2 * n + i] = pe.z;
    float ye = ssa(target_yaw[i] - ssa(B.derived[2 * n + i]));
    const float *a = action + (size_t)i * 4, *pa = prev_action + (size_t)i * 4;
    bool crash = B.crashes[i] != 0;

    float dist = norm(pe);
    float pos_reward = exp_reward(rp[0], rp[1], dist);
    float very_close = exp_reward(rp[2], rp[3], dist);
    float vel_norm = norm(v);
    float vden = vel_norm + 1e-6f, gden = dist + 1e-6f;
    float vdc = (v.x / vden) * (pe.x / gden) + (v.y / vden) * (pe.y / gden) + (v.z / vden) * (pe.z / gden);
    float reasonable_vel = exp_reward(2.0f, 2.0f, vel_norm - 2.0f);
    float vdc_reward = ((vdc > 0.0f) ? rp[4] * vdc * reasonable_vel : -0.2f) * fminf(dist / 3.0f, 1.0f);
    float vel_mag_pen = exp_penalty(2.0f, 2.0f, fmaxf(vel_norm - 3.0f, 0.0f));
    float close_to_goal = 1.0f - exp_reward(1.0f, 2.0f, dist);
    float neg_x_pen = exp_penalty(2.0f, 8.0f, fmaxf(v.x, 0.0f)) * close_to_goal;
    float vel_pen = vel_mag_pen + neg_x_pen;
    float low_vel = exp_reward(1.5f, 10.0f, vel_norm) + exp_reward(1.5f, 0.5f, vel_norm);
    float correct_yaw = exp_reward(2.0f, 0.2f, ye) + exp_reward(4.0f, 15.0f, ye);
    float alignment = exp_reward(1.0f, 2.0f, ye);
    float low_angvel = exp_reward(1.5f, 5.0f, wz) * alignment;
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
    if (crash) r = rp[21];
    reward[i] = r;

    bool trunc = B.sim_steps[i] > episode_len;
    B.truncations[i] = trunc ? 1 : 0;
    reset = (crash && reset_on_collision) || trunc;
    B.reset_mask[i] = reset ? 1 : 0;
  }
  if (__ballot(reset) != 0ull && (threadIdx.x & 63) == 0) atomicOr(B.reset_flag + B.flag_parity, 1);
}

// one wave per env: lane 0 writes the 17 state entries, all lanes copy the pooled LiDAR cells
AGX_DEV void obs_lidar_navigation_env(const AgxEnvBuffers &B, int n, int i, const float *__restrict__ target,
                                      const float *__restrict__ target_yaw, const float *__restrict__ u_vec,
                                      const float *__restrict__ u_euler, const float *__restrict__ downsampled, int cells,
                                      float *__restrict__ obs) {
  const int lane = threadIdx.x & 63;
  const int obs_dim = 17 + cells;
  float *o = obs + (size_t)i * obs_dim;
  float *row = B.step_rows[B.flag_parity] ? B.step_rows[B.flag_parity] + (size_t)i * (obs_dim + 3) : nullptr;
  if (lane == 0) {
    float uv[3], ue[3];
    if (u_vec) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { uv[c] = u_vec[(size_t)i * 3 + c]; ue[c] = u_euler[(size_t)i * 3 + c]; }
    } else {
      float u6[6];
      rng_fill<6>(B.rng_seed, B.env_index_base + i, agx::step_index(B), RNG_OBS_NOISE, u6);
#pragma unroll
      for (int c = 0; c < 3; ++c) { uv[c] = u6[c]; ue[c] = u6[3 + c]; }
    }
    V3 p = V3{B.state[0 * n + i], B.state[1 * n + i], B.state[2 * n + i]};
    Q4 qveh = Q4{B.derived[3 * n + i], B.derived[4 * n + i], B.derived[5 * n + i], B.derived[6 * n + i]};
    V3 tgt = V3{target[0 * n + i], target[1 * n + i], target[2 * n + i]};
    V3 v = quat_rotate_inverse(qveh, tgt - p);
    float dist = norm(v);
    float e0 = ssa(B.derived[0 * n + i]), e1 = ssa(B.derived[1 * n + i]), e2 = ssa(B.derived[2 * n + i]);
    float s[17];
    s[0] = (v.x + 0.2f * (uv[0] - 0.5f)) / dist;  // 0.1 * 2 * (rand_like - 0.5)
    s[1] = (v.y + 0.2f * (uv[1] - 0.5f)) / dist;
    s[2] = (v.z + 0.2f * (uv[2] - 0.5f)) / dist;
    s[3] = dist;
    s[4] = e0 + 0.1f * (ue[0] - 0.5f);
    s[5] = e1 + 0.1f * (ue[1] - 0.5f);
    s[6] = ssa(target_yaw[i] - e2);
#pragma unroll
    for (int c = 0; c < 6; ++c) s[7 + c] = B.derived[(10 + c) * n + i];
#pragma unroll
    for (int c = 0; c < 4; ++c) s[13 + c] = B.actions[c * n + i];
#pragma unroll
    for (int c = 0; c < 17; ++c) o[c] = s[c];
    if (row) {
#pragma unroll
      for (int c = 0; c < 17; ++c) row_store(B, row + c, s[c]);
      row_store(B, row + obs_dim, B.step_reward[i]);
      row_store(B, row + obs_dim + 1, B.crashes[i] ? 1.0f : 0.0f);
      row_store(B, row + obs_dim + 2, B.truncations[i] ? 1.0f : 0.0f);
    }
  }
  for (int c = lane; c < cells; c += 64) {
    float d = downsampled[(size_t)i * cells + c];
    o[17 + c] = d;
    if (row) row_store(B, row + 17 + c, d);
  }
}
__global__ void __launch_bounds__(256) k_obs_lidar_navigation(AgxEnvBuffers B, int n, const float *__restrict__ target,
                                                               const float *__restrict__ target_yaw,
                                                               const float *__restrict__ u_vec,
                                                               const float *__restrict__ u_euler,
                                                               const float *__restrict__ downsampled, int cells,
                                                               float *__restrict__ obs) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per env
  push_wait_for_slot(B);
  if (i < n) obs_lidar_navigation_env(B, n, i, target, target_yaw, u_vec, u_euler, downsampled, cells, obs);
  step_rows_signal(B);
}

}  // namespace agx

using namespace agx;

extern "C" int agx_lidar_image_obs(const AgxEnvBuffers *B, int n, int H, int W, int pool_h, int pool_w, int low_row0,
                                   const float *pointcloud, const float *noise_mask, const float *noise_val,
                                   const float *max_mask, const float *low_mask, const float *low_val, int device_noise,
                                   float *time_to_collision, float *downsampled, void *stream) {
  AGX_REQUIRE(B && B->state && n > 0, "bad arguments");
  AGX_REQUIRE(H > 0 && W > 0 && pool_h > 0 && pool_w > 0 && H >= pool_h && W >= pool_w, "bad image / pool sizes");
  AGX_REQUIRE(pointcloud && time_to_collision && downsampled, "null buffer");
  AGX_REQUIRE(!noise_mask || noise_val, "noise_mask needs noise_val");
  AGX_REQUIRE(!low_mask || low_val, "low_mask needs low_val");
  const size_t lds = ((size_t)H * W + 4) * sizeof(float);
  AGX_REQUIRE(lds <= 64 * 1024, "image too large for the LDS range buffer (%d x %d)", H, W);
  // 16-byte loads: every env's block must start on a 16-byte boundary (H W 12 bytes per env) and hold whole groups of four points
  const int vec4 = ((H * W) % 4 == 0 && (reinterpret_cast<uintptr_t>(pointcloud) & 15u) == 0) ? 1 : 0;
  hipLaunchKernelGGL(k_lidar_image_obs, dim3(n), dim3(256), lds, (hipStream_t)stream, *B, n, H, W, pool_h, pool_w, low_row0,
                     pointcloud, noise_mask, noise_val, max_mask, low_mask, low_val, device_noise, vec4, time_to_collision,
                     downsampled);
  return check_launch("agx_lidar_image_obs");
}

extern "C" int agx_reward_lidar_navigation(const AgxEnvBuffers *B, int n, const float *target, const float *target_yaw,
                                           const float *action, const float *prev_action, const float *time_to_collision,
                                           const float *rp, float cpf, float *pos_err, float *prev_pos_err, int episode_len,
                                           int reset_on_collision, float *reward, void *stream) {
  AGX_REQUIRE(B && n > 0, "bad arguments");
  AGX_REQUIRE(B->flag_parity == 0 || B->flag_parity == 1, "flag_parity must be 0 or 1");
  AGX_REQUIRE(target && target_yaw && action && prev_action && time_to_collision && rp && pos_err && prev_pos_err && reward,
              "null buffer");
  AGX_REQUIRE(B->state && B->derived && B->crashes && B->truncations && B->sim_steps && B->reset_mask && B->reset_flag,
              "null env buffer");
  LidarNavParams R;
  for (int c = 0; c < 22; ++c) R.rp[c] = rp[c];  // rp is a HOST pointer (22 config scalars)
  const float mult = (float)(1.0 + 2.0 * (double)cpf);  // MULTIPLICATION_FACTOR_REWARD is evaluated in double (:585)
  const int block = pick_block(n);
  hipLaunchKernelGGL(k_reward_lidar_navigation, dim3(blocks_for(n, block)), dim3(block), 0, (hipStream_t)stream, *B, n, target,
                     target_yaw, action, prev_action, time_to_collision, R, cpf, mult, pos_err, prev_pos_err, episode_len,
                     reset_on_collision, reward);
  return check_launch("agx_reward_lidar_navigation");
}

extern "C" int agx_obs_lidar_navigation(const AgxEnvBuffers *B, int n, const float *target, const float *target_yaw,
                                        const float *u_vec, const float *u_euler, const float *downsampled, int cells,
                                        float *obs, void *stream) {
  AGX_REQUIRE(B && n > 0 && cells >= 0, "bad arguments");
  AGX_REQUIRE(B->flag_parity == 0 || B->flag_parity == 1, "flag_parity must be 0 or 1");
  AGX_REQUIRE(target && target_yaw && obs && (cells == 0 || downsampled), "null buffer");
  AGX_REQUIRE((u_vec == nullptr) == (u_euler == nullptr), "u_vec and u_euler: both tensors or both NULL (device generator)");
  AGX_REQUIRE(B->state && B->derived && B->actions, "null env buffer");
  AGX_REQUIRE((!B->step_rows[0] && !B->step_rows[1]) || (B->step_rows[0] && B->step_rows[1] && B->step_reward), "bad step_rows");
  hipLaunchKernelGGL(k_obs_lidar_navigation, dim3(blocks_for(n, 4)), dim3(256), 0, (hipStream_t)stream, *B, n, target, target_yaw,
                     u_vec, u_euler, downsampled, cells, obs);
  return check_launch("agx_obs_lidar_navigation");
}
