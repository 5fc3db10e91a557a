// Per-env pieces of the navigation tasks' reset and sensor-pose path, shared by the stand-alone kernels (agx_task_glue.hip,
// agx_raycast.hip) and by the ONE launch that runs them behind the robot reset (k_nav_robot_side in agx_dynamics.hip): the same
// device functions, so the fused launch leaves what the separate launches leave.
#pragma once
#include "agx_common.h"
#include "agx_device_math.h"
#include "agx_rng.h"

namespace agx {

struct Ratio3 {
  float lo[3], hi[3];
};

AGX_DEV void bounds_from_draws(const AgxResetArgs &R, const float ub[6], float bmin[3], float bmax[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float ulo = ub[c], uhi = ub[3 + c];
    bmin[c] = (R.lower_bound_max[c] - R.lower_bound_min[c]) * ulo + R.lower_bound_min[c];
    bmax[c] = (R.upper_bound_max[c] - R.upper_bound_min[c]) * uhi + R.upper_bound_min[c];
  }
}
AGX_DEV void sample_bounds(const AgxResetArgs &R, int i, int rng_env, int episode, float bmin[3], float bmax[3]) {
  float ub[6];
  if (R.u_state) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ub[c] = R.u_bounds_lo[(size_t)i * 3 + c];
      ub[3 + c] = R.u_bounds_hi[(size_t)i * 3 + c];
    }
  } else {
    rng_fill<6>(R.seed, rng_env, episode, RNG_BOUNDS, ub);
  }
  bounds_from_draws(R, ub, bmin, bmax);
}


// AssetManager.reset_idx (asset_manager.py:51-71) + the half-obstacle resample (env_manager.py:283-295) for asset `a` of env `env`
// (the caller has checked reset_flag / reset_mask).  u1 / u2 / u_sel: host draws (strict mode) or NULL = the device generator,
// keyed by the GLOBAL env index and the env's episode count BEFORE the robot reset increments it.
AGX_DEV void reset_asset_one(const AgxEnvBuffers &B, const AgxResetArgs &R, int env, int a, int K, const float *__restrict__ u1,
                             const float *__restrict__ u2, const float *__restrict__ u_sel, const float *__restrict__ min_ratio,
                             const float *__restrict__ max_ratio, int num_obstacles, int nk, float *__restrict__ asset_state) {
  const int ep = B.episode_count ? B.episode_count[env] : 0;
  const bool host_rng = u1 != nullptr;
  const int genv = B.env_index_base + env;  // the device generator is keyed by the global env index
  float usel = host_rng ? u_sel[env] : rng_block(R.seed, genv, ep, RNG_ASSET_SEL, 0).v[0];
  // strict mode hands over the bernoulli outcome (0/1); the device generator thresholds at 0.15
  const bool sel = host_rng ? (usel > 0.0f) : (usel < 0.15f);
  const int n_active = sel ? max(num_obstacles / 2, nk / 2) : max(num_obstacles, nk);
  float bmin[3], bmax[3];
  sample_bounds(R, env, genv, ep, bmin, bmax);
  const size_t base = ((size_t)env * K + a) * 13;
  float ratio[6], ua[6];
  if (host_rng) {
#pragma unroll
    for (int c = 0; c < 6; ++c) ua[c] = sel ? u2[base + c] : u1[base + c];
  } else {
    rng_fill<6>(R.seed, genv, ep, RNG_ASSETS + a, ua);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) ratio[c] = (max_ratio[base + c] - min_ratio[base + c]) * ua[c] + min_ratio[base + c];
  float *st = asset_state + base;
  if (a >= n_active) {
    st[0] = -1000.0f; st[1] = -1000.0f; st[2] = -1000.0f;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) st[c] = bmin[c] + (bmax[c] - bmin[c]) * ratio[c];
  }
  Q4 q = quat_from_euler(ratio[3], ratio[4], ratio[5]);
  st[3] = q.x; st[4] = q.y; st[5] = q.z; st[6] = q.w;
}

// reset_idx of the navigation tasks (navigation_task.py:166-175, lidar_navigation_task.py:164-181) for env i (the caller has
// checked reset_flag / reset_mask): target = bounds_min + (bounds_max - bounds_min) * U(min_ratio, max_ratio); optional
// target_yaw = U(-pi, pi); optional robot_prev_actions = 0.  u: host draws [N][4] or NULL (device generator, keyed by the
// env's episode count AFTER the robot reset incremented it).
AGX_DEV void nav_target_reset_env(const AgxEnvBuffers &B, int n, int i, int num_actions, const Ratio3 &R, const float *__restrict__ u,
                                  float *__restrict__ target, float *__restrict__ target_yaw, int zero_prev_actions) {
  float uu[4];
  if (u) {
#pragma unroll
    for (int c = 0; c < 4; ++c) uu[c] = u[(size_t)i * 4 + c];
  } else {
    rng_fill<4>(B.rng_seed, B.env_index_base + i, B.episode_count ? B.episode_count[i] : 0, RNG_TARGET, uu);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float ratio = (R.hi[c] - R.lo[c]) * uu[c] + R.lo[c];
    float lo = B.bounds_min[c * n + i], hi = B.bounds_max[c * n + i];
    target[c * n + i] = lo + (hi - lo) * ratio;
  }
  if (target_yaw) target_yaw[i] = (kPi - (-kPi)) * uu[3] + (-kPi);
  if (zero_prev_actions)
    for (int c = 0; c < num_actions; ++c) B.prev_actions[c * n + i] = 0.0f;
}

// WarpSensor.reset_idx (warp_sensor.py:153-172) for sensor s of env i (idx = i * ns + s)
AGX_DEV void sensor_mount_reset_env(const AgxEnvBuffers &B, int i, int s, int idx, const Ratio3 &Tr, const Ratio3 &Ro,
                                    const float *__restrict__ u_pos, const float *__restrict__ u_rot, float *__restrict__ local_pos,
                                    float *__restrict__ local_quat) {
  float uu[6];
  if (u_pos) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { uu[c] = u_pos[(size_t)idx * 3 + c]; uu[3 + c] = u_rot[(size_t)idx * 3 + c]; }
  } else {
    rng_fill<6>(B.rng_seed, B.env_index_base + i, B.episode_count ? B.episode_count[i] : 0, RNG_SENSOR_MOUNT + s, uu);
  }
  float e[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    local_pos[(size_t)idx * 3 + c] = (Tr.hi[c] - Tr.lo[c]) * uu[c] + Tr.lo[c];
    e[c] = (Ro.hi[c] - Ro.lo[c]) * uu[3 + c] + Ro.lo[c];
  }
  Q4 q = quat_from_euler(e[0], e[1], e[2]);
  float *o = local_quat + (size_t)idx * 4;
  o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
}

// WarpSensor.update pose composition (warp_sensor.py:177-187) of sensor idx = i * ns + s
AGX_DEV void sensor_pose_env(const AgxEnvBuffers &B, int n, int i, int idx, const float *__restrict__ local_pos,
                             const float *__restrict__ local_quat, Q4 frame_quat, float *__restrict__ pos, float *__restrict__ quat) {
  const V3 p = V3{B.state[0 * n + i], B.state[1 * n + i], B.state[2 * n + i]};
  const Q4 q = Q4{B.state[3 * n + i], B.state[4 * n + i], B.state[5 * n + i], B.state[6 * n + i]};
  const float *lp = local_pos + (size_t)idx * 3, *lq = local_quat + (size_t)idx * 4;
  V3 sp = tf_apply(q, p, V3{lp[0], lp[1], lp[2]});
  Q4 sq = quat_mul(q, quat_mul(Q4{lq[0], lq[1], lq[2], lq[3]}, frame_quat));
  pos[(size_t)idx * 3] = sp.x; pos[(size_t)idx * 3 + 1] = sp.y; pos[(size_t)idx * 3 + 2] = sp.z;
  quat[(size_t)idx * 4] = sq.x; quat[(size_t)idx * 4 + 1] = sq.y; quat[(size_t)idx * 4 + 2] = sq.z; quat[(size_t)idx * 4 + 3] = sq.w;
}

}  // namespace agx
