// Small per-env kernels that replace the torch glue of the navigation-type tasks in the sync-free mode:
// success / timeout bookkeeping with device-side curriculum counters, target (+ target yaw) resampling and
// sensor mount re-randomisation of the envs that reset.  Each replaces 5-15 tiny torch launches per env step;
// at the 512-2048 envs an RL run typically uses, those launches -- not the simulation -- bound the step time.
#include "agx_common.h"
#include "agx_device_math.h"
#include "agx_nav_parts.h"
#include "agx_rng.h"

namespace agx {

// navigation_task.py:311-326 / lidar_navigation_task.py:405-418: successes = truncated & within `radius` of the
// target & not crashed; timeouts = truncated & not success & not crashed.  counters += (successes, crashes, timeouts).
__global__ void __launch_bounds__(256) k_nav_bookkeeping(AgxEnvBuffers B, int n, const float *__restrict__ target, float radius,
                                                          uint8_t *__restrict__ successes, uint8_t *__restrict__ timeouts,
                                                          int32_t *__restrict__ counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool succ = false, tout = false, crash = false;
  if (i < n) {
    V3 d = V3{target[0 * n + i] - B.state[0 * n + i], target[1 * n + i] - B.state[1 * n + i], target[2 * n + i] - B.state[2 * n + i]};
    const bool near = norm(d) < radius;
    crash = B.crashes[i] != 0;
    const bool trunc = B.truncations[i] != 0;
    succ = trunc && near && !crash;
    tout = trunc && !succ && !crash;
    successes[i] = succ ? 1 : 0;
    timeouts[i] = tout ? 1 : 0;
  }
  const unsigned long long ms = __ballot(succ), mc = __ballot(crash), mt = __ballot(tout);
  if ((threadIdx.x & 63) == 0) {
    if (ms) atomicAdd(counters + 0, __popcll(ms));
    if (mc) atomicAdd(counters + 1, __popcll(mc));
    if (mt) atomicAdd(counters + 2, __popcll(mt));
  }
}

// EnvManager.reset_terminated_and_truncated_envs (env_manager.py:364-371) for callers that did not go through one of
// the task reward kernels (stand-alone EnvManager use, user tasks that set `truncations` in torch like every reference
// task does): reset set = crashes * reset_on_collision + truncations, and the step's device flag.
__global__ void __launch_bounds__(256) k_reset_set(AgxEnvBuffers B, int n, int reset_on_collision) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool reset = false;
  if (i < n) {
    reset = (B.crashes[i] != 0 && reset_on_collision != 0) || B.truncations[i] != 0;
    B.reset_mask[i] = reset ? 1 : 0;
  }
  if (__ballot(reset) != 0ull && (threadIdx.x & 63) == 0) atomicOr(B.reset_flag + B.flag_parity, 1);
}

// last node of a captured env step: the step index the NEXT replay's kernels read (AgxEnvBuffers.step_counter_dev)
__global__ void k_step_counter_advance(int32_t *counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *counter = (*counter + 1) & 0x7FFFFFFF;
}

// reset_idx of the navigation tasks (navigation_task.py:166-175, lidar_navigation_task.py:164-181) for the envs of
// reset_mask: target = bounds_min + (bounds_max - bounds_min) * U(min_ratio, max_ratio); optional target_yaw =
// U(-pi, pi); optional robot_prev_actions = 0.
__global__ void __launch_bounds__(256) k_nav_target_reset(AgxEnvBuffers B, int n, int num_actions, Ratio3 R, const float *__restrict__ u,
                                                           float *__restrict__ target, float *__restrict__ target_yaw,
                                                           int zero_prev_actions) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || B.reset_flag[B.flag_parity] == 0 || B.reset_mask[i] == 0) return;
  nav_target_reset_env(B, n, i, num_actions, R, u, target, target_yaw, zero_prev_actions);
}

// WarpSensor.reset_idx (warp_sensor.py:153-172) for the envs of reset_mask
__global__ void __launch_bounds__(256) k_sensor_mount_reset(AgxEnvBuffers B, int n, int ns, Ratio3 Tr, Ratio3 Ro,
                                                             const float *__restrict__ u_pos, const float *__restrict__ u_rot,
                                                             float *__restrict__ local_pos, float *__restrict__ local_quat) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * ns) return;
  const int i = idx / ns, s = idx % ns;
  if (B.reset_flag[B.flag_parity] == 0 || B.reset_mask[i] == 0) return;
  sensor_mount_reset_env(B, i, s, idx, Tr, Ro, u_pos, u_rot, local_pos, local_quat);
}

}  // namespace agx

using namespace agx;

static Ratio3 ratio3(const float *lo, const float *hi) {
  Ratio3 r;
  for (int c = 0; c < 3; ++c) { r.lo[c] = lo[c]; r.hi[c] = hi[c]; }
  return r;
}

// The action transformations the reference's task configs ship (navigation_task_config.py:87-117,
// lidar_navigation_task_config.py:98-108; the fully actuated set-point map of configs[3]): policy action [N][4] in [-1, 1] ->
// controller command [N][A_out].  The same arithmetic as the torch functions in config/task_config.py, one launch instead of
// nine; sine / cosine through sincos_bounded (correctly rounded, like the kernels behind it).
__global__ void __launch_bounds__(256) k_action_transform(int kind, int n, const float *__restrict__ a_in, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {  // torch.clamp(action, -1, 1): a NaN action stays NaN (fminf / fmaxf alone would turn it into -1)
    const float x = a_in[(size_t)i * 4 + c];
    a[c] = (x != x) ? x : fminf(fmaxf(x, -1.0f), 1.0f);
  }
  if (kind == AGX_ACTION_NAV_VELOCITY) {  // (speed, inclination, yaw rate) -> vehicle-frame velocity command
    const float speed = a[0] + 1.0f;
    const float inclination = 0.785398185253143310546875f * a[1];  // float(pi / 4) * a1
    float sn, cs;
    sincos_bounded(inclination, sn, cs);
    float *o = out + (size_t)i * 4;
    o[0] = speed * cs;
    o[1] = 0.0f;
    o[2] = speed * sn;
    o[3] = a[2] * 1.04719758033752441406f;  // a2 * float(pi / 3)
  } else if (kind == AGX_ACTION_LIDAR_ACCELERATION) {  // +-2 m/s^2, +-pi/3 rad/s
    float *o = out + (size_t)i * 4;
    o[0] = a[0] * 2.0f; o[1] = a[1] * 2.0f; o[2] = a[2] * 2.0f;
    o[3] = a[3] * 1.04719758033752441406f;
  } else {  // AGX_ACTION_FULLY_ACTUATED_POSE: position set-point within +-(5, 5, 2.5) m, level attitude at yaw * pi
    float *o = out + (size_t)i * 7;
    const float half = (0.5f * 3.14159274101257324f) * a[3];  // (0.5 * torch.pi) as a python float, rounded once, times a3
    float sn, cs;
    sincos_bounded(half, sn, cs);
    o[0] = a[0] * 5.0f; o[1] = a[1] * 5.0f; o[2] = a[2] * 2.5f;
    o[3] = 0.0f; o[4] = 0.0f; o[5] = sn; o[6] = cs;
  }
}

extern "C" int agx_action_transform(int kind, int n, const float *actions_in, float *out, void *stream) {
  AGX_REQUIRE(kind >= AGX_ACTION_NAV_VELOCITY && kind <= AGX_ACTION_FULLY_ACTUATED_POSE, "agx_action_transform: kind %d", kind);
  AGX_REQUIRE(n > 0 && actions_in && out, "agx_action_transform: null buffer");
  hipLaunchKernelGGL(k_action_transform, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, kind, n, actions_in, out);
  return check_launch("agx_action_transform");
}

extern "C" int agx_step_counter_advance(const AgxEnvBuffers *B, void *stream) {
  AGX_REQUIRE(B && B->step_counter_dev, "agx_step_counter_advance: buf->step_counter_dev is not set");
  hipLaunchKernelGGL(k_step_counter_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, const_cast<int32_t *>(B->step_counter_dev));
  return check_launch("agx_step_counter_advance");
}

extern "C" int agx_reset_set(const AgxEnvBuffers *B, int n, int reset_on_collision, void *stream) {
  AGX_REQUIRE(B && n > 0 && B->crashes && B->truncations && B->reset_mask && B->reset_flag, "agx_reset_set: null buffer");
  AGX_REQUIRE(B->flag_parity == 0 || B->flag_parity == 1, "flag_parity must be 0 or 1");
  hipLaunchKernelGGL(k_reset_set, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, reset_on_collision);
  return check_launch("agx_reset_set");
}

extern "C" int agx_nav_bookkeeping(const AgxEnvBuffers *B, int n, const float *target, float radius, uint8_t *successes,
                                   uint8_t *timeouts, int32_t *counters, void *stream) {
  AGX_REQUIRE(B && n > 0 && B->state && B->crashes && B->truncations, "bad arguments");
  AGX_REQUIRE(target && successes && timeouts && counters, "null buffer");
  hipLaunchKernelGGL(k_nav_bookkeeping, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, target, radius, successes,
                     timeouts, counters);
  return check_launch("agx_nav_bookkeeping");
}

extern "C" int agx_nav_target_reset(const AgxEnvBuffers *B, int n, int num_actions, const float *min_ratio, const float *max_ratio,
                                    const float *u, float *target, float *target_yaw, int zero_prev_actions, void *stream) {
  AGX_REQUIRE(B && n > 0 && B->reset_mask && B->reset_flag && B->bounds_min && B->bounds_max, "bad arguments");
  AGX_REQUIRE(B->flag_parity == 0 || B->flag_parity == 1, "flag_parity must be 0 or 1");
  AGX_REQUIRE(min_ratio && max_ratio && target, "null buffer");
  AGX_REQUIRE(u || B->episode_count, "device RNG needs buf->episode_count");
  AGX_REQUIRE(!zero_prev_actions || (B->prev_actions && num_actions >= 1 && num_actions <= AGX_MAX_ACTIONS), "bad prev_actions");
  hipLaunchKernelGGL(k_nav_target_reset, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, num_actions,
                     ratio3(min_ratio, max_ratio), u, target, target_yaw, zero_prev_actions);  // ratios: HOST pointers
  return check_launch("agx_nav_target_reset");
}

extern "C" int agx_sensor_mount_reset(const AgxEnvBuffers *B, int n, int ns, const float *min_translation,
                                      const float *max_translation, const float *min_rot, const float *max_rot, const float *u_pos,
                                      const float *u_rot, float *local_pos, float *local_quat, void *stream) {
  AGX_REQUIRE(B && n > 0 && ns > 0 && B->reset_mask && B->reset_flag, "bad arguments");
  AGX_REQUIRE(B->flag_parity == 0 || B->flag_parity == 1, "flag_parity must be 0 or 1");
  AGX_REQUIRE(min_translation && max_translation && min_rot && max_rot && local_pos && local_quat, "null buffer");
  AGX_REQUIRE((u_pos == nullptr) == (u_rot == nullptr), "u_pos and u_rot: both tensors or both NULL");
  AGX_REQUIRE(u_pos || B->episode_count, "device RNG needs buf->episode_count");
  hipLaunchKernelGGL(k_sensor_mount_reset, dim3(blocks_for(n * ns, 256)), dim3(256), 0, (hipStream_t)stream, *B, n, ns,
                     ratio3(min_translation, max_translation), ratio3(min_rot, max_rot), u_pos, u_rot, local_pos, local_quat);
  return check_launch("agx_sensor_mount_reset");
}
