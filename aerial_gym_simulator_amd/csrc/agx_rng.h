// Counter-based device generator of the sync-free mode: Philox4x32-10 (Salmon et al., SC'11),
// key = seed, counter = (env, episode | step, stream, block).  A draw is a pure function of its
// coordinates: no state, no launch, independent of batch size and sharding.
#pragma once
#include "agx_device_math.h"

namespace agx {

struct U4 {
  uint32_t x, y, z, w;
};
AGX_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
AGX_DEV float u01_from_bits(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0, 1), 24 bits like torch.rand

// stream ids (third counter word)
enum {
  RNG_BOUNDS = 0, RNG_STATE = 1, RNG_GAINS = 2, RNG_MOTOR = 3, RNG_ASSET_SEL = 4,
  RNG_LIDAR_NOISE = 5,  // counter word 1 = env step; block = pooled cell        (agx_lidar_image_obs)
  RNG_OBS_NOISE = 6,    // counter word 1 = env step; 6 draws                    (agx_obs_lidar_navigation)
  RNG_IMU_RESET = 7,    // counter word 1 = episode; 9 draws                     (agx_imu_reset)
  RNG_TARGET = 9,       // counter word 1 = episode; 4 draws (target ratio xyz, target yaw) (agx_nav_target_reset)
  RNG_IMU = 8,          // counter word 1 = env step; block = 3 * sub-step + j   (agx_imu_update)
  RNG_ASSETS = 16,      // + asset index (< 2^16 assets)
  RNG_SENSOR_MOUNT = 1 << 16,  // + sensor index; counter word 1 = episode; 6 draws  (agx_sensor_mount_reset)
  RNG_DISTURB = 1 << 20 // + sub-step; counter word 1 = env step
};

// Uniforms j = 4*blk .. 4*blk+3 of stream `stream` of env `env` in its `episode`-th reset: one
// Philox evaluation yields four draws.
struct F4 {
  float v[4];
};
AGX_DEV F4 rng_block(uint64_t seed, int env, int episode, int stream, int blk) {
  U4 r = philox4x32_10((uint32_t)env, (uint32_t)episode, (uint32_t)stream, (uint32_t)blk, (uint32_t)seed, (uint32_t)(seed >> 32));
  return F4{{u01_from_bits(r.x), u01_from_bits(r.y), u01_from_bits(r.z), u01_from_bits(r.w)}};
}
// COUNT uniforms of one stream into a register array (COUNT is a compile-time constant)
template <int COUNT>
AGX_DEV void rng_fill(uint64_t seed, int env, int episode, int stream, float (&out)[COUNT]) {
#pragma unroll
  for (int b = 0; b < (COUNT + 3) / 4; ++b) {
    F4 f = rng_block(seed, env, episode, stream, b);
#pragma unroll
    for (int l = 0; l < 4; ++l)
      if (4 * b + l < COUNT) out[4 * b + l] = f.v[l];
  }
}

}  // namespace agx
