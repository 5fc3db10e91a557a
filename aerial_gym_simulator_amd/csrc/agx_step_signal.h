// Device side of the producer half of the step exchange (include/aerial_gym_hip.h:
// AgxEnvBuffers.step_signal; consumer: k_wait_signal in agx_exchange.hip).
//
// gfx950 has one L2 per XCD: a plain store becomes visible to a kernel running on another XCD only
// after an L2 write-back.  A release fence at agent scope (`__threadfence()`) does that write-back
// for EVERY dirty line of the XCD, once per wave -- measured +6 us on the 14 us reset/observation
// kernel.  Instead
//   * exchange rows are stored with agent-scope atomic stores (row_store: write-through, `sc1`), so an
//     acknowledged store is already visible device-wide;
//   * every thread of a row-writing kernel calls step_rows_signal() exactly once, after its last row
//     store and on every control path (no early return in front of it: it holds a workgroup barrier):
//     each wave waits for its own stores to be acknowledged, the workgroup counts itself in, and the
//     last workgroup to arrive publishes the step number.
// The all-gather behind the flag is a new kernel on the communication stream: its start invalidates
// the caches it reads through.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aerial_gym_hip.h"

namespace agx {

// the env-step index of this launch: a kernel argument (eager stepping) or, when the step is replayed from a hipGraph
// whose kernel arguments are frozen, a word in device memory that the graph's last node advances
__device__ __forceinline__ int step_index(const AgxEnvBuffers &B) {
  return B.step_counter_dev ? *B.step_counter_dev : B.step_counter;
}

__device__ __forceinline__ void row_store(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void step_rows_signal(const AgxEnvBuffers &B) {
  if (B.step_signal == nullptr) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // compiler: keep the row stores above
  __builtin_amdgcn_s_waitcnt(0);                          // hardware: all of this wave's stores acknowledged
  __syncthreads();                                        // ... and those of the other waves of the workgroup
  if (threadIdx.x == 0) {
    const uint32_t groups = gridDim.x * gridDim.y * gridDim.z;
    const uint32_t arrived = __hip_atomic_fetch_add(B.step_signal + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == groups) {
      __hip_atomic_store(B.step_signal + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch (same stream)
      __hip_atomic_store(B.step_signal + B.flag_parity, (uint32_t)agx::step_index(B) + 1u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace agx
