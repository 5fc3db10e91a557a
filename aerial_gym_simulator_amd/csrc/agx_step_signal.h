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

// Stores into ANOTHER device's memory (a peer's receive buffer, mapped through HIP IPC and reached over xGMI): system scope,
// write-through (`sc0 sc1`) -- the data leaves this device's L2 with the store itself instead of waiting for an end-of-kernel
// write-back.  The arrival flags are raised by the FIRST kernel of the next step on the same stream, i.e. after this kernel
// has drained; the runtime may order two kernels of one stream with an AGENT-scope release only (enough for this device's own
// L2s, not a promise about lines bound for a peer), so the visibility of the rows at the peer must not depend on that release:
// with write-through stores it depends only on the stores being complete when the kernel ends, which every release scope
// guarantees.  (was a plain store + the assumption of a system-scope release between back-to-back kernels.)
__device__ __forceinline__ void store_peer(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
typedef float agx_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_peer4(float *p, float a, float b, float c, float d) {
  const agx_f4v v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// One element of an exchange row.  Without peer push: a write-through store (agent scope) that the wave later waits on before the
// launch publishes step_signal.  With peer push (AgxEnvBuffers.push_world > 0): a PLAIN store at this rank's own slice (local
// memory: made visible to the next kernel by the kernel boundary) and a system-scope write-through store at the same offset in
// every peer's receive buffer -- nothing waits for either inside the kernel; the arrival flags are raised by the FIRST kernel
// of the next step (push_publish_previous), which the stream runs behind this one.  (Write-through stores + an in-kernel
// s_waitcnt + the flags at the tail of this kernel cost 4.4 us per step even with every destination in local HBM:
// profiles/r03_exchange_experiments.txt.)
__device__ __forceinline__ void row_store(const AgxEnvBuffers &B, float *p, float v) {
  if (B.push_world > 0) {
    *p = v;
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j < B.push_world - 1) store_peer((float *)((char *)p + B.push_delta[j]), v);
  } else {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Four consecutive elements of an exchange row (p 16-byte aligned) under PEER PUSH: one 16-byte store per destination instead
// of four 4-byte ones.  What crosses xGMI is then 16-byte (one-lane kernels) or -- where four neighbouring lanes hold the four
// quarters of a 64-byte row, the lane-quad observation kernel -- whole-line writes, and a rank with seven peers issues 8 store
// instructions per row quarter instead of 32.  Callers check B.push_world > 0 (the other modes need the write-through stores).
__device__ __forceinline__ void row_store4_push(const AgxEnvBuffers &B, float *p, float a, float b, float c, float d) {
  *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
#pragma unroll
  for (int j = 0; j < 7; ++j)
    if (j < B.push_world - 1) store_peer4((float *)((char *)p + B.push_delta[j]), a, b, c, d);
}

// peer push: the rows of the PREVIOUS step are complete at every destination (the kernel that stored them has ended); lanes
// 0 .. world - 1 of the calling wave tell every rank so.  Called by one wave at the head of the first kernel of a step.
__device__ __forceinline__ void push_publish_previous(const AgxEnvBuffers &B) {
  if (B.push_world <= 0 || B.push_pub_seq == 0u) return;
  const int lane = (int)(threadIdx.x & 63u);
  if (lane < B.push_world)
    __hip_atomic_store(B.push_flags[lane] + B.push_pub_index, B.push_pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// peer push: before a row-writing kernel stores its first row, the slot it writes into must have been vacated -- every
// rank's rows of step push_wait_seq (two steps back) have arrived HERE, so every rank is past the kernels that read the
// slot being overwritten (four slots; agx_exchange.hip has the argument).  Lanes 0 .. world - 1 of every wave look at one
// flag each; the wait is bounded (10 s of the 100 MHz clock) and reports through push_timed_out.  Must be called by whole
// waves, before the first row_store.
__device__ __forceinline__ void push_wait_for_slot(const AgxEnvBuffers &B) {
  if (B.push_world <= 0 || B.push_wait_seq == 0u) return;
  const int lane = (int)(threadIdx.x & 63u);
  if (lane < B.push_world) {
    const uint32_t *f = B.push_flags[B.push_rank] + B.push_wait_index + lane;
    const uint64_t t0 = wall_clock64();
    uint32_t polls = 0;
    // relaxed: nothing the flag guards is READ by this kernel (it only must not overwrite too early), and an acquire at system
    // scope would invalidate the caches on every poll
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - B.push_wait_seq) < 0) {
      if (++polls < 128) __builtin_amdgcn_s_sleep(8);
      else __builtin_amdgcn_s_sleep(127);
      if (wall_clock64() - t0 > 1000000000ull) {
        if (B.push_timed_out) __hip_atomic_store(B.push_timed_out, B.push_wait_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
}

// The same wait split in two, for the env-step kernels: the flags are LOADED at the head of the kernel (lanes 0 .. world - 1 of
// the calling wave) and LOOKED AT at its end, so that the round trip to the (uncached) flag words overlaps the kernel's own
// work instead of delaying one of its waves; only if some rank's rows of two steps ago have still not arrived does the wave
// spin.  What it holds back is the row-writing kernel queued behind this one.
__device__ __forceinline__ uint32_t push_wait_peek(const AgxEnvBuffers &B) {
  if (B.push_world <= 0 || B.push_wait_seq == 0u) return 0u;
  const int lane = (int)(threadIdx.x & 63u);
  if (lane >= B.push_world) return B.push_wait_seq;
  return __hip_atomic_load(B.push_flags[B.push_rank] + B.push_wait_index + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void push_wait_finish(const AgxEnvBuffers &B, uint32_t peeked) {
  if (B.push_world <= 0 || B.push_wait_seq == 0u) return;
  if ((int32_t)(peeked - B.push_wait_seq) < 0) push_wait_for_slot(B);  // (lanes whose flag had arrived pass straight through)
}

__device__ __forceinline__ void step_rows_signal(const AgxEnvBuffers &B) {
  if (B.step_signal == nullptr || B.push_world > 0) return;  // (peer push: nothing to wait for or publish inside this kernel)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // compiler: keep the row stores above
  __builtin_amdgcn_s_waitcnt(0);                          // hardware: all of this wave's stores acknowledged
  __syncthreads();                                        // ... and those of the other waves of the workgroup
  if (threadIdx.x == 0) {
    uint32_t *counter = B.step_signal + 2;
    const uint32_t groups = gridDim.x * gridDim.y * gridDim.z;
    const uint32_t arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (arrived == groups) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch (same stream)
      __hip_atomic_store(B.step_signal + B.flag_parity, (uint32_t)agx::step_index(B) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace agx
