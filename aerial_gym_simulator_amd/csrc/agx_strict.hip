// The reference-faithful ("strict_rng") position step without a stream synchronisation and with ONE launch for its draws.
//
// What the mode promises: torch's generator is consumed by the calls the reference makes, in its order, with its shapes -- and ONLY
// on the steps on which the reference makes them: `if len(env_ids) > 0` after `nonzero()` (env_manager.py:364-375), then the
// rand_like calls of the robot reset (IGE_env_manager.py:513-519, base_multirotor.py:177-205, motor_model.py:140-154).  The host
// therefore needs one bit per step -- "does any env reset?" -- before it may enqueue the draws.  Rounds 1-5 read it with
// `tensor.item()` (a device-to-host copy + stream synchronisation: 16 us) and drew with seven `uniform_` launches through the
// dispatcher (38 us of host time): 62 us per step against 13 for the sync-free step (profiles/r06_strict_probe.jsonl).  Here:
//
//   * k_publish_flag: a one-lane launch behind the env-step kernel stores (seq << 1 | flag) into a word of mapped, coherent
//     HOST memory; the host spins on that word (agx_position_task_step_strict below): no copy engine, no hipStreamSynchronize;
//   * k_torch_uniform_fill: every number `Tensor.uniform_(0, 1)` would produce for up to 8 float32 tensors called one after
//     another on the device generator at (seed, offset), in one launch -- a restatement of
//       ATen/native/cuda/DistributionTemplates.h  calc_execution_policy + distribution_elementwise_grid_stride_kernel +
//                                                 uniform_kernel's transform (value == to ? from : value)
//       rocrand/rocrand_philox4x32_10.h           seed / discard_subsequence / discard (Philox4x32-10, counter = (offset / 4 +
//                                                 call, subsequence = thread index), key = seed)
//       rocrand/rocrand_uniform.h                 uniform_distribution: 2^-32 + v * 2^-32
//     (torch 2.10 + ROCm 7: what hiprand_init / hiprand_uniform4 resolve to).  The caller then moves the generator's offset by the
//     amount those calls would have (Generator.set_offset).  tests/test_gpu_strict_fast.py compares both, bit for bit, with the
//     real `uniform_` calls on the same seeds -- an upgrade of torch that changes its kernel shows up there, and
//     EnvManager falls back to the dispatcher calls when the self-check at construction fails.
#include "agx_common.h"
#include "agx_rng.h"

#include <chrono>

namespace agx {

struct UniformSegs {
  float *out[AGX_MAX_UNIFORM_SEGMENTS];
  unsigned long long numel[AGX_MAX_UNIFORM_SEGMENTS];
  unsigned long long offset[AGX_MAX_UNIFORM_SEGMENTS];  // philox offset of the call (a multiple of 4)
  unsigned int threads[AGX_MAX_UNIFORM_SEGMENTS];       // blockDim.x * gridDim.x of torch's launch for this tensor
  unsigned int first_block[AGX_MAX_UNIFORM_SEGMENTS + 1];
  int count;
};

__global__ void __launch_bounds__(256) k_torch_uniform_fill(UniformSegs S, unsigned long long seed) {
  int j = 0;
#pragma unroll
  for (int s = 1; s < AGX_MAX_UNIFORM_SEGMENTS; ++s)
    if (s < S.count && blockIdx.x >= S.first_block[s]) j = s;
  const unsigned long long li = (unsigned long long)(blockIdx.x - S.first_block[j]) * 256ull + threadIdx.x;
  if (li >= S.numel[j]) return;
  // element li is written by torch's thread idx = li mod T in loop iteration `call` = li / (4 T), as component ii = (li mod 4 T) / T
  // of the float4 that iteration's curand_uniform4 returned
  const unsigned long long T = S.threads[j];
  const unsigned long long call = li / (4ull * T), rem = li % (4ull * T);
  const unsigned int ii = (unsigned int)(rem / T);
  const unsigned long long idx = rem % T;
  const unsigned long long ctr = S.offset[j] / 4ull + call;
  const U4 r = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t v = ii == 0 ? r.x : (ii == 1 ? r.y : (ii == 2 ? r.z : r.w));
  const float two_m32 = 2.3283064e-10f;              // ROCRAND_2POW32_INV (= 2^-32 exactly: v * 2^-32 is exact, one rounding in the sum)
  const float u = two_m32 + (float)v * two_m32;       // (0, 1]
  S.out[j][li] = u == 1.0f ? 0.0f : u;                // uniform_kernel: value == to ? from : value, range 1, from 0
}

__global__ void k_publish_flag(const int32_t *__restrict__ reset_flag, int parity, uint32_t seq, uint32_t *host_word) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint32_t f = reset_flag[parity] != 0 ? 1u : 0u;
    __hip_atomic_store(host_word, (seq << 1) | f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static int fill_plan(int count, float *const *out, const int64_t *numel, uint64_t offset, int sm_count, int max_threads_per_sm, UniformSegs &S,
                     uint64_t *offset_after, unsigned *blocks) {
  AGX_REQUIRE(count >= 1 && count <= AGX_MAX_UNIFORM_SEGMENTS && out && numel, "1 .. %d tensors", AGX_MAX_UNIFORM_SEGMENTS);
  AGX_REQUIRE(sm_count > 0 && max_threads_per_sm >= 256, "device properties (multiProcessorCount, maxThreadsPerMultiProcessor) needed");
  AGX_REQUIRE((offset & 3ull) == 0, "the generator's offset is a multiple of 4 (CUDAGeneratorImpl)");
  S.count = count;
  unsigned total = 0;
  for (int j = 0; j < count; ++j) {
    AGX_REQUIRE(out[j] && numel[j] > 0 && numel[j] < (1ll << 31), "tensor %d: null or numel outside (0, 2^31)", j);
    // calc_execution_policy(numel, unroll_factor = 4): grid = min(SMs * (maxThreadsPerSM / 256), ceil(numel / 256)),
    // counter_offset = ((numel - 1) / (256 * grid * 4) + 1) * 4
    const uint64_t n = (uint64_t)numel[j];
    uint64_t grid = (n + 255ull) / 256ull;
    const uint64_t cap = (uint64_t)sm_count * (uint64_t)(max_threads_per_sm / 256);
    if (grid > cap) grid = cap;
    S.out[j] = out[j];
    S.numel[j] = n;
    S.offset[j] = offset;
    S.threads[j] = (unsigned)(256ull * grid);
    S.first_block[j] = total;
    total += (unsigned)((n + 255ull) / 256ull);
    offset += ((n - 1ull) / (256ull * grid * 4ull) + 1ull) * 4ull;
  }
  S.first_block[count] = total;
  *offset_after = offset;
  *blocks = total;
  return AGX_OK;
}

}  // namespace agx
using namespace agx;

extern "C" int agx_torch_uniform_fill(int count, float *const *out, const int64_t *numel, uint64_t seed, uint64_t offset, int sm_count,
                                      int max_threads_per_sm, uint64_t *offset_after, void *stream) {
  UniformSegs S;
  unsigned blocks = 0;
  uint64_t after = 0;
  if (int e = fill_plan(count, out, numel, offset, sm_count, max_threads_per_sm, S, &after, &blocks)) return e;
  hipLaunchKernelGGL(k_torch_uniform_fill, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (unsigned long long)seed);
  if (offset_after) *offset_after = after;
  return check_launch("agx_torch_uniform_fill");
}

extern "C" int agx_host_word_create(uint32_t **word) {
  AGX_REQUIRE(word, "null result");
  void *p = nullptr;
  hipError_t e = hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) return fail(AGX_E_LAUNCH, "hipHostMalloc (mapped, coherent): %s", hipGetErrorString(e));
  *(volatile uint32_t *)p = 0u;
  *word = (uint32_t *)p;
  return AGX_OK;
}

extern "C" int agx_host_word_destroy(uint32_t *word) {
  if (word) (void)hipHostFree(word);
  return AGX_OK;
}

extern "C" int agx_position_task_step_strict(const AgxStrictStepPlan *sp, const float *actions_in, int *drew, uint64_t *offset_after,
                                             void *stream) {
  AGX_REQUIRE(sp && sp->plan && sp->host_word && drew && offset_after, "null plan member");
  const AgxPositionStepPlan *plan = sp->plan;
  AGX_REQUIRE(plan->params && plan->buf && plan->task && plan->reset && plan->buf->reset_flag, "null plan member");
  AGX_REQUIRE(plan->buf->push_world == 0, "the strict step is a single-GPU path (no peer push bound)");
  AGX_REQUIRE(plan->reset->u_state, "strict step: the reset takes its draws from tensors (AgxResetArgs.u_*)");
  UniformSegs S;
  unsigned blocks = 0;
  uint64_t after = 0;
  if (int e = fill_plan(sp->count, sp->out, sp->numel, sp->offset, sp->sm_count, sp->max_threads_per_sm, S, &after, &blocks)) return e;
  plan->buf->flag_parity ^= 1;
  if (int e = agx_env_step(plan->params, plan->buf, plan->num_envs, actions_in, plan->k_substeps, plan->task, stream)) return e;
  const uint32_t seq = ((uint32_t)plan->buf->step_counter & 0x3FFFFFFFu) + 1u;
  hipLaunchKernelGGL(k_publish_flag, dim3(1), dim3(64), 0, (hipStream_t)stream, plan->buf->reset_flag, plan->buf->flag_parity, seq, sp->host_word);
  if (int e = check_launch("k_publish_flag")) return e;
  // the one bit the host needs before it may consume the generator: spin on the mapped word (the kernels above are in flight)
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t w;
  unsigned spins = 0;
  while (((w = __atomic_load_n(sp->host_word, __ATOMIC_ACQUIRE)) >> 1) != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFu) == 0u) {
      const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms > (sp->timeout_ms > 0 ? sp->timeout_ms : 10000)) {
        const hipError_t e = hipStreamQuery((hipStream_t)stream);
        return fail(AGX_E_LAUNCH, "strict step: the reset flag of step %u did not arrive within %d ms (stream: %s)", seq, (int)ms,
                    hipGetErrorString(e));
      }
    }
  }
  *drew = (int)(w & 1u);
  *offset_after = sp->offset;
  if (w & 1u) {  // the reference's `if len(env_ids) > 0`: the rand_like calls happen, all of them in one launch
    hipLaunchKernelGGL(k_torch_uniform_fill, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, (unsigned long long)sp->seed);
    if (int e = check_launch("k_torch_uniform_fill")) return e;
    *offset_after = after;
  }
  return agx_post_step_position(plan->params, plan->buf, plan->num_envs, plan->reset, plan->target, plan->obs, stream);
}
