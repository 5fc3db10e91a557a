// What bounds a straight-line kernel of a few thousand instructions on 128 one-wave workgroups (the shape of k_env_step at
// 8192 envs)?  Launch interval (back to back, HIP events) of
//   empty      no work (dispatch floor), with a 1.4 KB by-value argument like k_env_step's
//   loop N     N v_fma_f32 as 8 instructions x N/8 iterations  (instruction cache hot after the first iteration)
//   line N     N v_fma_f32 as straight-line code               (8 bytes each: every 64-byte line is fetched once, cold)
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 profiles/src/ifetch_probe.hip -o gpurun_out/ifetch_probe && gpurun_out/ifetch_probe
#include <hip/hip_runtime.h>

#include <cstdio>

struct Big {
  float v[352];
};

#define F1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define F8 F1 F1 F1 F1 F1 F1 F1 F1
#define F64 F8 F8 F8 F8 F8 F8 F8 F8
#define F512 F64 F64 F64 F64 F64 F64 F64 F64
#define F2048 F512 F512 F512 F512

__global__ void __launch_bounds__(64) k_empty(Big big, float *out) {
  if (big.v[0] == 123.0f) out[threadIdx.x] = big.v[1];
}
__global__ void __launch_bounds__(64) k_loop(Big big, float *out, int iters, float a, float b) {
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) { F8 }
  out[blockIdx.x * 64 + threadIdx.x] = x + big.v[0];
}
template <int K>
__global__ void __launch_bounds__(64) k_line(Big big, float *out, float a, float b) {
  float x = threadIdx.x;
  F2048
  if (K >= 2) { F2048 }
  if (K >= 4) { F2048 F2048 }
  out[blockIdx.x * 64 + threadIdx.x] = x + big.v[0];
}
// two waves per workgroup running DIFFERENT straight-line halves (the shape of a role-split kernel)
__global__ void __launch_bounds__(128) k_split(Big big, float *out, float a, float b) {
  float x = threadIdx.x;
  if (threadIdx.x < 64) { F2048 } else { F2048 }
  out[blockIdx.x * 128 + threadIdx.x] = x + big.v[0];
}

template <class F>
static float time_us(F launch, int reps = 400) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  float *out;
  hipMalloc(&out, 1 << 24);
  Big big{};
  for (int wgs : {128, 512, 1024}) {
    printf("{\"workgroups\": %d", wgs);
    printf(", \"empty_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(64), 0, 0, big, out); }));
    printf(", \"loop2048_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_loop, dim3(wgs), dim3(64), 0, 0, big, out, 256, 1.0001f, 0.5f); }));
    printf(", \"loop4096_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_loop, dim3(wgs), dim3(64), 0, 0, big, out, 512, 1.0001f, 0.5f); }));
    printf(", \"line2048_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_line<1>, dim3(wgs), dim3(64), 0, 0, big, out, 1.0001f, 0.5f); }));
    printf(", \"line4096_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_line<2>, dim3(wgs), dim3(64), 0, 0, big, out, 1.0001f, 0.5f); }));
    printf(", \"line8192_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_line<4>, dim3(wgs), dim3(64), 0, 0, big, out, 1.0001f, 0.5f); }));
    printf(", \"split_2x2048_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_split, dim3(wgs), dim3(128), 0, 0, big, out, 1.0001f, 0.5f); }));
    printf("}\n");
  }
  return 0;
}
