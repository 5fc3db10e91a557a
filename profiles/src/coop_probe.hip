// Cost of a hardware / runtime grid barrier: hipLaunchCooperativeKernel + cooperative_groups::grid_group::sync() on 128
// one-wave workgroups, back to back, against the same work as a plain launch.
//   hipcc --offload-arch=gfx950 -O3 -w profiles/src/coop_probe.hip -o gpurun_out/coop_probe && gpurun_out/coop_probe
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
namespace cg = cooperative_groups;

#define F1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define F8 F1 F1 F1 F1 F1 F1 F1 F1

__global__ void __launch_bounds__(64) k_plain(float *out, int iters, float a, float b) {
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) { F8 }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
__global__ void __launch_bounds__(64) k_coop(float *out, int iters, int iters2, float a, float b, int syncs) {
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) { F8 }
  cg::grid_group g = cg::this_grid();
  for (int s = 0; s < syncs; ++s) g.sync();
  for (int i = 0; i < iters2; ++i) { F8 }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <class F>
static float time_us(F launch, int reps = 300) {
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
  hipMalloc(&out, 1 << 22);
  float a = 1.0001f, b = 0.5f;
  int wgs = 128;
  auto coop = [&](int it1, int it2, int syncs) {
    void *args[] = {&out, &it1, &it2, &a, &b, &syncs};
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_coop), dim3(wgs), dim3(64), args, 0, 0);
    if (e != hipSuccess) { printf("coop launch: %s\n", hipGetErrorString(e)); exit(1); }
  };
  printf("{\"workgroups\": %d", wgs);
  printf(", \"plain_empty_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_plain, dim3(wgs), dim3(64), 0, 0, out, 0, a, b); }));
  printf(", \"plain_2048_us\": %.2f", time_us([&] { hipLaunchKernelGGL(k_plain, dim3(wgs), dim3(64), 0, 0, out, 256, a, b); }));
  printf(", \"coop_empty_nosync_us\": %.2f", time_us([&] { coop(0, 0, 0); }));
  printf(", \"coop_empty_1sync_us\": %.2f", time_us([&] { coop(0, 0, 1); }));
  printf(", \"coop_empty_5sync_us\": %.2f", time_us([&] { coop(0, 0, 5); }));
  printf(", \"coop_2048_sync_512_us\": %.2f", time_us([&] { coop(256, 64, 1); }));
  printf("}\n");
  return 0;
}
