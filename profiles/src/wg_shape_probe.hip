#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float v[352]; };
__global__ void k_empty(Big big, float *out) { if (big.v[0] == 123.0f) out[threadIdx.x] = big.v[1]; }
#define F1 asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define F8 F1 F1 F1 F1 F1 F1 F1 F1
__global__ void k_work(Big big, float *out, int iters, float a, float b) {
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) { F8 }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x + big.v[0];
}
template <class F> static float time_us(F launch, int reps = 400) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize(); hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / reps;
}
int main() {
  float *out; hipMalloc(&out, 1 << 24); Big big{};
  int cfg[][2] = {{128, 64}, {512, 64}, {256, 128}, {128, 256}, {64, 512}, {32, 1024}, {2048, 64}, {512, 256}};
  for (auto &c : cfg) {
    float e = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(c[0]), dim3(c[1]), 0, 0, big, out); });
    float w = time_us([&] { hipLaunchKernelGGL(k_work, dim3(c[0]), dim3(c[1]), 0, 0, big, out, 128, 1.0001f, 0.5f); });
    printf("{\"workgroups\": %d, \"threads\": %d, \"empty_us\": %.2f, \"work1024_us\": %.2f}\n", c[0], c[1], e, w);
  }
  return 0;
}
