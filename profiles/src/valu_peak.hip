// Measured VALU issue ceiling of this GPU for plain (non-packed) fp32 vector instructions, in wave64
// instructions per second: the roofline the ray-cast and env-step kernels are priced against (bench.py
// `roofline.bound = "valu"`).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 profiles/src/valu_peak.hip -o gpurun_out/valu_peak && gpurun_out/valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(256) k_valu(float *out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                   : "v"(a), "v"(b));
    } else if (KIND == 1) {
      asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                   "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                   : "v"(a), "v"(b));
    } else {
      asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_lt_f32 vcc, %1, %9\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_gt_f32 vcc, %3, %9\n"
                   "v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_lt_f32 vcc, %5, %9\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_gt_f32 vcc, %7, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                   : "v"(a), "v"(b)
                   : "vcc");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int KIND>
static double run(const char *name, int waves_per_simd) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 20000;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  float *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4 * iters * 8;
  const double rate = winst / (ms * 1e-3);
  const double per_simd_clk = rate / ((double)cus * 4 * p.clockRate * 1e3);
  printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"cus\": %d, \"clock_khz\": %d, \"wave_instr_per_s\": %.4e, \"cycles_per_wave_instr\": %.3f}\n",
         name, waves_per_simd, cus, p.clockRate, rate, 1.0 / per_simd_clk);
  hipFree(out);
  return rate;
}

int main() {
  for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32", w);
  for (int w : {4, 8}) run<1>("v_mul/add/min/max_f32", w);
  for (int w : {4, 8}) run<2>("v_cndmask/v_cmp", w);
  return 0;
}
