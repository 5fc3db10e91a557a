// Measured VALU issue ceiling of this GPU for plain (non-packed) fp32 vector instructions, in wave64 instructions per
// second: what the env-step and ray-cast kernels' SQ_INSTS_VALU rates are compared with (bench.py `roofline.valu`), next to
// the hardware guide's 2 cycles per wave64 v_fma_f32 per SIMD-32 at 2.4 GHz (1.229 T/s).
//
// Round 3: 16 INDEPENDENT chains per wave (a dependent v_fma chain issues every ~4 cycles, so 8
// chains at 2 cycles per instruction were exactly on the edge), the second and third operands a scalar register / an
// inline constant (one VGPR read port per instruction instead of three), long launches (~25 ms).  Every launch prints
// its wall time; run once more under `rocprofv3 --pmc GRBM_GUI_ACTIVE` and profiles/collect_valu_peak.py divides that
// counter by the wall time: the EFFECTIVE clock of the launch (the chip clocks to its power budget).
//   hipcc --offload-arch=gfx950 -O3 profiles/src/valu_peak.hip -o gpurun_out/valu_peak && gpurun_out/valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHAINS16(OP)                                                                                                       \
  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ void __launch_bounds__(256) k_valu(float *out, int iters, float a) {
  float x[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) x[c] = (float)threadIdx.x + (float)c;
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {  // 64 vector instructions per scalar loop-control triple
    if (KIND == 0) {  // v_fma_f32 v, v, s, 0.5
#define OP(c) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x[c]) : "s"(a));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 1) {  // VOP2 mix: v_mul_f32 v, s, v / v_add_f32 v, 1.0, v / v_max_f32 v, s, v / v_min_f32 v, s, v
#define OP(c)                                                                                               \
  if ((c & 3) == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[c]) : "s"(a));                              \
  else if ((c & 3) == 1) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x[c]));                                 \
  else if ((c & 3) == 2) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x[c]) : "s"(a));                         \
  else asm volatile("v_min_f32 %0, %1, %0" : "+v"(x[c]) : "s"(a));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 2) {  // three VGPR sources (round 2's probe): v_fma_f32 v, v, v, v
#define OP(c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(x[(c + 1) & 15]), "v"(x[(c + 2) & 15]));
      CHAINS16(OP)
#undef OP
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Round 4: what the OTHER instruction classes cost (the env-step kernels evaluate their elementary functions in float64, divide,
// take square roots and run Philox: SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 / TRANS_F32 / TRANS_F64 / INT32 ... per launch x these costs is
// the time the vector pipe is really busy -- an instruction count against the plain-fp32 ceiling undercounts it).  Same shape:
// 16 independent chains per wave, 8 waves per SIMD.  kPerOp = instructions per OP.
template <int KIND>
__global__ void __launch_bounds__(256) k_valu_class(float *out, int iters, float a, double ad, unsigned au) {
  float x[16];
  double y[16];
  unsigned z[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    x[c] = 1.0f + 0.001f * (float)(threadIdx.x + c);
    y[c] = 1.0 + 0.001 * (double)(threadIdx.x + c);
    z[c] = threadIdx.x * 16u + (unsigned)c + 1u;
  }
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    if (KIND == 10) {
#define OP(c) asm volatile("v_fma_f64 %0, %0, %1, 0.5" : "+v"(y[c]) : "s"(ad));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 11) {
#define OP(c)                                                                     \
  if (c & 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(y[c]) : "s"(ad));          \
  else asm volatile("v_add_f64 %0, %0, 0.5" : "+v"(y[c]));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 12) {
#define OP(c) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 13) {
#define OP(c) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[c]));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 14) {
#define OP(c) asm volatile("v_rcp_f64 %0, %0" : "+v"(y[c]));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 15) {
#define OP(c) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(z[c]) : "s"(au));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 16) {  // two instructions per OP
#define OP(c) asm volatile("v_cvt_f64_f32 %1, %0\n\tv_cvt_f32_f64 %0, %1" : "+v"(x[c]), "+v"(y[c]));
      CHAINS16(OP)
#undef OP
    } else if (KIND == 17) {
#define OP(c) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(z[c]) : "s"(au));
      CHAINS16(OP)
#undef OP
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += x[c] + (float)y[c] + (float)z[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, int waves_per_simd, int &launch_index) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  const int iters = 400000 / waves_per_simd;  // ~equal wall time per launch (tens of ms)
  float *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1000, 1.0001f);  // warm-up (dispatch launch_index)
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);  // timed (dispatch launch_index + 1)
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4 * iters * 16;
  const double rate = winst / (ms * 1e-3);
  const double cyc_at_max = ((double)cus * 4 * p.clockRate * 1e3) / rate;
  printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"cus\": %d, \"max_clock_khz\": %d, \"ms\": %.4f, \"timed_dispatch_index\": %d, "
         "\"wave_instr\": %.0f, \"wave_instr_per_s\": %.4e, \"cycles_per_wave_instr_at_max_clock\": %.3f}\n",
         name, waves_per_simd, cus, p.clockRate, ms, launch_index + 1, winst, rate, cyc_at_max);
  launch_index += 2;
  hipFree(out);
}

template <int KIND>
static void run_class(const char *name, const char *cls, int per_op, int &launch_index) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, waves_per_simd = 8;
  const int blocks = cus * waves_per_simd;
  const int iters = 12000;  // (the slow classes: a few ms per launch)
  float *out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_valu_class<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0001f, 1.0001, 2654435761u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_valu_class<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1.0001, 2654435761u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winst = (double)blocks * 4 * iters * 16 * per_op;
  const double rate = winst / (ms * 1e-3);
  const double cyc_at_max = ((double)cus * 4 * p.clockRate * 1e3) / rate;
  printf("{\"kind\": \"%s\", \"class\": \"%s\", \"waves_per_simd\": %d, \"cus\": %d, \"max_clock_khz\": %d, \"ms\": %.4f, "
         "\"timed_dispatch_index\": %d, \"wave_instr\": %.0f, \"wave_instr_per_s\": %.4e, \"cycles_per_wave_instr_at_max_clock\": %.3f}\n",
         name, cls, waves_per_simd, cus, p.clockRate, ms, launch_index + 1, winst, rate, cyc_at_max);
  launch_index += 2;
  hipFree(out);
}

int main() {
  int li = 0;
  for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32 v,v,s,const", w, li);
  for (int w : {4, 8}) run<1>("v_mul/add/max/min_f32 (VOP2, scalar or inline operand)", w, li);
  for (int w : {4, 8}) run<2>("v_fma_f32 v,v,v,v (three vector sources)", w, li);
  run_class<10>("v_fma_f64 v,v,s,const", "FMA_F64", 1, li);
  run_class<11>("v_mul_f64 / v_add_f64", "MUL_F64+ADD_F64", 1, li);
  run_class<12>("v_rcp_f32", "TRANS_F32", 1, li);
  run_class<13>("v_sqrt_f32", "TRANS_F32", 1, li);
  run_class<14>("v_rcp_f64", "TRANS_F64", 1, li);
  run_class<15>("v_mul_lo_u32", "INT32 (multiply)", 1, li);
  run_class<17>("v_mul_hi_u32", "INT32 (multiply)", 1, li);
  run_class<16>("v_cvt_f64_f32 + v_cvt_f32_f64", "CVT", 2, li);
  return 0;
}
