// agx_math_eval: the elementary functions of the dynamics path (agx_device_math.h) evaluated on the device for a vector
// of arguments.  Diagnostic entry of the C ABI: the parity tests check device == CPU restatement bit for bit on
// millions of arguments (the dynamics kernels inline exactly these functions).
#include "agx_common.h"
#include "agx_device_math.h"

namespace {
using namespace agx;

__global__ void k_math_eval(int which, int n, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s, c;
  switch (which) {
    case AGX_MATH_SIN: sincos_bounded(x[i], s, c); out[i] = s; break;
    case AGX_MATH_COS: sincos_bounded(x[i], s, c); out[i] = c; break;
    case AGX_MATH_ATAN2: out[i] = atan2_cw(x[i], y[i]); break;
    case AGX_MATH_ASIN: out[i] = asin_cw(x[i]); break;
    default: out[i] = exp_cw(x[i]); break;
  }
}
// float4 streaming copy: the HBM rate a kernel of this library can reach on this device (the "achievable" line of the
// roofline; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy against 8 TB/s of specification)
// One float4 per thread, no loop: the shape that reaches the guide's figure on this box (6.2 TB/s; grid-stride loops with 8-64
// workgroups per CU stay at 4.4-5.2 TB/s, hipMemcpyDtoD at 4.8: profiles/src/hbm_copy.hip, profiles/r03_hbm_copy.jsonl).
__global__ void __launch_bounds__(256) k_copy_f4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) dst[i] = src[i];
}
}  // namespace

extern "C" int agx_copy_f4(const void *src, void *dst, size_t bytes, void *stream) {
  AGX_REQUIRE(src && dst && bytes % 16 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "16-byte aligned buffers required");
  const size_t n4 = bytes / 16;
  if (n4 == 0) return AGX_OK;
  const size_t blocks = (n4 + 255) / 256;
  AGX_REQUIRE(blocks <= 0x7FFFFFFFull, "at most 2^31 - 1 workgroups (512 GiB)");
  hipLaunchKernelGGL(k_copy_f4, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)src, (float4 *)dst, n4);
  return agx::check_launch("agx_copy_f4");
}

extern "C" int agx_math_eval(int which, int n, const float *x, const float *y, float *out, void *stream) {
  AGX_REQUIRE(which >= AGX_MATH_SIN && which <= AGX_MATH_EXP, "which = %d: not an AGX_MATH_* id", which);
  AGX_REQUIRE(n >= 0, "n = %d", n);
  AGX_REQUIRE(n == 0 || (x && out && (which != AGX_MATH_ATAN2 || y)), "null buffer");
  if (n == 0) return AGX_OK;
  hipLaunchKernelGGL(k_math_eval, dim3(agx::blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, which, n, x, y, out);
  return agx::check_launch("agx_math_eval");
}
