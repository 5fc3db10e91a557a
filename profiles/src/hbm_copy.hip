// HBM streaming rates of this box, float4 accesses, several kernel shapes: which copy kernel the "achievable HBM" line of
// bench.py (agx_copy_f4) should be, and what read-only / write-only streams reach.  MI355X_MICROARCH.md: 8.0 TB/s spec,
// 6.29 TB/s measured for a float4 copy.
//   hipcc --offload-arch=gfx950 -O3 profiles/src/hbm_copy.hip -o gpurun_out/hbm_copy && gpurun_out/hbm_copy
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void __launch_bounds__(256) k_one(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}
__global__ void __launch_bounds__(256) k_stride(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) d[i] = s[i];
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_unroll(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * st < n; i += U * st) {
    typedef float vf4 __attribute__((ext_vector_type(4)));
    vf4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = NT ? __builtin_nontemporal_load((const vf4 *)(s + i + u * st)) : *(const vf4 *)(s + i + u * st);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], (vf4 *)(d + i + u * st));
      else *(vf4 *)(d + i + u * st) = v[u];
    }
  }
  for (; i < n; i += st) d[i] = s[i];
}
__global__ void __launch_bounds__(256) k_read(const float4 *__restrict__ s, float *__restrict__ out, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) {
    const float4 v = s[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(float4 *__restrict__ d, size_t n) {
  const size_t st = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) d[i] = float4{1.0f, 2.0f, 3.0f, 4.0f};
}

template <class F>
static void timeit(const char *name, double bytes, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("{\"kernel\": \"%s\", \"GBps\": %.1f}\n", name, bytes * 10 / (best * 1e-3) / 1e9);
}

int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  float4 *s, *d;
  float *o;
  hipMalloc(&s, bytes);
  hipMalloc(&d, bytes);
  hipMalloc(&o, 64);
  hipMemset(s, 1, bytes);
  timeit("one float4 per thread", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_one, dim3((unsigned)(n / 256)), dim3(256), 0, 0, s, d, n); });
  for (int bpc : {8, 16, 32, 64}) {
    char nm[96];
    snprintf(nm, sizeof nm, "grid-stride, %d workgroups per CU", bpc);
    timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(k_stride, dim3(256 * bpc), dim3(256), 0, 0, s, d, n); });
  }
  for (int bpc : {8, 32}) {
    char nm[96];
    snprintf(nm, sizeof nm, "grid-stride x4 in flight, %d workgroups per CU", bpc);
    timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_unroll<4, false>), dim3(256 * bpc), dim3(256), 0, 0, s, d, n); });
    snprintf(nm, sizeof nm, "grid-stride x8 in flight, %d workgroups per CU", bpc);
    timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_unroll<8, false>), dim3(256 * bpc), dim3(256), 0, 0, s, d, n); });
    snprintf(nm, sizeof nm, "grid-stride x4 in flight, nontemporal, %d workgroups per CU", bpc);
    timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((k_unroll<4, true>), dim3(256 * bpc), dim3(256), 0, 0, s, d, n); });
  }
  timeit("read only (grid-stride, 32 per CU)", 1.0 * bytes, [&] { hipLaunchKernelGGL(k_read, dim3(256 * 32), dim3(256), 0, 0, s, o, n); });
  timeit("write only (grid-stride, 32 per CU)", 1.0 * bytes, [&] { hipLaunchKernelGGL(k_write, dim3(256 * 32), dim3(256), 0, 0, d, n); });
  timeit("hipMemcpyDtoD", 2.0 * bytes, [&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
