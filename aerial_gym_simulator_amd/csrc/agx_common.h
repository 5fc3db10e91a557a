// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/aerial_gym_hip.h"

namespace agx {

// thread-local last error text (agx_last_error)
char *error_buffer();
int fail(int code, const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGX_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return AGX_OK;
}

// process-wide options (agx_set_option, agx_api.cpp); read per launch
int option_env_step_quad();
int option_ray_split();

inline int blocks_for(int n, int block) { return (n + block - 1) / block; }

// Small batches are latency bound: spread them over as many CUs as possible
// (one wave per workgroup); big batches use 256-thread workgroups.
inline int pick_block(int n) { return n <= 65536 ? 64 : 256; }

}  // namespace agx

#define AGX_REQUIRE(cond, ...)                              \
  do {                                                      \
    if (!(cond)) return agx::fail(AGX_E_ARG, __VA_ARGS__);  \
  } while (0)
