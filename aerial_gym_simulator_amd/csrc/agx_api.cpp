// Error plumbing of the C ABI (no device code here).
#include "agx_common.h"

#include <atomic>
#include <cstring>

namespace agx {
char *error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}
static std::atomic<int> g_env_step_quad{1}, g_ray_split{0};
int option_env_step_quad() { return g_env_step_quad.load(std::memory_order_relaxed); }
int option_ray_split() { return g_ray_split.load(std::memory_order_relaxed); }
static std::atomic<int> *option_slot(const char *name, int *lo, int *hi) {
  if (name && !strcmp(name, "env_step_quad")) { *lo = 0; *hi = 1; return &g_env_step_quad; }
  if (name && !strcmp(name, "ray_split")) { *lo = 0; *hi = 1 << 20; return &g_ray_split; }
  return nullptr;
}
}  // namespace agx

extern "C" int agx_set_option(const char *name, int value) {
  int lo, hi;
  std::atomic<int> *slot = agx::option_slot(name, &lo, &hi);
  AGX_REQUIRE(slot, "agx_set_option: unknown option '%s' (env_step_quad, ray_split)", name ? name : "(null)");
  AGX_REQUIRE(value >= lo && value <= hi, "agx_set_option: %s = %d outside [%d, %d]", name, value, lo, hi);
  slot->store(value, std::memory_order_relaxed);
  return AGX_OK;
}
extern "C" int agx_get_option(const char *name, int *value) {
  int lo, hi;
  std::atomic<int> *slot = agx::option_slot(name, &lo, &hi);
  AGX_REQUIRE(slot && value, "agx_get_option: unknown option '%s' or null result", name ? name : "(null)");
  *value = slot->load(std::memory_order_relaxed);
  return AGX_OK;
}

extern "C" const char *agx_last_error(void) { return agx::error_buffer(); }
extern "C" int agx_abi_version(void) { return AGX_ABI_VERSION; }
#ifndef AGX_BUILD_ID
#error "AGX_BUILD_ID must be defined by the build (aerial_gym_simulator_amd/_build.py: hash of the sources and flags)"
#endif
// (tagged, so that the id can be read out of the FILE without mapping a possibly stale library into the process that is about
// to rebuild and load it: _build.binary_build_id)
static const char kBuildTag[] = "agx-build-id:" AGX_BUILD_ID;
extern "C" const char *agx_build_id(void) { return kBuildTag + 13; }
