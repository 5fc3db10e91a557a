// Error plumbing of the C ABI (no device code here).
#include "agx_common.h"

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
}  // namespace agx

extern "C" const char *agx_last_error(void) { return agx::error_buffer(); }
extern "C" int agx_abi_version(void) { return AGX_ABI_VERSION; }
#ifndef AGX_BUILD_ID
#error "AGX_BUILD_ID must be defined by the build (aerial_gym_simulator_amd/_build.py: hash of the sources and flags)"
#endif
// (tagged, so that the id can be read out of the FILE without mapping a possibly stale library into the process that is about
// to rebuild and load it: _build.binary_build_id)
static const char kBuildTag[] = "agx-build-id:" AGX_BUILD_ID;
extern "C" const char *agx_build_id(void) { return kBuildTag + 13; }
