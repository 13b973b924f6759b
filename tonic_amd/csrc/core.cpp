// Library-level entry points: error reporting and version queries.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tonic_hip.h"

namespace tonic {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

}  // namespace tonic

extern "C" const char* tonic_last_error(void) { return tonic::g_error; }
// 2: off-policy parameter blocks use the padded layout (tonic_mlp_weight_stride)
// 3: pinned-host collector (tonic_collector_*)
extern "C" int32_t tonic_abi_version(void) { return TONIC_ABI_VERSION; }
extern "C" const char* tonic_target_arch(void) { return "gfx950"; }
