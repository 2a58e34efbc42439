// capi.hip -- error reporting and ABI version of libbevfusion_hip.
#include "common.h"

namespace bfhip {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace bfhip

BFHIP_EXPORT int bfhip_abi_version(void) { return 1; }
BFHIP_EXPORT const char *bfhip_last_error(void) { return bfhip::g_err; }
