// Error state and version of the C ABI declared in include/efm_hip.h.
#include "efm_common.h"

namespace efm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace efm

extern "C" {
int efm_version(void) { return EFM_ABI_VERSION; }
const char* efm_last_error_string(void) { return efm::g_err; }
}
