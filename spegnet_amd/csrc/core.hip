// Error plumbing + version for libspegnet_hip.so
#include <stdarg.h>
#include "common.h"

namespace spg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SPG_ERR_LAUNCH;
  }
  return SPG_OK;
}
}  // namespace spg

// bumped with every change of an exported signature; spegnet_amd/_lib.py refuses a library whose number differs from the one its
// argument table was written for (a stale .so would receive shifted arguments)
extern "C" int spg_version(void) { return SPG_ABI_VERSION; }
extern "C" const char* spg_last_error(void) { return spg::g_err; }
