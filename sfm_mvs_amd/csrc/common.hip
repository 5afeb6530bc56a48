// Error channel + ABI version of libsfmhip.so.
#include "common.h"
#include <cstring>

namespace sfm {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sfm

extern "C" int sfm_abi_version(void) { return SFM_ABI_VERSION; }
extern "C" const char* sfm_last_error(void) { return sfm::g_err; }
