// Error reporting + ABI version for libcim_hip.so.
#include "common.h"
#include "../../include/cim_hip.h"
#include <stdarg.h>

namespace cim {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cim

extern "C" const char* cim_last_error(void) { return cim::g_err; }
extern "C" int cim_abi_version(void) { return 15; }
