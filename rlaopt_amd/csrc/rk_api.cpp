// Error state and version of librk_hip.so (see include/rk.h for the conventions).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/rk.h"

namespace rk {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = '\0'; }

}  // namespace rk

extern "C" const char* rk_last_error(void) { return rk::g_err; }
extern "C" int rk_abi_version(void) { return 15; }
