// ABI bookkeeping: version + thread-local error string (no other global mutable state).
#include <stdarg.h>

#include "common.h"

namespace nlsh {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace nlsh

extern "C" int nlsh_abi_version(void) { return NLSH_ABI_VERSION; }
extern "C" const char *nlsh_last_error(void) { return nlsh::g_err; }
