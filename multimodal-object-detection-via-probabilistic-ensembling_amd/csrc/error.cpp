// pe_last_error / pe_version (host only).
#include <stdarg.h>
#include <stdio.h>

#include "proben_hip.h"

namespace pe {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace pe

extern "C" const char* pe_last_error(void) { return pe::g_err; }
extern "C" int pe_version(void) { return 1; }
