// pe_last_error / pe_version (host only).
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

#include "proben_hip.h"

namespace pe {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what) {
    if (bytes <= 64 * 1024) return PE_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error("%s: hipGetDevice failed", what);
        return PE_ERR_HIP;
    }
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;   // (kernel, device) -> largest size applied
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = granted[std::make_pair(kernel, dev)];
    if (have >= bytes) return PE_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("%s: %zu bytes of dynamic LDS refused on device %d: %s", what, bytes, dev, hipGetErrorString(e));
        return PE_ERR_HIP;
    }
    have = bytes;
    return PE_OK;
}
}  // namespace pe

extern "C" const char* pe_last_error(void) { return pe::g_err; }
extern "C" int pe_version(void) { return 1; }
