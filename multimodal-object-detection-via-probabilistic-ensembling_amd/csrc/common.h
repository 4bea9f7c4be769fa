// Shared helpers for the gfx950 kernels and their C-ABI wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "proben_hip.h"

namespace pe {

void set_error(const char* fmt, ...);

#define PE_CHECK_ARG(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            pe::set_error(__VA_ARGS__);          \
            return PE_ERR_INVALID_ARG;           \
        }                                        \
    } while (0)

#define PE_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            pe::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return PE_ERR_HIP;                                                     \
        }                                                                          \
    } while (0)

constexpr int kWave = 64;  // gfx950 wavefront width

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << (threadIdx.x & 63)) - 1ull;
}
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace pe
