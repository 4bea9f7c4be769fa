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

// Opt a kernel in to `bytes` of dynamic LDS (> the 64 KiB default).  The attribute belongs to the (function, device) pair: it is
// applied once per pair (not once per process under whichever device happened to be current), from any thread, and its status
// is checked - on a part that cannot grant the request the launch path returns PE_ERR_HIP instead of failing at the launch.
int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what);

#define PE_CHECK_ARG(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            pe::set_error(__VA_ARGS__);          \
            return PE_ERR_INVALID_ARG;           \
        }                                        \
    } while (0)

#define PE_ENSURE_LDS(kernel, bytes, name)                                                     \
    do {                                                                                       \
        const int st_ = pe::ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), bytes, name); \
        if (st_ != PE_OK) return st_;                                                          \
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
// ReLU / max-pool maxima propagate NaN like torch's relu_ and max_pool2d (fmaxf / v_max_f32 return the OTHER operand for a
// NaN and would turn a poisoned activation into a clean 0: the reference's robustness contract - NaN / inf features give no
// proposals and no detections, tests/test_model_e2e.py:91-120 - depends on NaN surviving).  gfx950: one v_maximum3_f32.
__device__ __forceinline__ float relu_nan(float x) { return __builtin_elementwise_maximum(x, 0.f); }

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace pe
