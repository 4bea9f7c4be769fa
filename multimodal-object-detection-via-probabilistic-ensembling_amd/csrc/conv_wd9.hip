// Dispatch of the persistent one-wave-per-SIMD weights-direct kernels (csrc/conv_wd9.h).  No C-ABI of its own: the entry points of
// csrc/conv_wd.hip (pe_conv3x3_wd_f16, ...) hand their launches over when the geometry and the launch size qualify, and both kernel
// generations produce the same bits, so the choice never shows in results.
//
// Built with -fno-slp-vectorize (the epilogue's scalar fp32 adds must not be packed into v_pk_add_f32 + shuffles) and
// -mllvm -amdgpu-spill-vgpr-to-agpr=0 (the AGPRs a[0:255] belong to the asm statements: the compiler must never park a VGPR there);
// tests/test_build_audit.py checks the emitted code for scratch use and for accumulator-file instructions outside the asm blocks.
#include <atomic>

#include "conv_wd9.h"

namespace pe {
// bit 0: the pure 3x3 kernel takes launches of at least kWd9MinTiles tiles; bit 1: ... whenever the geometry allows (tests);
// bit 3: the fused RPN head (3x3 + ReLU + 1x1 to 16 columns) runs on conv_wd9.h's head epilogue under the pure kernel's size rule
// (same bits as conv_wd.h's).  Default 1 | 8: worth +1 % / +0.4 % in every pipeline of bench.py (profiles/r04_pipeline_ab_*.txt).
// (Bit 2 was round 4's fused bottleneck tail on this structure: 8-14 % faster as a launch of its own, -1.3 % in the frame-pair
// pipeline - a 512-register / 160-KiB workgroup owns its CU and the other detector's kernels can no longer co-reside - and different
// bits from the two-wave tail, so no shape rule could ever select it.  It left the library in round 5: scripts/lab/conv_wd9_tail.h,
// profiles/r05_pipeline_ab_wd9tail.txt.)
constexpr int kWd9ModeDefault = 1 | 8;
std::atomic<int> g_wd9_mode{kWd9ModeDefault};
// workgroups of the persistent kernels: one per CU.  Fewer (leaving CUs to the other detector's stream) measured worse in every
// two-stream pipeline (profiles/r04_pipeline_ab_2.txt), so the streams hint below does not change it.
std::atomic<int> g_wd9_wgs{256};
constexpr int kWd9MinTiles = 128;

static bool wd9_takes(int H, int W, long long M, int Cout) {
    const int mode = g_wd9_mode.load(std::memory_order_relaxed);
    if (!(mode & 3) || !wd9::geometry_ok(H, W, 8) || W < 64) return false;
    const long long tiles = (long long)ceil_div(M, 256) * (Cout / 256);
    return (mode & 2) || tiles >= kWd9MinTiles;
}

// pure 3x3 (+ bias, optional ReLU): PE_OK when launched, PE_ERR_UNSUPPORTED when the caller should use conv_wd.h's kernel
int wd9_conv3x3(ConvWdArgs a, hipStream_t st) {
    if (!wd9_takes(a.H, a.W, a.M, a.Cout)) return PE_ERR_UNSUPPORTED;
    return wd9::launch<8, 4, 5>(a, st, g_wd9_wgs.load(std::memory_order_relaxed));
}

// fused RPN head: as the pure kernel (bit-identical to conv_wd.h's HEAD == 1, so the size rule may look at the batch)
static bool wd9_head_takes(int H, int W, long long M) {
    const int mode = g_wd9_mode.load(std::memory_order_relaxed);
    if (!(mode & 8) || !wd9::geometry_ok(H, W, 8) || W < 64) return false;
    return (mode & 2) || ceil_div(M, 256) >= kWd9MinTiles;
}

int wd9_rpn_head(ConvWdArgs a, hipStream_t st) {
    if (!wd9_head_takes(a.H, a.W, a.M)) return PE_ERR_UNSUPPORTED;
    return wd9::launch_head<8, 4, 9>(a, st, g_wd9_wgs.load(std::memory_order_relaxed));
}

}  // namespace pe

extern "C" int pe_test_set_wd9_wgs(int pure, int tail) {
    (void)tail;
    auto clamp = [](int n) { return n < 8 ? 8 : (n > 256 ? 256 : n / 8 * 8); };
    if (pure > 0) pe::g_wd9_wgs = clamp(pure);
    return PE_OK;
}

// measurement hook (csrc/test_hooks.h): 1 when a 3x3 launch of this shape runs on the conv_wd9.h kernel
extern "C" int pe_test_wd9_takes(int N, int H, int W, int Cin, int Cout) {
    (void)Cin;
    return pe::wd9_takes(H, W, (long long)N * H * W, Cout) ? 1 : 0;
}

extern "C" int pe_test_wd9_head_takes(int N, int H, int W) { return pe::wd9_head_takes(H, W, (long long)N * H * W) ? 1 : 0; }

// test hook (csrc/test_hooks.h)
extern "C" int pe_test_set_wd9_mode(int mode) {
    pe::g_wd9_mode = mode < 0 ? pe::kWd9ModeDefault : mode;
    return PE_OK;
}
