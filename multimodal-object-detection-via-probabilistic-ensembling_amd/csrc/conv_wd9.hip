// Dispatch of the persistent one-wave-per-SIMD weights-direct kernels (csrc/conv_wd9.h).  No C-ABI of its own: the entry points of
// csrc/conv_wd.hip (pe_conv3x3_wd_f16, ...) hand their launches over when the geometry and the launch size qualify, and both kernel
// generations produce the same bits, so the choice never shows in results.
//
// Built with -fno-slp-vectorize (the epilogue's scalar fp32 adds must not be packed into v_pk_add_f32 + shuffles) and
// -mllvm -amdgpu-spill-vgpr-to-agpr=0 (the AGPRs a[0:255] belong to the asm statements: the compiler must never park a VGPR there);
// tests/test_build_audit.py checks the emitted code for scratch use and for accumulator-file instructions outside the asm blocks.
#include "conv_wd9.h"

namespace pe {
// 0 = never, 1 = when the launch has at least kWd9MinTiles tiles (default), 2 = whenever the geometry allows (tests)
int g_wd9_mode = 1;
constexpr int kWd9MinTiles = 128;

static bool wd9_takes(int H, int W, long long M, int Cout) {
    if (g_wd9_mode == 0 || !wd9::geometry_ok(H, W, 8) || W < 64) return false;
    const long long tiles = (long long)ceil_div(M, 256) * (Cout / 256);
    return g_wd9_mode == 2 || tiles >= kWd9MinTiles;
}

// pure 3x3 (+ bias, optional ReLU): PE_OK when launched, PE_ERR_UNSUPPORTED when the caller should use conv_wd.h's kernel
int wd9_conv3x3(ConvWdArgs a, hipStream_t st) {
    if (!wd9_takes(a.H, a.W, a.M, a.Cout)) return PE_ERR_UNSUPPORTED;
    return wd9::launch<8, 4, 5>(a, st);
}
}  // namespace pe

// measurement hook (csrc/test_hooks.h): 1 when a 3x3 launch of this shape runs on the conv_wd9.h kernel
extern "C" int pe_test_wd9_takes(int N, int H, int W, int Cin, int Cout) {
    (void)Cin;
    return pe::wd9_takes(H, W, (long long)N * H * W, Cout) ? 1 : 0;
}

// test hook (csrc/test_hooks.h)
extern "C" int pe_test_set_wd9_mode(int mode) {
    pe::g_wd9_mode = mode;
    return PE_OK;
}
