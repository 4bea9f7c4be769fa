// Dispatch of the persistent one-wave-per-SIMD weights-direct kernels (csrc/conv_wd9.h).  No C-ABI of its own: the entry points of
// csrc/conv_wd.hip (pe_conv3x3_wd_f16, ...) hand their launches over when the geometry and the launch size qualify, and both kernel
// generations produce the same bits, so the choice never shows in results.
//
// Built with -fno-slp-vectorize (the epilogue's scalar fp32 adds must not be packed into v_pk_add_f32 + shuffles) and
// -mllvm -amdgpu-spill-vgpr-to-agpr=0 (the AGPRs a[0:255] belong to the asm statements: the compiler must never park a VGPR there);
// tests/test_build_audit.py checks the emitted code for scratch use and for accumulator-file instructions outside the asm blocks.
#include "conv_wd9.h"
#include "conv_wd9_tail.h"

namespace pe {
// bit 0: the pure 3x3 kernel takes launches of at least kWd9MinTiles tiles; bit 1: ... whenever the geometry allows (tests);
// bit 2: the fused bottleneck tail of image width 64 runs on conv_wd9_tail.h; bit 3: the fused RPN head (3x3 + ReLU + 1x1 to 16 columns)
// runs on conv_wd9.h's head epilogue under the pure kernel's size rule (same bits as conv_wd.h's).
// Default 1: the pure kernel is worth +1 % in every pipeline of bench.py (profiles/r04_pipeline_ab_*.txt).  The tail kernel is 8-14 %
// faster than the two-wave tail as a launch of its own at batch 32, but a 512-register / 160-KiB workgroup owns its CU: the other
// detector's kernels can no longer co-reside with it, and whole frame-pair pipelines measure -0.5 .. -5 % with it (two R101 detectors on
// two streams: -3 % at 256 workgroups, break-even at 128; thermal-only batch 16: -3 %; three detectors: -5 %).  It is therefore opt-in
// (pe_test_set_wd9_mode(5), `bench.py --wd9-mode 5`), like a cuDNN algorithm that wins its own benchmark and loses the network's.
constexpr int kWd9ModeDefault = 1 | 8;
int g_wd9_mode = kWd9ModeDefault;
// workgroups of the persistent kernels (pure 3x3, fused tail): one per CU when a launch has the chip to itself; the two-detector
// pipeline runs the detectors on two streams, and a kernel that occupies every CU for its whole duration shuts the other stream out
int g_wd9_wgs = 256, g_wd9_tail_wgs = 256;
constexpr int kWd9MinTiles = 128;

static bool wd9_takes(int H, int W, long long M, int Cout) {
    if (!(g_wd9_mode & 3) || !wd9::geometry_ok(H, W, 8) || W < 64) return false;
    const long long tiles = (long long)ceil_div(M, 256) * (Cout / 256);
    return (g_wd9_mode & 2) || tiles >= kWd9MinTiles;
}

// pure 3x3 (+ bias, optional ReLU): PE_OK when launched, PE_ERR_UNSUPPORTED when the caller should use conv_wd.h's kernel
int wd9_conv3x3(ConvWdArgs a, hipStream_t st) {
    if (!wd9_takes(a.H, a.W, a.M, a.Cout)) return PE_ERR_UNSUPPORTED;
    return wd9::launch<8, 4, 5>(a, st, g_wd9_wgs);
}

// fused RPN head: as the pure kernel (bit-identical to conv_wd.h's HEAD == 1, so the size rule may look at the batch)
static bool wd9_head_takes(int H, int W, long long M) {
    if (!(g_wd9_mode & 8) || !wd9::geometry_ok(H, W, 8) || W < 64) return false;
    return (g_wd9_mode & 2) || ceil_div(M, 256) >= kWd9MinTiles;
}

int wd9_rpn_head(ConvWdArgs a, hipStream_t st) {
    if (!wd9_head_takes(a.H, a.W, a.M)) return PE_ERR_UNSUPPORTED;
    return wd9::launch_head<8, 4, 9>(a, st, g_wd9_wgs);
}

// fused bottleneck tail: the kernel is chosen by GEOMETRY only (image width 64 = res4 of an 800 x 1024 padded input) - the two
// generations add the shortcut at different points of the sum, so a batch-size-dependent choice would show in the results
static bool wd9_tail_takes(int H, int W, int Cin, int tail_cout) { return (g_wd9_mode & 4) && wd9t::geometry_ok(H, W, Cin, tail_cout); }

int wd9_bottleneck_tail(ConvWdArgs a, hipStream_t st) {
    if (!wd9_tail_takes(a.H, a.W, a.Cin, a.tail_cout)) return PE_ERR_UNSUPPORTED;
    // start skew 8 k cycles: measured -3.5 % on the long pole's cycles, ~-2 % wall (profiles/r04_wd9_tail_probe_4.txt)
    return wd9t::launch<4, 4>(a, st, g_wd9_tail_wgs, nullptr, 8000);
}
}  // namespace pe

extern "C" int pe_conv_wd_set_concurrent_streams(int32_t streams) {
    PE_CHECK_ARG(streams >= 1 && streams <= 8, "pe_conv_wd_set_concurrent_streams: streams must be in 1 .. 8 (got %d)", streams);
    pe::g_wd9_tail_wgs = 256 / streams / 8 * 8;
    return PE_OK;
}

extern "C" int pe_test_set_wd9_wgs(int pure, int tail) {
    auto clamp = [](int n) { return n < 8 ? 8 : (n > 256 ? 256 : n / 8 * 8); };
    if (pure > 0) pe::g_wd9_wgs = clamp(pure);
    if (tail > 0) pe::g_wd9_tail_wgs = clamp(tail);
    return PE_OK;
}

extern "C" int pe_test_wd9_tail_takes(int H, int W, int Cin, int tail_cout) { return pe::wd9_tail_takes(H, W, Cin, tail_cout) ? 1 : 0; }

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
