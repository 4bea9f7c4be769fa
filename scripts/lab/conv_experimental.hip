// Experimental 256 x 256 convolution kernels for gfx950: correct (tests/test_ops_gpu.py), measured, NOT dispatched by
// default (pe_set_conv_tile256 policy bits 6, 7 and 10).  They document what was tried against the LDS-DMA issue
// limit (scripts/overlap_probe.hip): a 4-stage ring with counted vmcnt, the phase-split
// schedule with staggered wave rows, and the phase-split schedule on top of the kw-reuse slab.
#include "conv_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// Ring-buffered variant: 256 x 256 block tile, 8 waves (2 x 4, each 128 x 64), K-step 32, FOUR 32 KiB LDS stages.
// Three K-steps of LDS-DMA are in flight while a fourth feeds the MFMAs: a wave waits with a COUNTED
// `s_waitcnt vmcnt(8)` (its 4 loads of the oldest tile have landed, 8 newer ones stay in flight), then a raw
// `s_barrier` publishes the tile to the workgroup - `__syncthreads()` would drain the DMA queue (vmcnt(0)).
// One barrier per K-step; the DMA latency (~1-2 us under load) is covered by three K-steps of MFMAs instead of
// by co-resident workgroups, so the big tile's 4x flops per L2 byte can actually be used.
// LDS rows are 64 B: chunk p of row r holds K-chunk p ^ ((r >> 2) & 3) (conflict-free ds_read_b128 for any 16
// rows distinct mod 16).
constexpr int RBK = 32, RROW = 64, RSTAGES = 4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_ring_kernel(Conv2Args a) {
    constexpr int BM = 256, BN = 256, THREADS = 512;
    constexpr int WM = 128, WN = 64, TM = 4, TN = 2;
    constexpr int A_BYTES = BM * RROW;                  // 16 KiB
    constexpr int STAGE_BYTES = (BM + BN) * RROW;       // 32 KiB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 2, lp = lane & 3;  // row within the 16-row DMA group, physical 16-B chunk

    // DMA descriptors: every wave owns 2 row groups of A and 2 of B (16 groups of 16 rows each)
    const _Float16* a_base[2];
    int a_oh[2], a_ow[2], a_coff[2];
    bool a_ok[2];
    const _Float16* b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 16 + lrow;
        const int m = m0 + r;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        const int ow = mm % a.Wo, t = mm / a.Wo;
        const int oh = t % a.Ho, n = t / a.Ho;
        a_oh[i] = oh * a.stride;
        a_ow[i] = ow * a.stride;
        a_base[i] = a.in + (size_t)n * a.H * a.W * a.Cin;
        a_coff[i] = (lp ^ ((r >> 2) & 3)) * 8;
        const int nn = n0 + r;
        b_src[i] = (nn < a.Cout) ? a.wgt + (size_t)nn * a.K + (lp ^ ((r >> 2) & 3)) * 8 : nullptr;
    }
    const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_page);

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, fsw = (frow >> 2) & 3, fkh = lane >> 5;
    const unsigned char* la = smem + (wm * WM + frow) * RROW;
    const unsigned char* lb = smem + A_BYTES + (wn * WN + frow) * RROW;
    const int nk = a.K / RBK;

    auto dma = [&](int kt) {
        const int k0 = kt * RBK;
        int kh = 0, kw = 0, c0 = k0;
        if (MODE == MODE_3X3) {
            const int tap = k0 / a.Cin;
            c0 = k0 - tap * a.Cin;
            kh = tap / 3 - 1;
            kw = tap - (tap / 3) * 3 - 1;
        }
        unsigned char* base = smem + (kt & (RSTAGES - 1)) * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ih = a_oh[i] + kh, iw = a_ow[i] + kw;
            bool ok = a_ok[i];
            if (MODE == MODE_3X3) ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const _Float16* p = ok ? a_base[i] + ((size_t)ih * a.W + iw) * a.Cin + c0 + a_coff[i] : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + (wave * 2 + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const _Float16* p = b_src[i] ? b_src[i] + k0 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + A_BYTES + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    // Software pipeline (K-step = 2 MFMA k-slices): the barrier that publishes tile kt+1 sits in the MIDDLE of
    // K-step kt, so the first fragments of tile kt+1 are fetched from LDS while the last MFMAs of tile kt run -
    // no LDS read latency is exposed after a barrier.  Fragment registers are double-buffered (F0 / F1).
    auto load_frags = [&](int stage, int ks, half8 (&af)[TM], half8 (&bf)[TN]) {
        const unsigned char* pa = la + stage * STAGE_BYTES;
        const unsigned char* pb = lb + stage * STAGE_BYTES;
        const int ch = ((ks * 2 + fkh) ^ fsw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8*>(pa + i * 32 * RROW + ch);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8*>(pb + j * 32 * RROW + ch);
    };
    auto mfma_all = [&](const half8 (&af)[TM], const half8 (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    half8 a0[TM], b0[TN], a1[TM], b1[TN];
    // prologue: three K-steps in flight; publish tile 0 and fetch its first fragments
    dma(0);
    if (nk > 1) dma(1);
    if (nk > 2) dma(2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, a0, b0);
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & (RSTAGES - 1);
        load_frags(st, 1, a1, b1);
        mfma_all(a0, b0);
        if (kt + 1 < nk) {
            // tile kt+1 landed for this wave once only the (at most one) newer tile is outstanding
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // tile kt+1 published; every wave is past tile kt-1
            if (kt + 3 < nk) dma(kt + 3);          // refill stage (kt-1) & 3
            load_frags((kt + 1) & (RSTAGES - 1), 0, a0, b0);
        }
        mfma_all(a1, b1);
    }
    __syncthreads();

    // ---- epilogue: 4 passes of 64 rows (identical to conv_big_kernel) ----
    constexpr int EP_ROW = BN + 4;
    constexpr int VEC_PER_ROW = BN / 8;
    constexpr int NV = 64 * VEC_PER_ROW / THREADS;
    float* ep = reinterpret_cast<float*>(smem);
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        half8 rres[NV];
        if (a.res_mode) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = tid + i * THREADS;
                const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
                const int m = m0 + pass * 64 + r, c = n0 + c8;
                rres[i] = zero8;
                if (m < a.M && c < a.cout_store) {
                    size_t ro;
                    if (a.res_mode == 1) {
                        ro = (size_t)m * a.Cout + c;
                    } else {
                        const int ow = m % a.Wo, t = m / a.Wo;
                        const int oh = t % a.Ho, n = t / a.Ho;
                        ro = (((size_t)n * a.resH + (oh >> 1)) * a.resW + (ow >> 1)) * a.Cout + c;
                    }
                    rres[i] = *reinterpret_cast<const half8*>(a.res + ro);
                }
            }
        }
        if (wm == (pass >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        const int c = wn * WN + j * 32 + (lane & 31);
                        ep[r * EP_ROW + c] = acc[2 * (pass & 1) + ii][j][e];
                    }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
            const int m = m0 + pass * 64 + r, c = n0 + c8;
            if (m >= a.M || c >= a.cout_store) continue;
            const float4v x0 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8);
            const float4v x1 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8 + 4);
            float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            if (a.bias) {
                const float4v b0 = *reinterpret_cast<const float4v*>(a.bias + c);
                const float4v b1 = *reinterpret_cast<const float4v*>(a.bias + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] += b0[e]; x[e + 4] += b1[e]; }
            }
            if (a.res_mode) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += (float)rres[i][e];
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
            }
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)x[e];
            *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(a.out) + (size_t)m * a.out_stride + c) = h;
        }
        __syncthreads();
    }
}

template <int MODE>
int launch_ring(const Conv2Args& a0, hipStream_t st) {
    Conv2Args a = a0;
    a.tiles_m = pe::ceil_div(a.M, 256);
    a.tiles_n = pe::ceil_div(a.Cout, 256);
    constexpr size_t lds = (size_t)RSTAGES * 512 * RROW;  // 128 KiB
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ring_kernel<MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done = true;
    }
    hipLaunchKernelGGL((conv_ring_kernel<MODE>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(256x256 ring)");
    return PE_OK;
}

// ------------------------------------------------------------------------------------------------------
// Phase-split variant ("p8"): 256 x 256 block tile, 8 waves (2 x 4, each 128 x 64), K-tile 64, LDS = 2 buffers x
// {A0, A1, B0, B1} half-tiles of 16 KiB.  Built after the 256^2 8-phase GEMM recipe of the CDNA4 guide:
//   * a K-tile is FOUR phases; phase p computes one quadrant of the wave's 128 x 64 output (64 x 32 = 2 MFMA
//     tiles x 4 k-slices = 8 MFMAs) in the order (a0,b0) (a0,b1) (a1,b1) (a1,b0): phase 1 reads a0 (8
//     ds_read_b128) + b0 (4), phase 2 b1 (4), phase 3 a1 (8), phase 4 nothing (b0 stayed in registers);
//   * the half-tiles are cut so that what every wave reads in one phase is ONE LDS half-tile: A half h = rows
//     {wm*128 + h*64 + r}, B half h = columns {wn*64 + h*32 + c}; each phase restages exactly one half-tile
//     (2 x global_load_lds per wave) 2+ phases after its last read and 5+ phases before its next use, so ~5
//     half-tiles (80 KiB / CU) are always in flight;
//   * waits are COUNTED (`s_waitcnt vmcnt(8)`: the half-tile needed next phase has landed, four newer ones stay
//     in flight) and barriers are raw `s_barrier`s - never a queue drain inside the K loop;
//   * the two wave rows run STAGGERED by one barrier (wm = 1 executes one extra s_barrier up front, wm = 0 one
//     at the end).  Every SIMD hosts one wave of each row, so while one does its 8-MFMA cluster (s_setprio 1)
//     the other issues its LDS reads and DMA for the next phase: the MFMA pipe never waits for a barrier.
// Hazards: the wait for data read in phase p sits BEFORE the first barrier of phase p-1, so that the staggered
// (one barrier behind) wave row has executed it too by the time the other row reads (RAW); a half-tile is restaged
// >= 2 phases after its last read (WAR).
template <int MODE, int ABL = 0>   // ABL (measurement only): 1 = no LDS-DMA, 2 = no MFMAs, 3 = no fragment reads
__global__ __launch_bounds__(512, 2) void conv_p8_kernel(Conv2Args a) {
    constexpr int BM = 256, BN = 256;
    constexpr int TM = 4, TN = 2;              // 128 x 64 per wave
    constexpr int HALF_BYTES = 128 * ROW_B;    // 16 KiB
    constexpr int BUF_BYTES = 4 * HALF_BYTES;  // A0 A1 B0 B1
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 3, lp = lane & 7;

    // DMA descriptors: for half-tile h, instruction i of this wave fills local rows (wave*2+i)*8 + lrow
    const _Float16* a_base[2][2];
    int a_oh[2][2], a_ow[2][2], a_coff[2];
    bool a_ok[2][2];
    const _Float16* b_src[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (wave * 2 + i) * 8 + lrow;  // local row 0..127 of a half-tile
        a_coff[i] = (lp ^ ((lr >> 1) & 7)) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
            a_ok[h][i] = m < a.M;
            const int mm = a_ok[h][i] ? m : 0;
            const int ow = mm % a.Wo, t = mm / a.Wo;
            const int oh = t % a.Ho, n = t / a.Ho;
            a_oh[h][i] = oh * a.stride;
            a_ow[h][i] = ow * a.stride;
            a_base[h][i] = a.in + (size_t)n * a.H * a.W * a.Cin;
            const int nn = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
            b_src[h][i] = (nn < a.Cout) ? a.wgt + (size_t)nn * a.K + a_coff[i] : nullptr;
        }
    }
    const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_page);

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, fsw = (frow >> 1) & 7, fkh = lane >> 5;
    const unsigned char* la = smem + (wm * 64 + frow) * ROW_B;                   // + buf*BUF + h*HALF + i*32 rows
    const unsigned char* lb = smem + 2 * HALF_BYTES + (wn * 32 + frow) * ROW_B;  // + buf*BUF + h*HALF
    const int nk = a.K / BK;

    auto issue_a = [&](int h, int kt) {
        if (kt >= nk || ABL == 1) return;
        const int k0 = kt * BK;
        int kh = 0, kw = 0, c0 = k0;
        if (MODE == MODE_3X3) {
            const int tap = k0 / a.Cin;
            c0 = k0 - tap * a.Cin;
            kh = tap / 3 - 1;
            kw = tap - (tap / 3) * 3 - 1;
        }
        unsigned char* base = smem + (kt & 1) * BUF_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ih = a_oh[h][i] + kh, iw = a_ow[h][i] + kw;
            bool ok = a_ok[h][i];
            if (MODE == MODE_3X3) ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const _Float16* p = ok ? a_base[h][i] + ((size_t)ih * a.W + iw) * a.Cin + c0 + a_coff[i] : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    auto issue_b = [&](int h, int kt) {
        if (kt >= nk || ABL == 1) return;
        unsigned char* base = smem + (kt & 1) * BUF_BYTES + (2 + h) * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const _Float16* p = b_src[h][i] ? b_src[h][i] + kt * BK : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
    half8 fa[2][4], fb0[4], fb1[4];  // A sub-tile (2 row tiles x 4 k-slices), B sub-tiles (4 k-slices each)
    if (ABL == 3) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fa[0][ks] = fa[1][ks] = fb0[ks] = fb1[ks] = *reinterpret_cast<const half8*>(la + ks * 16);
        }
    }
    auto read_a = [&](int h, int buf) {
        if (ABL == 3) return;
        const unsigned char* p = la + buf * BUF_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[i][ks] = *reinterpret_cast<const half8*>(p + i * 32 * ROW_B + (((ks * 2 + fkh) ^ fsw) << 4));
    };
    auto read_b = [&](int h, int buf, half8 (&f)[4]) {
        if (ABL == 3) return;
        const unsigned char* p = lb + buf * BUF_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const half8*>(p + (((ks * 2 + fkh) ^ fsw) << 4));
    };
#define P8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P8_MFMA(AH, BH, FB)                                                                                      \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int ks = 0; ks < (ABL == 2 ? 0 : 4); ++ks) {                                      \
            acc[2 * AH][BH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], FB[ks], acc[2 * AH][BH], 0, 0, 0); \
            acc[2 * AH + 1][BH] =                                                                                \
                __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], FB[ks], acc[2 * AH + 1][BH], 0, 0, 0);         \
        }                                                                                                        \
        if (ABL == 2) acc[2 * AH][BH][0] += (float)fa[0][0][0] + (float)fa[1][3][0] + (float)FB[0][0] + (float)FB[3][0]; \
        __builtin_amdgcn_s_setprio(0);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)

    // ---- prologue: the six half-tiles the steady state would have in flight at tile 0, phase 1 ----
    issue_a(0, 0); issue_b(0, 0); issue_b(1, 0); issue_a(1, 0); issue_a(0, 1); issue_b(0, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A0(0), B0(0) landed
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    P8_BARRIER();
    // stagger: group 1 runs one barrier behind group 0.  (a.ablate >> 4) picks the partition for measurements.
    const int gsel = a.ablate >> 4;
    const int grp = gsel == 0 ? (wave >> 2) : gsel == 1 ? (wave & 1) : gsel == 2 ? ((wave >> 1) & 1) : 0;
    if (grp == 1) P8_BARRIER();

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
        // ---- phase 1: quadrant (a0, b0) ----
        read_a(0, buf);
        read_b(0, buf, fb0);
        issue_b(1, t + 1);
        if (n1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // B1(t) landed (read in phase 2)
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        P8_BARRIER();
        P8_MFMA(0, 0, fb0);
        P8_BARRIER();
        // ---- phase 2: (a0, b1) ----
        read_b(1, buf, fb1);
        issue_a(1, t + 1);
        if (n1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A1(t) landed (read in phase 3)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P8_BARRIER();
        P8_MFMA(0, 1, fb1);
        P8_BARRIER();
        // ---- phase 3: (a1, b1) ----
        read_a(1, buf);
        issue_a(0, t + 2);
        P8_BARRIER();
        P8_MFMA(1, 1, fb1);
        P8_BARRIER();
        // ---- phase 4: (a1, b0), no LDS reads ----
        issue_b(0, t + 2);
        if (n2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // A0(t+1), B0(t+1) landed (read in the next phase 1)
        else if (n1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        P8_BARRIER();
        P8_MFMA(1, 0, fb0);
        P8_BARRIER();
    }
    if (grp == 0 && gsel != 3) P8_BARRIER();  // balance the stagger
#undef P8_MFMA
#undef P8_BARRIER
    __syncthreads();
    epilogue256(a, acc, smem, m0, n0, tid, lane, wm, wn);
}

template <int MODE>
int launch_p8(const Conv2Args& a0, hipStream_t st) {
    Conv2Args a = a0;
    a.tiles_m = pe::ceil_div(a.M, 256);
    a.tiles_n = pe::ceil_div(a.Cout, 256);
    constexpr size_t lds = (size_t)2 * 4 * 128 * ROW_B;  // 128 KiB
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p8_kernel<MODE, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done = true;
    }
    if (a.ablate & 15) {
        auto set = [&](const void* f) { (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); };
        const int ab = a.ablate & 15;
        if (ab == 1) { set((const void*)conv_p8_kernel<MODE, 1>); hipLaunchKernelGGL((conv_p8_kernel<MODE, 1>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a); }
        else if (ab == 2) { set((const void*)conv_p8_kernel<MODE, 2>); hipLaunchKernelGGL((conv_p8_kernel<MODE, 2>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a); }
        else { set((const void*)conv_p8_kernel<MODE, 3>); hipLaunchKernelGGL((conv_p8_kernel<MODE, 3>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a); }
        PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(256x256 phase-split, ablation)");
        return PE_OK;
    }
    hipLaunchKernelGGL((conv_p8_kernel<MODE>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(256x256 phase-split)");
    return PE_OK;
}

// ------------------------------------------------------------------------------------------------------
// "p8r": the phase-split 256 x 256 kernel with the kw-reuse slab of the 3x3 kernel (3x3 / stride 1 only).
// Motivation (scripts/overlap_probe.hip): under MFMA load a SIMD issues one 1 KiB LDS-DMA instruction per ~190 clk,
// so the instruction count per MFMA is what bounds these kernels.  Per wave and tap (32 MFMAs = 1024 clk):
//   kw-reuse 256 x 128 x 2 workgroups: 7.3 instructions, phase-split generic 256 x 256: 8, THIS kernel: 4 (weights)
//   + 5/3 (slab) = 5.7.
// Structure = conv_p8_kernel (four quadrant phases per tap, staggered wave rows, counted vmcnt, raw barriers), but
// the A operand is read from a slab of 258 consecutive input pixels per (channel chunk, kernel row) that serves the
// three kw taps from rows r, r+1, r+2 (edge columns masked on the fragment, see conv3x3r_kernel) and is
// double-buffered: the next slab streams in as five 1 KiB pieces per wave spread over the first two taps.
// LDS: 2 slabs x 33 KiB + 2 taps x {B0, B1} x 16 KiB = 130 KiB.
// DMA order per wave and group of three taps (s = slab piece, b = weight half-tile = 2 instructions):
//   tap0: P1 s  P2 s  P3 b0(t+2)  P4 b1(t+2) | tap1: the same | tap2: P1 s  P2 -  P3 b0(t+2)  P4 b1(t+2)
// Waits (before the first barrier of the phase, see conv_p8_kernel for the hazard rules):
//   P1 (for b1(t), read in P2):            vmcnt(7), vmcnt(6) when t % 3 == 0
//   P4 (for b0(t+1) [+ the whole next slab when t % 3 == 2], read in the next P1): vmcnt(8), vmcnt(4) when t % 3 == 2
//   last group (nothing left to prefetch): vmcnt(0).
template <int DUMMY>
__global__ __launch_bounds__(512, 2) void conv_p8r_kernel(Conv2Args a) {
    constexpr int BM = 256, BN = 256;
    constexpr int HALF_BYTES = 128 * ROW_B;            // 16 KiB weight half-tile
    constexpr int SLAB_BYTES = 33 * 1024;              // 264 rows x 128 B
    constexpr int B_OFF = 2 * SLAB_BYTES;              // weight buffers behind the two slabs
    constexpr int BUF_BYTES = 2 * HALF_BYTES;          // B0 B1 of one tap
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 2, wn = wave & 3;
    const int lrow = lane >> 3, lp = lane & 7;

    // ---- slab pieces of this wave: piece q < 4 is row group wave + 8q, piece 4 is row group 32 (issued by every wave:
    //      identical bytes to the same place, keeps the vmcnt arithmetic wave-uniform).  Per lane only the pixel index
    //      of piece 0 and a validity mask are kept: bit 4*q + kh says that slab row's centre user exists and its
    //      input row oh + kh - 1 lies inside the image ----
    const int s_m0 = m0 + wave * 8 + lrow - 1;   // piece q < 4: s_m0 + 64 q; piece 4: m0 + 255 + lrow
    unsigned s_valid = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int j = (q < 4 ? wave + 8 * q : 32) * 8 + lrow;
        const int m = m0 + j - 1;
        if (m >= 0 && m < a.M && j < BM + 2) {
            const int oh = (m / a.Wo) % a.Ho;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                if ((unsigned)(oh + kh - 1) < (unsigned)a.H) s_valid |= 1u << (4 * q + kh);
        }
    }
    const int s_coff = (lp ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7)) * 8;  // ((g*8 + lrow) >> 1) & 7 with g = wave + 8q
    const int s_coff4 = (lp ^ ((lrow >> 1) & 7)) * 8;                        // g = 32
    // ---- weight half-tiles: half h = columns {wn'*64 + h*32 + c}; this wave fills local rows (wave*2+i)*8 + lrow.
    //      Cout % 256 == 0 (dispatch), so every column exists; half 1 is 32 weight rows further on ----
    const _Float16* b_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (wave * 2 + i) * 8 + lrow;
        b_src[i] = a.wgt + (size_t)(n0 + (lr >> 5) * 64 + (lr & 31)) * a.K + (lp ^ ((lr >> 1) & 7)) * 8;
    }
    const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_page);

    float16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, fsw = (frow >> 1) & 7, fkh = lane >> 5;
    // edge masks of the four 32-row tiles of this wave (bit i: not the first column, bit 4+i: not the last column)
    unsigned edge = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 128 + i * 32 + frow;
        const int ow = (m < a.M ? m : 0) % a.Wo;
        edge |= (ow != 0 ? 1u : 0u) << i;
        edge |= (ow != a.Wo - 1 ? 1u : 0u) << (4 + i);
    }
    const unsigned char* lb = smem + B_OFF + (wn * 32 + frow) * ROW_B;
    const int chunks = a.Cin / BK;
    const int groups = 3 * chunks;          // (channel chunk, kernel row), chunk outer
    const int ntaps = 3 * groups;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto issue_slab = [&](int q, int s) {   // piece q of slab s
        if (s >= groups) return;
        const int cc = s / 3, kh = s - cc * 3;
        const bool ok = (s_valid >> (4 * q + kh)) & 1u;
        const int g = q < 4 ? wave + 8 * q : 32;
        const int mm = q < 4 ? s_m0 + 64 * q : m0 + 255 + lrow;
        const _Float16* p = ok ? a.in + (size_t)(mm + (kh - 1) * a.W) * a.Cin + cc * BK + (q < 4 ? s_coff : s_coff4) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + (s & 1) * SLAB_BYTES + g * 1024), 16, 0, 0);
    };
    auto issue_b = [&](int h, int t) {      // weight half-tile h of tap t
        if (t >= ntaps) return;
        const int s = t / 3, kw = t - s * 3;
        const int cc = s / 3, kh = s - cc * 3;
        const size_t k0 = (size_t)(kh * 3 + kw) * a.Cin + cc * BK + (size_t)h * 32 * a.K;
        unsigned char* base = smem + B_OFF + (t & 1) * BUF_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + k0), (lptr_t)(base + (wave * 2 + i) * 1024), 16, 0, 0);
    };
    half8 fa[2][4], fb0[4], fb1[4];
    auto read_a = [&](int h, int sb, int kw) {   // rows wm*128 + h*64 + i*32 + frow of the tile = slab rows + kw
        const unsigned char* base = smem + sb * SLAB_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wm * 128 + h * 64 + i * 32 + frow + kw;
            const int sw = (r >> 1) & 7;
            const bool keep = kw == 1 || ((edge >> ((kw == 0 ? 0 : 4) + 2 * h + i)) & 1u);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 v = *reinterpret_cast<const half8*>(base + r * ROW_B + (((ks * 2 + fkh) ^ sw) << 4));
                fa[i][ks] = keep ? v : zero8;
            }
        }
    };
    auto read_b = [&](int h, int buf, half8 (&f)[4]) {
        const unsigned char* p = lb + buf * BUF_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const half8*>(p + (((ks * 2 + fkh) ^ fsw) << 4));
    };
#define P8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P8_MFMA(AH, BH, FB)                                                                                      \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                       \
            acc[2 * AH][BH] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], FB[ks], acc[2 * AH][BH], 0, 0, 0); \
            acc[2 * AH + 1][BH] =                                                                                \
                __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], FB[ks], acc[2 * AH + 1][BH], 0, 0, 0);         \
        }                                                                                                        \
        __builtin_amdgcn_s_setprio(0);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)

    // ---- prologue: slab 0 and the weights of taps 0 and 1, drained ----
#pragma unroll
    for (int q = 0; q < 5; ++q) issue_slab(q, 0);
    issue_b(0, 0); issue_b(1, 0); issue_b(0, 1); issue_b(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_BARRIER();
    if (wm == 1) P8_BARRIER();  // stagger: wave row 1 runs one barrier behind wave row 0

    int t = 0;
    for (int s = 0; s < groups; ++s) {
        const int sb = s & 1;
        const bool last = s + 1 >= groups;
#pragma unroll 1
        for (int kw = 0; kw < 3; ++kw, ++t) {   // NOT unrolled: the slab read addresses depend on kw and would be hoisted
            const int buf = t & 1;
            // ---- phase 1: (a0, b0) ----
            read_a(0, sb, kw);
            read_b(0, buf, fb0);
            issue_slab(kw * 2, s + 1);                       // pieces 0, 2, 4
            if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (kw == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            P8_BARRIER();
            P8_MFMA(0, 0, fb0);
            P8_BARRIER();
            // ---- phase 2: (a0, b1) ----
            read_b(1, buf, fb1);
            if (kw < 2) issue_slab(kw * 2 + 1, s + 1);       // pieces 1, 3
            P8_BARRIER();
            P8_MFMA(0, 1, fb1);
            P8_BARRIER();
            // ---- phase 3: (a1, b1) ----
            read_a(1, sb, kw);
            issue_b(0, t + 2);
            P8_BARRIER();
            P8_MFMA(1, 1, fb1);
            P8_BARRIER();
            // ---- phase 4: (a1, b0) ----
            issue_b(1, t + 2);
            if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (kw == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            P8_BARRIER();
            P8_MFMA(1, 0, fb0);
            P8_BARRIER();
        }
    }
    if (wm == 0) P8_BARRIER();  // balance the stagger
#undef P8_MFMA
#undef P8_BARRIER
    __syncthreads();
    epilogue256(a, acc, smem, m0, n0, tid, lane, wm, wn);
}

int launch_p8r(const Conv2Args& a0, hipStream_t st) {
    Conv2Args a = a0;
    a.tiles_m = pe::ceil_div(a.M, 256);
    a.tiles_n = pe::ceil_div(a.Cout, 256);
    constexpr size_t lds = (size_t)2 * 33 * 1024 + 4 * 128 * ROW_B;  // 130 KiB
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p8r_kernel<0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done = true;
    }
    hipLaunchKernelGGL((conv_p8r_kernel<0>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(256x256 phase-split, kw-reuse)");
    return PE_OK;
}

}  // namespace

namespace pe {
int launch_conv_ring(const Conv2Args& a, int mode3x3, hipStream_t st) {
    return mode3x3 ? launch_ring<MODE_3X3>(a, st) : launch_ring<MODE_1X1>(a, st);
}
int launch_conv_p8(const Conv2Args& a, int mode3x3, hipStream_t st) {
    return mode3x3 ? launch_p8<MODE_3X3>(a, st) : launch_p8<MODE_1X1>(a, st);
}
int launch_conv_p8r(const Conv2Args& a, hipStream_t st) { return launch_p8r(a, st); }
}  // namespace pe
