// Implicit-GEMM convolution, second generation: LDS-DMA staging (global_load_lds, 16 B / lane) into an
// XOR-swizzled, unpadded LDS image, single LDS stage, 3-4 workgroups per CU.
//
// Same contract and epilogue as csrc/conv_igemm.hip (which keeps the 7x7 stem); same reference rows replaced
// (layers/wrappers.py:62-98 + layers/batch_norm.py:45-65 + relu_ + backbone/resnet.py:205-221 residual add +
// backbone/fpn.py:129-137 top-down add; roi_heads/box_head.py:73-81 FC with H = W = 1).
//
// Why: with register staging + padded double-buffered LDS (74 KB) only 2 workgroups fit a CU, and each one
// exposes a full HBM/L2 round trip per K-step (measured r01: 3x3 res4 525 TFLOP/s, memory-bound 1x1 at
// 1.5-1.8 TB/s).  Here the A (gathered pixels) and B (weights) K-slabs go HBM -> LDS without touching VGPRs:
//   * one `global_load_lds_dwordx4` fills 8 rows x 128 B (1 KiB, lane-linear) per wavefront instruction;
//   * the LDS image is [row][8 chunks of 16 B] with chunk p of row r holding K-chunk p ^ ((r >> 1) & 7):
//     the swizzle is applied to the per-lane SOURCE address (the DMA destination must stay linear) and to
//     the fragment reads, which makes every ds_read_b128 of an MFMA operand bank-conflict-free;
//   * out-of-image taps (3x3 halo) and rows beyond M read a 16-byte zero page instead of branching;
//   * 32 KiB of LDS per workgroup (+ a two-pass fp32 epilogue staging of 33 KiB that reuses it) lets 4
//     workgroups share a CU, so one workgroup's DMA wait overlaps three others' MFMAs.
#include <atomic>

#include "conv_common.h"

namespace {

template <int BM, int BN, int MODE, int STAGES>
__global__ __launch_bounds__(BM * 2, BM == 128 ? (STAGES == 1 ? 3 : 2) : (STAGES == 1 ? 4 : 2)) void conv_igemm2_kernel(Conv2Args a) {
    constexpr int THREADS = BM * 2, WAVES = BM / 32;  // waves as (BM/64) x 2, each owning a 64 x (BN/2) sub-tile
    constexpr int WM = 64, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_INSTR = BM / 8 / WAVES;  // DMA instructions per wave per K-step for A (8 rows each) = 4
    constexpr int B_INSTR = BN / 8 / WAVES;
    constexpr int A_BYTES = BM * ROW_B;
    constexpr int STAGE_BYTES = (BM + BN) * ROW_B;
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
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 3, lp = lane & 7;  // row within the 8-row DMA group, physical 16-B chunk

    // ---- per-lane DMA descriptors ----
    const _Float16* a_base[A_INSTR];
    int a_oh[A_INSTR], a_ow[A_INSTR], a_coff[A_INSTR];
    bool a_ok[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (wave * A_INSTR + i) * 8 + lrow;
        const int m = m0 + r;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        const int ow = mm % a.Wo, t = mm / a.Wo;
        const int oh = t % a.Ho, n = t / a.Ho;
        a_oh[i] = oh * a.stride;
        a_ow[i] = ow * a.stride;
        a_base[i] = a.in + (size_t)n * a.H * a.W * a.Cin;
        a_coff[i] = (lp ^ ((r >> 1) & 7)) * 8;  // logical K-chunk held by this lane's physical slot
    }
    const _Float16* b_src[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (wave * B_INSTR + i) * 8 + lrow;
        const int n = n0 + r;
        b_src[i] = (n < a.Cout) ? a.wgt + (size_t)n * a.K + (lp ^ ((r >> 1) & 7)) * 8 : nullptr;
    }
    const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_page);

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read offsets: row = wave tile base + (lane & 31); K-chunk (2*ks + (lane >> 5)) ^ swizzle(row)
    const int frow = lane & 31, fsw = (frow >> 1) & 7, fkh = lane >> 5;
    const unsigned char* la = smem + (wm * WM + frow) * ROW_B;
    const unsigned char* lb = smem + A_BYTES + (wn * WN + frow) * ROW_B;

    const int nk = a.K / BK;
    const Bias8 bias8 = preload_bias8<BN>(a, n0, tid);
    // pass 0 of the residual under the whole K-loop: same-box A/B -6 % on res3 conv3 (K = 128: the tile is all epilogue), +-1..3 % on
    // the others, +0.25 % in the pipeline (profiles/r05_res_prefetch_ab.txt)
    const ResVecs<BN, THREADS> res0 = load_res<BN, THREADS>(a, m0, n0, tid, 0);
    // LDS-DMA of K-step kt into LDS stage `st`: global -> LDS, no VGPR round trip
    auto dma = [&](int kt, int st) {
        const int k0 = kt * BK;
        int kh = 0, kw = 0, c0 = k0;
        if (MODE == MODE_3X3) {
            const int tap = k0 / a.Cin;
            c0 = k0 - tap * a.Cin;
            kh = tap / 3 - 1;
            kw = tap - (tap / 3) * 3 - 1;
        }
        unsigned char* base = smem + st * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int ih = a_oh[i] + kh, iw = a_ow[i] + kw;
            bool ok = a_ok[i];
            if (MODE == MODE_3X3) ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const _Float16* p = ok ? a_base[i] + ((size_t)ih * a.W + iw) * a.Cin + c0 + a_coff[i] : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + (wave * A_INSTR + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const _Float16* p = b_src[i] ? b_src[i] + k0 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + A_BYTES + (wave * B_INSTR + i) * 1024), 16, 0, 0);
        }
    };
    auto compute = [&](int st) {
        const unsigned char* pa = la + st * STAGE_BYTES;
        const unsigned char* pb = lb + st * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = ((ks * 2 + fkh) ^ fsw) << 4;
            half8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8*>(pa + i * 32 * ROW_B + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8*>(pb + j * 32 * ROW_B + ch);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    if (STAGES == 1) {
        // single stage: DMA wait fully exposed per workgroup; 3-4 co-resident workgroups hide each other's waits
        for (int kt = 0; kt < nk; ++kt) {
            dma(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();  // every wave is done reading before the next K-step's DMA lands
        }
    } else {
        // two stages: K-step kt+1 streams into the other stage while K-step kt feeds the MFMAs; one barrier per
        // K-step (it both publishes stage kt+1 and retires the reads of stage kt)
        dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) dma(kt + 1, (kt + 1) & 1);
            compute(kt & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    epilogue<BM, BN, THREADS>(a, acc, smem, m0, n0, tid, lane, wm, wn, bias8, &res0);
}

// ------------------------------------------------------------------------------------------------------
// 3x3 (stride 1, pad 1) with kw-tap reuse of the A slab.
// The generic kernel above moves one 16 KiB A tile per (tap, 64-channel chunk): every input pixel crosses
// L2 -> LDS nine times.  For a fixed kernel row kh the three kw taps read the SAME pixels shifted by one
// position, so here the block loads ONE slab of BM + 2 consecutive (linear) input pixels per (kh, chunk)
// and runs the three kw taps from LDS rows i, i+1, i+2 - only the 16 KiB weight tile changes per tap.
// L2 -> LDS bytes per 3 taps: 17 + 3*16 = 65 KiB instead of 96 KiB; same LDS footprint (33 KiB).
// Correctness of the shift across image-row / image boundaries:
//   * slab row j holds input pixel q = m0 - 1 + j + (kh-1)*W; it is loaded as ZERO when its centre user
//     (output pixel m0 + j - 1) has oh + kh - 1 outside [0, H) or lies outside [0, M);
//   * its two other users (kw = 0 / 2) either sit in the same image row (same validity) or are exactly the
//     cases ow == 0 (kw = 0) / ow == W-1 (kw = 2), which are masked to zero on the A FRAGMENT (per lane).
// The WEIGHT tile is double-buffered: tap t+1's 16 KiB B tile streams into the other LDS stage while
// tap t feeds the MFMAs, so per (kh, chunk) group only the A slab DMA is exposed (4 barriers per 3 taps).  LDS: slab + 2 x 16 KiB (66 KiB at BM = 256: still two 8-wave workgroups per CU).
template <int BM, int BN>
__global__ __launch_bounds__(BM * 2, BM == 128 ? 3 : 4) void conv3x3rb_kernel(Conv2Args a) {
        constexpr int THREADS = BM * 2, WAVES = BM / 32;   // waves as (BM/64) x 2, each 64 x (BN/2)
    constexpr int WM = 64, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int SLAB_ROWS = BM + 8;         // BM + 2 rounded up to a multiple of 8
    constexpr int LAST_GROUP = BM / 8;        // DMA group holding slab rows BM .. BM+7 (wave 0's extra one)
    constexpr int B_INSTR = BN / 8 / WAVES;
    constexpr int A_BYTES = SLAB_ROWS * ROW_B;
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
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane >> 3, lp = lane & 7;

    // slab DMA: wave w owns row groups 4w..4w+3, wave 0 also the last group (descriptors are recomputed per slab
    // - once per three K-steps - instead of living in registers)
    constexpr int A_INSTR = 5;
    const _Float16* b_src[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (wave * B_INSTR + i) * 8 + lrow;
        const int n = n0 + r;
        b_src[i] = (n < a.Cout) ? a.wgt + (size_t)n * a.K + (lp ^ ((r >> 1) & 7)) * 8 : nullptr;
    }
    const _Float16* zero = reinterpret_cast<const _Float16*>(g_zero_page);

    // fragment rows and edge masks (per lane, per M sub-tile)
    const int frow = lane & 31, fkh = lane >> 5;
    bool not_left[TM], not_right[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WM + i * 32 + frow;
        const int ow = (m < a.M ? m : 0) % a.Wo;
        not_left[i] = ow != 0;
        not_right[i] = ow != a.Wo - 1;
    }
    const unsigned char* lb = smem + A_BYTES + (wn * WN + frow) * ROW_B;
    const int fswb = (frow >> 1) & 7;

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    constexpr int B_BYTES = BN * ROW_B;
    const int chunks = a.Cin / BK;
    const int groups = 3 * chunks;
    auto dma_b = [&](int tap_k0, int stage) {
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const _Float16* p = b_src[i] ? b_src[i] + tap_k0 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + A_BYTES + stage * B_BYTES + (wave * B_INSTR + i) * 1024), 16, 0, 0);
        }
    };
    dma_b(0, 0);  // weight tile of the very first tap
    int g = 0;    // running tap counter: tap g uses B stage g & 1
    for (int grp = 0; grp < groups; ++grp) {
        // channel chunk OUTER, kernel row INNER: consecutive slabs are the same pixels shifted by one image row, so
        // the re-read hits the XCD's L2 (LDS-DMA from L2 runs at ~35 TB/s, from MALL/HBM at ~5.5: scripts/lds_dma_probe)
        const int cc = grp / 3, kh = grp - cc * 3;
        const int c0 = cc * BK;
        // ---- A slab for (kh, chunk); every wave finished reading the previous slab at the barrier below ----
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            if (i == 4 && wave != 0) continue;  // wave-uniform
            const int gg = i < 4 ? wave * 4 + i : LAST_GROUP;
            const int j = gg * 8 + lrow;
            const int m = m0 + j - 1;
            bool ok = m >= 0 && m < a.M && j < BM + 2;
            const int mm = ok ? m : 0;
            const int ih = (mm / a.Wo) % a.Ho + kh - 1;
            ok = ok && (unsigned)ih < (unsigned)a.H;
            const _Float16* p = ok ? a.in + (size_t)(mm + (kh - 1) * a.W) * a.Cin + c0 + (lp ^ ((j >> 1) & 7)) * 8 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + gg * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            // prefetch the NEXT tap's weight tile into the other stage
            if (kw < 2) {
                dma_b((kh * 3 + kw + 1) * a.Cin + c0, (g + 1) & 1);
            } else if (grp + 1 < groups) {
                const int ncc = (grp + 1) / 3, nkh = (grp + 1) - ncc * 3;
                dma_b((nkh * 3) * a.Cin + ncc * BK, (g + 1) & 1);
            }
            const unsigned char* lbs = lb + (g & 1) * B_BYTES;
            const int arow0 = wm * WM + frow + kw;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                half8 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int r = arow0 + i * 32;
                    const int ch = ((ks * 2 + fkh) ^ ((r >> 1) & 7)) << 4;
                    af[i] = *reinterpret_cast<const half8*>(smem + r * ROW_B + ch);
                    if (kw == 0) af[i] = not_left[i] ? af[i] : zero8;
                    if (kw == 2) af[i] = not_right[i] ? af[i] : zero8;
                }
                const int chb = ((ks * 2 + fkh) ^ fswb) << 4;
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8*>(lbs + j * 32 * ROW_B + chb);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            ++g;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // next weight tile landed; this tap's LDS reads retired
        }
    }
    // (128 registers at 256-row tiles: no room to carry the bias or a residual through the loop - both are requested here)
    epilogue<BM, BN, THREADS>(a, acc, smem, m0, n0, tid, lane, wm, wn, preload_bias8<BN>(a, n0, tid), nullptr);
}

template <int BM, int BN>
int launch3x3r(const Conv2Args& a0, hipStream_t st) {
    Conv2Args a = a0;
    a.tiles_m = pe::ceil_div(a.M, BM);
    a.tiles_n = pe::ceil_div(a.Cout, BN);
    constexpr size_t epi = (size_t)64 * (BN + 4) * 4;
    constexpr size_t stage_b = (size_t)(BM + 8 + 2 * BN) * ROW_B;
    constexpr size_t lds_b = stage_b > epi ? stage_b : epi;
    const dim3 grid(a.tiles_m * a.tiles_n), block(BM * 2);
    PE_ENSURE_LDS((conv3x3rb_kernel<BM, BN>), lds_b, "pe_conv2d_nhwc_f16(3x3 row-reuse)");
    hipLaunchKernelGGL((conv3x3rb_kernel<BM, BN>), grid, block, lds_b, st, a);
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(3x3 row-reuse, B double-buffered)");
    return PE_OK;
}

// ------------------------------------------------------------------------------------------------------
// Big-tile variant: 256 x 256 x 64 block tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA tiles, 128 fp32
// accumulator registers), TWO LDS stages of 64 KiB: K-step kt+1 streams in by LDS-DMA while K-step kt feeds the
// MFMAs, one barrier per K-step.  One workgroup per CU (2 waves per SIMD).  Twice the flops per L2->LDS byte of
// the 128 x 128 kernel (the CU's L2 port, ~64 B/clk, is what the small tile saturates) and 0.75 instead of 1.0
// fragment reads per MFMA.  Used for launches with Cout % 256 == 0 whose grid fills the chip.
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_big_kernel(Conv2Args a) {
    constexpr int BM = 256, BN = 256;
    constexpr int WM = 128, WN = 64, TM = 4, TN = 2;
    constexpr int A_INSTR = 4, B_INSTR = 4;         // 32 row groups of 8 rows over 8 waves, for A and for B
    constexpr int A_BYTES = BM * ROW_B;
    constexpr int STAGE_BYTES = (BM + BN) * ROW_B;  // 64 KiB
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

    const _Float16* a_base[A_INSTR];
    int a_oh[A_INSTR], a_ow[A_INSTR], a_coff[A_INSTR];
    bool a_ok[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int r = (wave * A_INSTR + i) * 8 + lrow;
        const int m = m0 + r;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        const int ow = mm % a.Wo, t = mm / a.Wo;
        const int oh = t % a.Ho, n = t / a.Ho;
        a_oh[i] = oh * a.stride;
        a_ow[i] = ow * a.stride;
        a_base[i] = a.in + (size_t)n * a.H * a.W * a.Cin;
        a_coff[i] = (lp ^ ((r >> 1) & 7)) * 8;
    }
    const _Float16* b_src[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int r = (wave * B_INSTR + i) * 8 + lrow;
        const int n = n0 + r;
        b_src[i] = (n < a.Cout) ? a.wgt + (size_t)n * a.K + (lp ^ ((r >> 1) & 7)) * 8 : nullptr;
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
    const unsigned char* la = smem + (wm * WM + frow) * ROW_B;
    const unsigned char* lb = smem + A_BYTES + (wn * WN + frow) * ROW_B;
    const int nk = a.K / BK;

    auto dma = [&](int kt, int st) {
        const int k0 = kt * BK;
        int kh = 0, kw = 0, c0 = k0;
        if (MODE == MODE_3X3) {
            const int tap = k0 / a.Cin;
            c0 = k0 - tap * a.Cin;
            kh = tap / 3 - 1;
            kw = tap - (tap / 3) * 3 - 1;
        }
        unsigned char* base = smem + st * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const int ih = a_oh[i] + kh, iw = a_ow[i] + kw;
            bool ok = a_ok[i];
            if (MODE == MODE_3X3) ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const _Float16* p = ok ? a_base[i] + ((size_t)ih * a.W + iw) * a.Cin + c0 + a_coff[i] : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + (wave * A_INSTR + i) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const _Float16* p = b_src[i] ? b_src[i] + k0 : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(base + A_BYTES + (wave * B_INSTR + i) * 1024), 16, 0, 0);
        }
    };
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) dma(kt + 1, (kt + 1) & 1);
        const unsigned char* pa = la + (kt & 1) * STAGE_BYTES;
        const unsigned char* pb = lb + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = ((ks * 2 + fkh) ^ fsw) << 4;
            half8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8*>(pa + i * 32 * ROW_B + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8*>(pb + j * 32 * ROW_B + ch);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    epilogue256(a, acc, smem, m0, n0, tid, lane, wm, wn);
}

template <int MODE>
int launch_big(const Conv2Args& a0, hipStream_t st) {
    Conv2Args a = a0;
    a.tiles_m = pe::ceil_div(a.M, 256);
    a.tiles_n = pe::ceil_div(a.Cout, 256);
    constexpr size_t lds = (size_t)2 * 512 * ROW_B;  // 128 KiB (the epilogue's 65 KiB staging reuses it)
    PE_ENSURE_LDS((conv_big_kernel<MODE>), lds, "pe_conv2d_nhwc_f16(256x256)");
    hipLaunchKernelGGL((conv_big_kernel<MODE>), dim3(a.tiles_m * a.tiles_n), dim3(512), lds, st, a);
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(256x256)");
    return PE_OK;
}

template <int BM, int BN, int MODE, int STAGES = 1>
int launch2(const Conv2Args& a0, hipStream_t st) {
    Conv2Args a = a0;
    a.tiles_m = pe::ceil_div(a.M, BM);
    a.tiles_n = pe::ceil_div(a.Cout, BN);
    constexpr size_t stage = (size_t)(BM + BN) * ROW_B * STAGES;
    constexpr size_t epi = (size_t)64 * (BN + 4) * 4;
    constexpr size_t lds = stage > epi ? stage : epi;
    PE_ENSURE_LDS((conv_igemm2_kernel<BM, BN, MODE, STAGES>), lds, "pe_conv2d_nhwc_f16(v2)");
    hipLaunchKernelGGL((conv_igemm2_kernel<BM, BN, MODE, STAGES>), dim3(a.tiles_m * a.tiles_n), dim3(BM * 2), lds, st, a);
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(v2)");
    return PE_OK;
}

}  // namespace

namespace pe {
bool conv1x1_ring_eligible(int M, int K, int Cout, int cout_store, int out_stride, long long in_pixels);
int conv1x1_ring_launch(const void* in, const void* wgt, const float* bias, const void* res, int res_mode, int resH, int resW, void* out, int N,
                        int H, int W, int Ho, int Wo, int stride, int M, int K, int Cout, int out_stride, int relu, hipStream_t st);
std::atomic<int> g_conv_tile256{329};  // bit 0: 256-row tiles for big 3x3 launches, bit 1: for big 1x1 launches, bit 2: two-stage 1x1 pipeline,
                                      // bit 3: 256x256 two-stage kernel for long-K GEMMs, bit 4: ... for every eligible launch,
                                      // bit 5 / 6: the persistent loader / consumer 1x1 kernel for res4 conv1 / for every eligible residual-free launch,
                                      // bit 8: ... for the residual layers as well
std::atomic<int> g_conv3x3_reuse{1};  // 0: the generic per-tap 3x3 kernel instead of the kw-reuse one (A/B measurements)
// called from pe_conv2d_nhwc_f16 (conv_igemm.hip) for the 1x1 / 3x3 cases
int conv2_dispatch(const void* in, const void* wgt, const float* bias, const void* res, void* out, int N, int H, int W,
                   int Cin, int Cout, int Ho, int Wo, int K, int M, int mode3x3, int stride, int relu, int res_mode,
                   int resH, int resW, int out_f32, int cout_store, int out_stride, hipStream_t st) {
    Conv2Args a{};
    a.in = (const _Float16*)in; a.wgt = (const _Float16*)wgt; a.bias = bias; a.res = (const _Float16*)res; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.stride = stride; a.M = M; a.K = K;
    a.relu = relu; a.res_mode = res_mode; a.resH = resH; a.resW = resW; a.out_f32 = out_f32;
    a.cout_store = cout_store; a.out_stride = out_stride;
    const int policy = g_conv_tile256.load(std::memory_order_relaxed);
    const bool narrow = Cout <= 64;
    // persistent loader / consumer 1x1 kernel (csrc/conv1x1_ring.hip; bit-identical results): policy bit 5 = the long-K, 256-output
    // layers (res4 conv1), bit 6 = every eligible launch (residual-free, fp16 out, bias, Cout % 256 == 0; K >= 512 at stride 1, the
    // stride-2 shortcut convolutions from K = 256).  Chosen by channel counts and stride only, never by the batch.
    if ((policy & 96) && !mode3x3 && (stride == 1 || stride == 2) && !out_f32 && bias && conv1x1_ring_eligible(M, K, Cout, cout_store, out_stride, (long long)N * H * W)) {
        bool take;
        if (!(policy & 64)) take = stride == 1 && !res_mode && K >= 1024 && Cout == 256;
        else if (res_mode) take = stride == 1 && K >= 128 && (policy & 256) && (long long)M * Cout * 2 < (1ll << 31);      // bit 8: the residual layers too (conv3 of res3 / res5, FPN laterals)
        else take = K >= (stride == 1 ? 512 : 256);
        if (take) return conv1x1_ring_launch(in, wgt, bias, res, res_mode, resH, resW, out, N, H, W, Ho, Wo, stride, M, K, Cout, out_stride, relu, st);
    }
    // 256 x 256 two-stage kernel: fp16 output, whole 256-channel tiles, a grid that fills the 256 CUs
    // (measured r01: +24 % on the K = 12544 FC GEMM, neutral-to-negative on the convolutions -> long-K GEMMs only;
    //  policy bit 4 forces it everywhere it applies, for A/B runs)
    if ((policy & 8) && !out_f32 && Cout % 256 == 0 && cout_store == Cout &&
        (long long)pe::ceil_div(M, 256) * (Cout / 256) >= 224 && ((policy & 16) || (!mode3x3 && K >= 4096)))
        return mode3x3 ? launch_big<MODE_3X3>(a, st) : launch_big<MODE_1X1>(a, st);
    if (mode3x3 && g_conv3x3_reuse.load(std::memory_order_relaxed)) {
        if (narrow) return launch3x3r<128, 64>(a, st);
        // 256-row tiles (8 waves) halve the weight-tile traffic per flop; keep 128 when the grid would not fill the chip
        const bool big = (policy & 1) && (long long)pe::ceil_div(M, 256) * pe::ceil_div(Cout, 128) >= 512;
        return big ? launch3x3r<256, 128>(a, st) : launch3x3r<128, 128>(a, st);
    }
    if (mode3x3) return narrow ? launch2<128, 64, MODE_3X3>(a, st) : launch2<128, 128, MODE_3X3>(a, st);
    if (narrow) return launch2<128, 64, MODE_1X1>(a, st);
    const bool big1 = (policy & 2) && (long long)pe::ceil_div(M, 256) * pe::ceil_div(Cout, 128) >= 512;
    if (policy & 4) return big1 ? launch2<256, 128, MODE_1X1, 2>(a, st) : launch2<128, 128, MODE_1X1, 2>(a, st);
    return big1 ? launch2<256, 128, MODE_1X1>(a, st) : launch2<128, 128, MODE_1X1>(a, st);
}
}  // namespace pe

// Measurement / test hook - deliberately NOT part of include/proben_hip.h (csrc/test_hooks.h): selects among kernels that all
// compute the same convolution, so that tests can cover every variant and scripts/ablate_conv.py can A/B them.
extern "C" int pe_test_set_conv_policy(int tile_bits, int reuse3x3) {
    pe::g_conv_tile256 = tile_bits;
    pe::g_conv3x3_reuse = reuse3x3 ? 1 : 0;
    return PE_OK;
}

