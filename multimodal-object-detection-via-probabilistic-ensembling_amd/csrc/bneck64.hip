// One launch for the second half of a 64-channel-wide BottleneckBlock AND the first convolution of the next block (res2 of
// ResNet-50/101: backbone/resnet.py:107-221, stride-1 blocks with bottleneck_channels = 64, out_channels = 256):
//
//     t2   = relu(conv2_3x3(t1) + b2)                       64 -> 64      t1 = relu(conv1(x)) of THIS block (input, fp16)
//     out  = relu(conv3_1x1(t2) + b3 + shortcut)            64 -> 256     shortcut = x (identity) or conv_1x1(s) + bsc (first block)
//     t1n  = relu(conv1_next_1x1(out) + b1n)                256 -> 64     optional: the NEXT block's conv1 (written next to out)
//
// Why: at 200 x 256 pixels every one of these layers is HBM-bound (1.7 M pixels x 128 .. 512 B).  Unfused, a block moves
// t2 out and back (2 x 210 MB), the shortcut is written by its own launch and read back (2 x 839 MB in the first block), and
// the next conv1 re-reads the 839 MB block output that was just written.  Here each pixel's t1 row is read once (+ halo),
// the shortcut source once, and out / t1n are written once; nothing else touches HBM.
//
// How: a workgroup owns a 4-row x 64-column pixel tile; t1 with a one-pixel halo (6 x 66 rows of 128 B, padded to 144 B) is
// copied to LDS once - with 64 input channels that slab is the WHOLE K extent of the 3x3, so the 36 K-steps run without a
// barrier.  Wave w owns image row w of the tile (two 32-pixel blocks) and ALL channels of its pixels, so every later stage
// is register-direct: the MFMA computes D[channel][pixel] with the channel permutation of csrc/conv_wd.h (a lane ends up
// with 32 consecutive channels of one pixel); converted to fp16, those accumulators ARE the B fragments of the next 1x1
// when its weights are packed in the matching K order (K-step j, lane half h <-> channels 32 h + 8 j .. + 8).  The shortcut
// is added last, for every pixel alike, so a pixel's fp32 summation order does not depend on where it sits in the tile or
// the batch.  Weights (pe_bneck64_pack: one fragment-ordered stream for all stages) go L2 -> LDS ring -> A fragments.
#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned int uint4v __attribute__((__vector_size__(16)));

// MFMA output row rho of 32-row block blk <-> channel of a 64-channel group (the permutation of csrc/conv_wd.h: a lane's
// 2 x 16 accumulator registers are 32 consecutive channels of its pixel)
__host__ __device__ inline int cout_perm(int blk, int rho) { return ((rho >> 2) & 1) * 32 + blk * 16 + (rho >> 3) * 4 + (rho & 3); }

constexpr int TR = 4, TC = 64;                 // output tile: rows x columns
constexpr int SROW = 144;                      // slab row: 128 B of channels + 16 B pad
constexpr int SCOLS = TC + 2, SROWS = TR + 2;
constexpr int SLAB_BYTES = SROWS * SCOLS * SROW;   // 57 024
constexpr int WCHUNK = 8192;                   // weight ring slot: 4 K-steps x 2 KiB
constexpr int LDS_BYTES = SLAB_BYTES + 2 * WCHUNK;   // 73 408: two workgroups per CU
constexpr int P1_STEPS = 36;                   // 9 taps x 4 K-steps of 16 channels
constexpr int P1_HALFS = P1_STEPS * 2 * 512;   // packed 3x3 stream (fp16 elements)

struct B64Args {
    const _Float16* t1;      // [N,H,W,64]
    const _Float16* res;     // identity: [N,H,W,256]; shortcut convolution: its input [N,H,W,64]
    const _Float16* wpk;     // pe_bneck64_pack
    const float* b2;         // [64]
    const float* b3;         // [256]
    const float* bsc;        // [256] or null
    const float* b1n;        // [64] or null
    _Float16* out;           // [N,H,W,256]
    _Float16* t1n;           // [N,H,W,64] or null
    int N, H, W;
    int tiles_x, tiles_y;
};

__device__ __forceinline__ half8 bload(const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) {
    return __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

typedef float float8v __attribute__((ext_vector_type(8)));
// 2 x 16 consecutive floats at two wave-uniform addresses through the scalar cache, one wait (the twin of conv_wd.h's wd_sload4)
__device__ __forceinline__ void b64_sload4(float8v& a0, float8v& a1, float8v& b0, float8v& b1, const float* pa, const float* pb) {
    auto uni = [](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %5, 0x0\n\ts_load_dwordx8 %3, %5, 0x20\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(a0), "=&s"(a1), "=&s"(b0), "=&s"(b1)
                 : "s"(uni(pa)), "s"(uni(pb)));
}
// the 16 bias values of accumulator block `blk` of this lane (lane half h owns channels [32 h + 16 blk, +16) of the 64 at `p`)
__device__ __forceinline__ float16v b64_bias16(const float* p, int blk, int h) {
    float8v s0, s1, s2, s3;
    b64_sload4(s0, s1, s2, s3, p + blk * 16, p + 32 + blk * 16);
    float16v b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        b[e] = h ? s2[e] : s0[e];
        b[e + 8] = h ? s3[e] : s1[e];
    }
    return b;
}

template <bool SC, bool NEXT>
__global__ __launch_bounds__(256, 2) void bneck64_kernel(B64Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SPC = 4 + (SC ? 4 : 0) + (NEXT ? 4 : 0);   // phase-2 K-steps per 64-channel output chunk
    constexpr int P2_TOTAL = 4 * SPC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, px = lane & 31;

    const int nwg = a.N * a.tiles_y * a.tiles_x;
    int bid = blockIdx.x;
    {   // contiguous runs of tiles per XCD (neighbouring tiles share halo rows in that XCD's L2)
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tx = bid % a.tiles_x, ty = (bid / a.tiles_x) % a.tiles_y, n = bid / (a.tiles_x * a.tiles_y);
    const int y0 = ty * TR, x0 = tx * TC;
    const long long img = (long long)n * a.H * a.W;
    const long long M = (long long)a.N * a.H * a.W;

    // ---- phase 0: t1 tile + halo -> LDS (zeros outside the image: out-of-range buffer offsets) ----
    const __amdgpu_buffer_rsrc_t rt1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.t1), 0, (int)(M * 128), 0x00020000);
    {
        constexpr int PIECES = SROWS * SCOLS * 8, NP = (PIECES + 255) / 256;
        half8 v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * 256;
            const int e = q >> 3, c = q & 7;
            const int row = e / SCOLS, col = e - row * SCOLS;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            const bool ok = q < PIECES && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            v[i] = bload(rt1, ok ? (unsigned)(((img + (long long)gy * a.W + gx) * 64 + c * 8) * 2) : 0xFFFFFFF0u, 0);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int q = tid + i * 256;
            if (q < PIECES) *reinterpret_cast<half8*>(smem + (q >> 3) * SROW + (q & 7) * 16) = v[i];
        }
    }

    // ---- per-lane pixels: wave = tile row, pixel block i = columns 32 i .. 32 i + 31 ----
    const int gy = y0 + wave;
    bool pvalid[2];
    unsigned poff[2];   // pixel index (within the whole tensor) or an out-of-range marker
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int gx = x0 + i * 32 + px;
        pvalid[i] = gy < a.H && gx < a.W;
        poff[i] = pvalid[i] ? (unsigned)(img + (long long)gy * a.W + gx) : 0x7FFFFFFu;
    }
    // shortcut source: identity -> 64 bytes per (lane, pixel block, output chunk), prefetched one chunk ahead;
    // shortcut convolution -> the 64 channels of s at the lane's pixels as four B fragments, loaded once
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res), 0, (int)(M * (SC ? 128 : 512)), 0x00020000);
    half8 rv[2][4];
    auto res_load = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned vo = pvalid[i] ? (SC ? poff[i] * 128u + h * 64 + q * 16 : poff[i] * 512u + c * 128 + h * 64 + q * 16) : 0xFFFFFFF0u;
                rv[i][q] = bload(rres, vo, 0);
            }
    };
    res_load(0);

    // ---- phase 1: 3x3, 64 -> 64.  acc2[blk][i]: lane holds channels 32 h + 16 blk + r of pixel (i, px) ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wpk), 0, (P1_HALFS + 4 * SPC * 1024) * 2, 0x00020000);
    float16v acc2[2][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        float16v b;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 v = *reinterpret_cast<const float4*>(a.b2 + h * 32 + blk * 16 + r4 * 4);
            b[r4 * 4 + 0] = v.x; b[r4 * 4 + 1] = v.y; b[r4 * 4 + 2] = v.z; b[r4 * 4 + 3] = v.w;
        }
        acc2[blk][0] = b; acc2[blk][1] = b;
    }
    // Weight stream: L2 -> registers -> LDS ring -> A fragments.  All four waves (and every pixel block) consume the same
    // 2 KiB per K-step, and a wave has only 4 MFMAs per K-step to hide an L2 round trip behind, so the workgroup fetches each
    // 8 KiB chunk (4 K-steps) once, cooperatively, four chunks ahead (three register stages = 24 registers per lane), and parks
    // it in a two-slot LDS ring; one barrier per chunk both publishes chunk k + 1 and retires the reads of chunk k - 1.
    constexpr int TOTAL = P1_STEPS + P2_TOTAL, NCHUNK = TOTAL / 4;
    static_assert(TOTAL % 4 == 0, "whole chunks");
    unsigned char* wlds = smem + SLAB_BYTES;
    half8 wreg[3][2];
    auto wg_load = [&](int stage, int chunk) {
        if (chunk < NCHUNK) {
            wreg[stage][0] = bload(rw, tid * 16, chunk * WCHUNK);
            wreg[stage][1] = bload(rw, tid * 16 + 4096, chunk * WCHUNK);
        }
    };
    auto wg_store = [&](int stage, int chunk) {
        if (chunk < NCHUNK) {
            *reinterpret_cast<half8*>(wlds + (chunk & 1) * WCHUNK + tid * 16) = wreg[stage][0];
            *reinterpret_cast<half8*>(wlds + (chunk & 1) * WCHUNK + tid * 16 + 4096) = wreg[stage][1];
        }
    };
    // called at the first K-step of chunk k (every wave, same sequence)
    auto chunk_begin = [&](int k) {
        __syncthreads();
        wg_store((k + 1) % 3, k + 1);
        wg_load((k + 1) % 3, k + 4);   // straight back into the stage just written out: three chunk periods of latency slack
    };
    auto a_frag = [&](int g, int blk) {   // A fragment of global K-step g
        return *reinterpret_cast<const half8*>(wlds + ((g >> 2) & 1) * WCHUNK + (g & 3) * 2048 + blk * 1024 + lane * 16);
    };
    wg_load(0, 0); wg_load(1, 1); wg_load(2, 2);
    wg_store(0, 0);
    wg_load(0, 3);
    int fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) fb[i] = (wave * SCOLS + i * 32 + px) * SROW + h * 64;
#pragma unroll
    for (int s = 0; s < P1_STEPS; ++s) {
        const int tap = s >> 2, j = s & 3;
        const int kh = tap / 3, kw = tap - kh * 3;
        if (j == 0) chunk_begin(s >> 2);   // the first one also publishes the slab
        half8 bf[2], af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) bf[i] = *reinterpret_cast<const half8*>(smem + fb[i] + (kh * SCOLS + kw) * SROW + j * 16);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) af[blk] = a_frag(s, blk);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc2[blk][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[blk], bf[i], acc2[blk][i], 0, 0, 0);
    }
    // t2 as B fragments: K-step j of the following 1x1s <-> channels 32 h + 8 j .. + 8 = acc2[j >> 1][i][8 (j & 1) ..]
    half8 t2f[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) t2f[i][j][e] = (_Float16)pe::relu_nan(acc2[j >> 1][i][(j & 1) * 8 + e]);

    // ---- phase 2: per 64-channel chunk of the 256 outputs: conv3 (+ shortcut convolution), shortcut, ReLU, store,
    //      and the chunk's contribution to the next block's conv1 ----
    float16v acc4[2][2];
    if (NEXT) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float16v b;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = *reinterpret_cast<const float4*>(a.b1n + h * 32 + blk * 16 + r4 * 4);
                b[r4 * 4 + 0] = v.x; b[r4 * 4 + 1] = v.y; b[r4 * 4 + 2] = v.z; b[r4 * 4 + 3] = v.w;
            }
            acc4[blk][0] = b; acc4[blk][1] = b;
        }
    }
    half8 sfr[2][4];
    if (SC) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) sfr[i][q] = rv[i][q];
    }
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(M * 512), 0x00020000);
    unsigned char* stg = smem + wave * (64 * SROW);       // 64 pixel rows of 128 B (+ pad) per wave, inside the dead slab
    // staging row k * 8 + lane / 8 = pixel column x0 + that index of this wave's image row
    const unsigned srow0 = (unsigned)(img + (long long)gy * a.W + x0) + (lane >> 3);
    const int scol0 = gy < a.H ? x0 + (lane >> 3) : 0x40000000;
    auto srow_off = [&](int k, unsigned bytes_per_px, unsigned col_bytes) {
        return scol0 + k * 8 < a.W ? (srow0 + k * 8) * bytes_per_px + col_bytes + (lane & 7) * 16 : 0xFFFFFFF0u;
    };
    auto stage_rows = [&](const half8 (&v)[2][4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<half8*>(stg + (i * 32 + px) * SROW + h * 64 + q * 16) = v[i][q];
    };
    int s2 = 0;   // running phase-2 step (compile-time after unrolling): stream position P1_STEPS + s2
    auto step_mfma = [&](int sidx, float16v (&acc)[2][2], const half8& bf0, const half8& bf1) {
        const int g = P1_STEPS + sidx;
        if ((g & 3) == 0) chunk_begin(g >> 2);
        const half8 a0 = a_frag(g, 0), a1 = a_frag(g, 1);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf1, acc[1][1], 0, 0, 0);
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float16v acc3[2][2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float16v b;
            // Round 5: the chunk's biases through the scalar cache.  As vector loads they were the youngest vector-memory operations at
            // the chunk's first MFMA: waiting for them drained the previous chunk's line stores and the prefetched shortcut (`vmcnt(0)`) in
            // a kernel that is HBM-bound.  Same bits; -2..-4 % on the two NEXT forms, +1 % on the last block's (profiles/r05_b64_bias_ab.txt).
            b = b64_bias16(a.b3 + c * 64, blk, h);
            if (SC) {
                const float16v u = b64_bias16(a.bsc + c * 64, blk, h);
#pragma unroll
                for (int e = 0; e < 16; ++e) b[e] += u[e];
            }
            acc3[blk][0] = b; acc3[blk][1] = b;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { step_mfma(s2, acc3, t2f[0][j], t2f[1][j]); ++s2; }
        if (SC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { step_mfma(s2, acc3, sfr[0][j], sfr[1][j]); ++s2; }
        }
        // shortcut (identity), ReLU, fp16: of[i][q] = channels 64 c + 32 h + 8 q .. + 8 of pixel (i, px)
        half8 of[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc3[q >> 1][i][(q & 1) * 8 + e];
                    if (!SC) v += (float)rv[i][q][e];
                    of[i][q][e] = (_Float16)pe::relu_nan(v);
                }
        if (!SC && c + 1 < 4) res_load(c + 1);   // next chunk's shortcut rows: in flight across the stores and the MFMAs below
        // Stores: a lane owns 64 bytes of one pixel, so a direct store instruction is 64 scattered 16-byte pieces - partial
        // line writes that cost ~0.1 ms per launch against whole lines (measured).  The chunk goes through a wave-private LDS
        // patch (the slab is dead in this phase) and leaves as whole 128-byte rows: lane l -> row l / 8, piece l % 8.
        stage_rows(of);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const half8 v = *reinterpret_cast<const half8*>(stg + (k * 8 + (lane >> 3)) * SROW + (lane & 7) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rout, srow_off(k, 512u, c * 128), 0, 0);
        }
        if (NEXT) {   // B fragments of the chunk: read back from the staging patch (the lane's own 16-byte pieces) rather than kept in registers
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const half8 f0 = *reinterpret_cast<const half8*>(stg + px * SROW + h * 64 + j * 16);
                const half8 f1 = *reinterpret_cast<const half8*>(stg + (32 + px) * SROW + h * 64 + j * 16);
                step_mfma(s2, acc4, f0, f1);
                ++s2;
            }
        }
    }
    if (NEXT) {
        const __amdgpu_buffer_rsrc_t rt1n = __builtin_amdgcn_make_buffer_rsrc(a.t1n, 0, (int)(M * 128), 0x00020000);
        half8 tn[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) tn[i][q][e] = (_Float16)pe::relu_nan(acc4[q >> 1][i][(q & 1) * 8 + e]);
        stage_rows(tn);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const half8 v = *reinterpret_cast<const half8*>(stg + (k * 8 + (lane >> 3)) * SROW + (lane & 7) * 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rt1n, srow_off(k, 128u, 0), 0, 0);
        }
    }
}

// A-fragment record: 64 lanes x 8 halfs; lane l = output row cout_perm(blk, l & 31) of a 64-row group, K values
// kbase + 32 (l >> 5) + 8 j + (0..7) of the source row.  One thread per (record, lane).
__global__ void bneck64_pack_kernel(const _Float16* w2, const _Float16* w3, const _Float16* wsc, const _Float16* w1n, _Float16* out, int spc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = (P1_STEPS * 2 + 4 * spc * 2) * 64;
    if (t >= total) return;
    const int l = t & 63, rec = t >> 6;
    const int blk = rec & 1, step = rec >> 1;
    const int rho = l & 31, hh = l >> 5;
    const _Float16* src;
    if (step < P1_STEPS) {          // 3x3: weight [64][3][3][64] = [64][576], K-step = tap * 4 + j
        const int tap = step >> 2, j = step & 3;
        src = w2 + (size_t)cout_perm(blk, rho) * 576 + tap * 64 + hh * 32 + j * 8;
    } else {
        const int s = step - P1_STEPS, c = s / spc, k = s - c * spc;
        const int j = k & 3, kind = k >> 2;                    // 0: conv3, then (shortcut conv), then (next conv1)
        const bool is_next = (kind == 2) || (kind == 1 && wsc == nullptr);
        if (!is_next) {             // [256][64]: rows of output chunk c
            src = (kind == 0 ? w3 : wsc) + (size_t)(c * 64 + cout_perm(blk, rho)) * 64 + hh * 32 + j * 8;
        } else {                    // next conv1 [64][256]: K slice = output chunk c
            src = w1n + (size_t)cout_perm(blk, rho) * 256 + c * 64 + hh * 32 + j * 8;
        }
    }
    *reinterpret_cast<half8*>(out + (size_t)t * 8) = *reinterpret_cast<const half8*>(src);
}

}  // namespace

extern "C" size_t pe_bneck64_packed_bytes(int32_t has_shortcut_conv, int32_t has_next) {
    const int spc = 4 + (has_shortcut_conv ? 4 : 0) + (has_next ? 4 : 0);
    return ((size_t)P1_HALFS + (size_t)4 * spc * 1024) * 2;
}

extern "C" int pe_bneck64_pack(const void* w2, const void* w3, const void* wsc, const void* w1n, void* packed, void* stream) {
    PE_CHECK_ARG(w2 && w3 && packed, "pe_bneck64_pack: null pointer");
    const int spc = 4 + (wsc ? 4 : 0) + (w1n ? 4 : 0);
    const int total = (P1_STEPS * 2 + 4 * spc * 2) * 64;
    hipLaunchKernelGGL(bneck64_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const _Float16*)w2, (const _Float16*)w3,
                       (const _Float16*)wsc, (const _Float16*)w1n, (_Float16*)packed, spc);
    PE_CHECK_LAUNCH("pe_bneck64_pack");
    return PE_OK;
}

extern "C" int pe_bneck64_f16(const void* t1, const void* shortcut_src, const void* packed, const float* bias2, const float* bias3,
                              const float* bias_sc, const float* bias1n, void* out, void* t1_next, int32_t N, int32_t H, int32_t W,
                              int32_t has_shortcut_conv, int32_t has_next, void* stream) {
    PE_CHECK_ARG(t1 && shortcut_src && packed && bias2 && bias3 && out, "pe_bneck64_f16: null pointer");
    PE_CHECK_ARG(!has_shortcut_conv || bias_sc, "pe_bneck64_f16: shortcut convolution needs its bias");
    PE_CHECK_ARG(!has_next || (bias1n && t1_next), "pe_bneck64_f16: next conv1 needs its bias and output");
    PE_CHECK_ARG(N > 0 && H > 0 && W > 0, "pe_bneck64_f16: bad dims");
    const long long M = (long long)N * H * W;
    PE_CHECK_ARG(M * 512 < (1ll << 31), "pe_bneck64_f16: tensors larger than 2 GiB (32-bit buffer offsets)");
    B64Args a{};
    a.t1 = (const _Float16*)t1; a.res = (const _Float16*)shortcut_src; a.wpk = (const _Float16*)packed;
    a.b2 = bias2; a.b3 = bias3; a.bsc = bias_sc; a.b1n = bias1n; a.out = (_Float16*)out; a.t1n = (_Float16*)t1_next;
    a.N = N; a.H = H; a.W = W; a.tiles_x = pe::ceil_div(W, TC); a.tiles_y = pe::ceil_div(H, TR);
    const dim3 grid((unsigned)(N * a.tiles_x * a.tiles_y)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define PE_B64_LAUNCH(SC, NX)                                                                                                       \
    do {                                                                                                                            \
        PE_ENSURE_LDS((bneck64_kernel<SC, NX>), LDS_BYTES, "pe_bneck64_f16");                                                            \
        hipLaunchKernelGGL((bneck64_kernel<SC, NX>), grid, block, LDS_BYTES, st, a);                                               \
    } while (0)
    if (has_shortcut_conv && has_next) PE_B64_LAUNCH(true, true);
    else if (has_shortcut_conv) PE_B64_LAUNCH(true, false);
    else if (has_next) PE_B64_LAUNCH(false, true);
    else PE_B64_LAUNCH(false, false);
#undef PE_B64_LAUNCH
    PE_CHECK_LAUNCH("pe_bneck64_f16");
    return PE_OK;
}
