// 1x1 convolution (stride 1, no residual, fp16 out) as a PERSISTENT loader / consumer kernel: round 5's answer to "52 % of the 1x1
// class's wave cycles wait" (VERDICT r04 item 1b).
//
// Replaces, for the layers it takes, the same reference rows as csrc/conv_igemm2.hip: layers/wrappers.py:62-98 (Conv2d.forward) +
// layers/batch_norm.py:45-65 (FrozenBatchNorm2d, folded into weight / bias) + relu_ - backbone/resnet.py:205-221 `conv1` of a
// bottleneck block (1024 -> 256 in res4: 44 launches per frame-pair step), fpn.py:129 lateral of the top level.
//
// Why a second 1x1 kernel.  conv_igemm2_kernel<128,128,1x1> runs DMA -> vmcnt(0) -> barrier -> 16 MFMA -> barrier per K-step with three
// workgroups per CU covering each other; on res4 conv1 (210 MB in, 52 MB out) every variant of that structure measured 73-75 us =
// 3.5 TB/s (profiles/r02_conv3x3_wd_ablation.txt).  Little's law says why: a CU then has ~24 KiB of UNIQUE pixel bytes in flight at its
// best moment (3 workgroups x 16 KiB, but the two Cout tiles of a pixel tile fetch the same lines) and about half of that on average,
// against the ~28 KiB that 25 GB/s per CU x 1.1 us of loaded HBM latency needs (MI355X_MICROARCH.md, ldsdma-fill).  Here:
//   * ONE workgroup per CU, 256 (or fewer) workgroups walk the launch: workgroup w owns a contiguous run of 32-pixel blocks, cut into
//     tiles of up to 128 pixels x 256 output channels (every pixel line is fetched from HBM once per 256 output channels);
//   * wave 8 is the PIXEL loader: `buffer_load_dwordx4 ... lds` into a 4-stage ring of 16 KiB K-slabs (64 channels of 128 pixels),
//     three stages (48 KiB) ahead of the consumers; wave 9 is the WEIGHT loader: a 3-stage ring of 32 KiB (L2-resident), two ahead.
//     Both rings run on across tile boundaries, so the next tile's first slabs land while the consumers store this tile;
//   * waves 0-7 (2 x 4, each 64 pixels x 64 channels = four 32x32 accumulators) issue NO vector-memory instruction in the K-loop:
//     the loaders eat the 100-170-cycle issue stalls of a loaded memory pipe, the matrix waves only ds_read + MFMA;
//   * one s_barrier per K-step: it publishes stage s (the loaders waited for their own vmcnt) and tells the loaders that every consumer
//     is done with stage s - 1, which is exactly the ring slot their next DMA overwrites;
//   * out-of-run / out-of-image pixel rows use an out-of-range buffer offset: the bounds check returns zeros and fetches nothing;
//   * the MFMA takes the WEIGHT fragment as A and the pixel fragment as B, and weight row perm(i) feeds MFMA row i, so a lane ends up
//     with 16 CONSECUTIVE output channels of one pixel; the epilogue turns that into whole 128-byte lines through a wave-private patch
//     of the weight slot the tile has just finished with (all 160 KiB belong to the rings; one extra barrier per tile frees the slot).
// Same products in the same K order with the same zero-initialised fp32 accumulators and bias added in the epilogue as
// conv_igemm2_kernel: the results are bit-identical (tests/test_ops_gpu.py::test_conv1x1_ring_kernel_matches_torch_and_conv_igemm2_bit_for_bit), so which kernel takes
// a launch never shows in a frame's result.
#include <atomic>

#include "conv_common.h"

namespace {

typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int RG_BM = 128, RG_BN = 256;
constexpr int RG_ASTAGES = 4, RG_BSTAGES = 3;
constexpr int RG_ASTAGE_B = RG_BM * ROW_B;               // 16 KiB
constexpr int RG_BSTAGE_B = RG_BN * ROW_B;               // 32 KiB
constexpr int RG_BBASE = RG_ASTAGES * RG_ASTAGE_B;       // 64 KiB
constexpr int RG_LDS = RG_BBASE + RG_BSTAGES * RG_BSTAGE_B;   // 160 KiB: the whole CU
constexpr int RG_CONSUMERS = 8, RG_THREADS = (RG_CONSUMERS + 2) * 64;
static_assert(RG_LDS == 160 * 1024, "the rings are sized to the CU's LDS");

struct RingArgs {
    const _Float16* in;
    const _Float16* wgt;
    const float* bias;
    _Float16* out;
    int M, K, Cout, out_stride, relu;
    int nblk;      // 32-pixel blocks in the launch
    int tiles_n;   // Cout / 256
    int stride, H, W, Ho, Wo;
    const _Float16* res;      // residual (conv_igemm2's modes): 1 = [M, Cout] like the output, 2 = FPN top-down: [N, resH, resW, Cout], pixel (oh >> 1, ow >> 1)
    int res_mode, resH, resW;
    unsigned res_bytes;
    unsigned in_bytes;           // stride 2 (shortcut convolutions): output pixel (n, oh, ow) reads input pixel (n, 2 oh, 2 ow) of [N, H, W, K]
#ifdef PE_LAB
    int abl;       // LAB builds only (`python -m proben_amd.build --lab` -> libproben_hip_lab.so; results wrong): 1 = no pixel DMA, 2 = no weight DMA,
                   // 4 = no ds_read / MFMA, 8 = no stores, 16 = pixel slabs fetched as if the input were K-chunk-major [K / 64][M][64]
#endif
};
// The measurement switches of DESIGN 8.6's ablation budget exist in the lab library only: the product object contains no such branch
// (tests/test_build_audit.py::test_product_library_has_no_measurement_switches).
#ifdef PE_LAB
#define RG_ABL(bits) ((a.abl & (bits)) != 0)
#else
#define RG_ABL(bits) false
#endif

__device__ __forceinline__ int4v rg_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    int4v r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// eight LDS-DMA pieces (64 lanes x 16 B each, lane-linear) at LDS byte addresses lds, lds + 1 KiB, ...; v0..v7 = per-lane byte offsets
// into the buffer (>= 2 GiB: out of range -> the bounds check returns zeros), soff = wave-uniform byte offset.  M0 is compiler-reserved:
// saved and restored inside the statement.
__device__ __forceinline__ void rg_dma8(unsigned lds, int4v rsrc, unsigned soff, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                        unsigned v4, unsigned v5, unsigned v6, unsigned v7) {
    unsigned keep;
    lds = __builtin_amdgcn_readfirstlane(lds);       // wave-uniform by construction; the "s" constraint needs the compiler to know it
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %4, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %5, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %6, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %7, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %8, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %9, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %10, %2, %3 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %11, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds), "s"(rsrc), "s"(soff), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7)
        : "scc");
}

constexpr unsigned RG_OOB = 0x80000000u;

// The bias of a wave's 64 channels through the SCALAR cache (the address is wave-uniform): eight s_load_dwordx8 issued together,
// one wait.  asm, because the compiler will not prove that the kernel's own stores leave the bias alone and falls back to vector
// loads - which queue behind the loaders' DMA stream with every consumer waiting (~1-2 us per tile).
typedef float float8v __attribute__((ext_vector_type(8)));
struct RgBias { float8v v[8]; };      // v[4 j + q] = channels [32 j + 8 q, +8)
__device__ __forceinline__ void rg_bias_issue(RgBias& b, const float* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long u = ((unsigned long long)hi << 32) | lo;
    asm volatile(
        "s_load_dwordx8 %0, %8, 0x0\n\ts_load_dwordx8 %1, %8, 0x20\n\ts_load_dwordx8 %2, %8, 0x40\n\ts_load_dwordx8 %3, %8, 0x60\n\t"
        "s_load_dwordx8 %4, %8, 0x80\n\ts_load_dwordx8 %5, %8, 0xa0\n\ts_load_dwordx8 %6, %8, 0xc0\n\ts_load_dwordx8 %7, %8, 0xe0"
        : "=&s"(b.v[0]), "=&s"(b.v[1]), "=&s"(b.v[2]), "=&s"(b.v[3]), "=&s"(b.v[4]), "=&s"(b.v[5]), "=&s"(b.v[6]), "=&s"(b.v[7])
        : "s"(u));
}
__device__ __forceinline__ void rg_bias_wait(RgBias& b) {      // every later use of b depends on this statement
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(b.v[0]), "+s"(b.v[1]), "+s"(b.v[2]), "+s"(b.v[3]), "+s"(b.v[4]), "+s"(b.v[5]), "+s"(b.v[6]), "+s"(b.v[7]));
}

// MFMA row i of a 32-channel block <- weight row perm(i): lane half h then holds channels 16 h + e in accumulator register e
__device__ __forceinline__ int rg_perm(int i) { return ((i >> 2) & 1) * 16 + (i >> 3) * 4 + (i & 3); }

// One K-step of a wave = four MFMA K-steps of 16: the fragments of the first TWO are requested at once (nothing of the step can be read
// before its barrier), the third / fourth go into the first / second set's registers behind its MFMAs (32 fragment VGPRs: with 64
// accumulators and the bias in registers the 168-register budget of three waves per SIMD has no room for all four sets).
template <int TMI>
__device__ __forceinline__ void rg_compute(float16v (&acc)[2][2], const unsigned char* pa, const unsigned char* pb, int fswa, int fswb,
                                           int fkh) {
    half8 pf[2][TMI], wf[2][2];
    auto load = [&](int ks, int buf) {
        const int cha = ((ks * 2 + fkh) ^ fswa) << 4;
        const int chb = ((ks * 2 + fkh) ^ fswb) << 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[buf][j] = *reinterpret_cast<const half8*>(pb + j * 32 * ROW_B + chb);
#pragma unroll
        for (int i = 0; i < TMI; ++i) pf[buf][i] = *reinterpret_cast<const half8*>(pa + i * 32 * ROW_B + cha);
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TMI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][j], pf[buf][i], acc[i][j], 0, 0, 0);
    };
    // sched_barrier(0): nothing crosses - left alone, the scheduler folds the two sets back into one and every MFMA K-step waits for
    // its own reads
    load(0, 0);
    load(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    load(2, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma(1);
    load(3, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    mma(1);
}

template <bool RELU, bool RES>      // compile-time: a run-time flag costs a v_cndmask per output element in a VALU-bound epilogue; RES: with a residual (its 32 registers only where needed)
__global__ __launch_bounds__(RG_THREADS) void conv1x1_ring_kernel(RingArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned smem_base = (unsigned)(unsigned long long)(lds_byte*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's run of 32-pixel blocks, as m-tiles of up to four blocks x tiles_n column tiles x K / 64 steps ----
    // Cout > 256 (tiles_n column tiles): when the grid allows it the column tiles of a run go to tiles_n workgroups of ONE XCD
    // (blockIdx % 8 is the XCD: workgroups w, w + 8, ... share it) that start together and walk the same pixels in step, so a pixel
    // line crosses the fabric once and the others hit the XCD's L2 - the run's pixels x K do not fit any cache between two passes.
    // Otherwise (small grids) one workgroup walks the column tiles of each m-tile one after the other.
    const bool npar = a.tiles_n > 1 && gridDim.x % (8 * a.tiles_n) == 0;
    const int G = npar ? gridDim.x / a.tiles_n : gridDim.x;                                   // runs
    const int run = npar ? (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * a.tiles_n)) : blockIdx.x;
    const int n0 = npar ? (blockIdx.x >> 3) % a.tiles_n : 0;                                  // first column tile of this workgroup
    const int nseq = npar ? 1 : a.tiles_n;                                                    // column tiles it walks per m-tile
    const int b0 = (int)((long long)run * a.nblk / G), b1 = (int)((long long)(run + 1) * a.nblk / G);
    const int nb = b1 - b0;
    if (nb <= 0) return;
    const int nmt = (nb + 3) >> 2;
    const int nk = a.K >> 6;
    const int S = nmt * nseq * nk;           // K-steps = barriers, the same number in every wave

    if (wave == RG_CONSUMERS) {
        // ================= pixel loader: ring of 4 x [128 px][64 ch], three stages ahead =================
        const int lrow = lane >> 3, lp = lane & 7;
        unsigned rel[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int r = q * 8 + lrow;
            rel[q] = (unsigned)((r * a.K + ((lp ^ ((r >> 1) & 7)) * 8)) * 2);
        }
        const int4v rs = rg_make_rsrc(a.in, (unsigned)a.in_bytes);
        unsigned srow[16];          // strided launches only
        int u_ks = 0, u_n = 0, u_mt = 0;
        auto issue = [&](int stage) {
            const int mrow0 = (b0 + 4 * u_mt) * 32;
            int vrows = 0;
            unsigned soff = 0;      // (mutable: the chunk-major ablation re-addresses the slab)
            if (u_mt < nmt) {
                const int nbt = nb - 4 * u_mt;
                vrows = (nbt < 4 ? nbt : 4) * 32;
                vrows = vrows < a.M - mrow0 ? vrows : a.M - mrow0;
                soff = (unsigned)(mrow0 * a.K + u_ks * 64) * 2u;
            }
            unsigned vo[16];
            if (a.stride == 1) {
#pragma unroll
                for (int q = 0; q < 16; ++q) vo[q] = (q * 8 + lrow < vrows) ? rel[q] : RG_OOB;
            } else {
                // strided launch: the rows of a tile are not one run of the input; their offsets change with the m-tile only
                if (u_ks == 0 && u_n == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int r = q * 8 + lrow, m = mrow0 + r;
                        const int ow = m % a.Wo, t = m / a.Wo;
                        const int oh = t % a.Ho, n = t / a.Ho;
                        srow[q] = r < vrows ? (unsigned)((((n * a.H + oh * a.stride) * a.W + ow * a.stride) * a.K + ((lp ^ ((r >> 1) & 7)) * 8)) * 2) : RG_OOB;
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) vo[q] = srow[q];
                soff = u_mt < nmt ? (unsigned)(u_ks * 128) : 0u;
            }
            const unsigned lds = smem_base + stage * RG_ASTAGE_B;
            if (RG_ABL(16)) {
                soff = u_mt < nmt ? (unsigned)((u_ks * a.M + mrow0) * 128) : 0u;
#pragma unroll
                for (int q = 0; q < 16; ++q) vo[q] = (q * 8 + lrow < vrows) ? (unsigned)(q * 1024 + lane * 16) : RG_OOB;
            }
            if (!RG_ABL(1)) {
                rg_dma8(lds, rs, soff, vo[0], vo[1], vo[2], vo[3], vo[4], vo[5], vo[6], vo[7]);
                rg_dma8(lds + 8192, rs, soff, vo[8], vo[9], vo[10], vo[11], vo[12], vo[13], vo[14], vo[15]);
            }
            if (++u_ks == nk) {
                u_ks = 0;
                if (++u_n == nseq) { u_n = 0; ++u_mt; }
            }
        };
        issue(0);
        issue(1);
        issue(2);
        int st = 3, c_ks = 0;
        for (int s = 0; s < S; ++s) {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");     // slab s has landed; slabs s + 1, s + 2 (16 pieces each) may be in flight
            __builtin_amdgcn_s_barrier();
            issue(st);                                             // slab s + 3 into the slot of slab s - 1 (steps beyond S: zeros, no traffic)
            st = (st + 1) & 3;
            if (++c_ks == nk) { c_ks = 0; __builtin_amdgcn_s_barrier(); }     // the consumers' end-of-tile barrier (see their epilogue)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no DMA may outlive the workgroup's LDS allocation
        return;
    }
    if (wave == RG_CONSUMERS + 1) {
        // ================= weight loader: ring of 3 x [256 ch][64 k], two stages ahead =================
        const int lrow = lane >> 3, lp = lane & 7;
        unsigned rel[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int r = q * 8 + lrow;
            rel[q] = (unsigned)((r * a.K + ((lp ^ ((r >> 1) & 7)) * 8)) * 2);
        }
        const int4v rs = rg_make_rsrc(a.wgt, (unsigned)a.Cout * (unsigned)a.K * 2u);
        int u_ks = 0, u_n = 0, u_mt = 0;
        auto issue = [&](int stage) {
            // The last two issues of a workgroup lie beyond its run (they only keep the vmcnt accounting uniform; nobody reads them): they
            // re-fetch the run's FIRST slab.  Not an out-of-range soffset - the buffer range check covers the per-lane offset only, an
            // SGPR offset of 2 GiB would be added to the base unchecked.
            const unsigned soff = u_mt < nmt ? (unsigned)((n0 + u_n) * RG_BN * a.K + u_ks * 64) * 2u : (unsigned)(n0 * RG_BN * a.K) * 2u;
            const unsigned lds = smem_base + RG_BBASE + stage * RG_BSTAGE_B;
            if (!RG_ABL(2)) {
                rg_dma8(lds, rs, soff, rel[0], rel[1], rel[2], rel[3], rel[4], rel[5], rel[6], rel[7]);
                rg_dma8(lds + 8192, rs, soff, rel[8], rel[9], rel[10], rel[11], rel[12], rel[13], rel[14], rel[15]);
                rg_dma8(lds + 16384, rs, soff, rel[16], rel[17], rel[18], rel[19], rel[20], rel[21], rel[22], rel[23]);
                rg_dma8(lds + 24576, rs, soff, rel[24], rel[25], rel[26], rel[27], rel[28], rel[29], rel[30], rel[31]);
            }
            if (++u_ks == nk) {
                u_ks = 0;
                if (++u_n == nseq) { u_n = 0; ++u_mt; }
            }
        };
        issue(0);
        issue(1);
        int st = 2, c_ks = 0;
        for (int s = 0; s < S; ++s) {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");     // slab s has landed; slab s + 1 (32 pieces) may be in flight
            __builtin_amdgcn_s_barrier();
            issue(st);                                             // slab s + 2 into the slot of slab s - 1
            st = st == 2 ? 0 : st + 1;
            if (++c_ks == nk) { c_ks = 0; __builtin_amdgcn_s_barrier(); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================= consumers: wave (wm, wn) owns pixels [64 wm, +64) x channels [64 wn, +64) of the tile =================
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, fkh = lane >> 5;
    const int prow = rg_perm(frow);
    const int fswa = (frow >> 1) & 7, fswb = (prow >> 1) & 7;      // (64 wm + 32 i + frow) >> 1 & 7 == (frow >> 1) & 7, same for the weights
    const unsigned char* la = smem + (wm * 64 + frow) * ROW_B;
    const unsigned char* lb = smem + RG_BBASE + (wn * 64 + prow) * ROW_B;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.out_stride * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.res ? a.res : a.in), 0, a.res ? (int)a.res_bytes : 0, 0x00020000);

    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Residual of the tile in ACCUMULATOR layout (a lane's 16 consecutive channels of its pixel = two 16-byte pieces per accumulator tile; loads
    // tolerate that pattern - the L1 merges a lane's pieces, DESIGN 8.3 - stores do not), requested under the tile's last K-step like the bias.
    half8 rv[RES ? 2 : 1][2][2];
    auto res_issue = [&](int nn, int mtile) {
        const int mrow0 = (b0 + 4 * mtile) * 32;
        const int nbt_ = nb - 4 * mtile;
        int vrows = (nbt_ < 4 ? nbt_ : 4) * 32;
        vrows = vrows < a.M - mrow0 ? vrows : a.M - mrow0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + frow, m = mrow0 + row;
            unsigned off = RG_OOB;
            if (row < vrows) {
                if (a.res_mode == 1) {
                    off = (unsigned)m * (unsigned)a.Cout * 2u;
                } else {
                    const int ow = m % a.Wo, t = m / a.Wo;
                    const int oh = t % a.Ho, im = t / a.Ho;
                    off = (unsigned)((im * a.resH + (oh >> 1)) * a.resW + (ow >> 1)) * (unsigned)a.Cout * 2u;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned co = (unsigned)(nn * RG_BN + wn * 64 + j * 32 + fkh * 16) * 2u;
                rv[i][j][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rres, off == RG_OOB ? RG_OOB : off + co, 0, 0));
                rv[i][j][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rres, off == RG_OOB ? RG_OOB : off + co + 16, 0, 0));
            }
        }
    };
    RgBias bs;       // live from the tile's last K-step to its epilogue only (a launch without a bias stays on conv_igemm2: conv2_dispatch)
    int sa = 0, sb = 0, ks = 0, n = n0, mt = 0;
    for (int s = 0; s < S; ++s) {
        __builtin_amdgcn_s_barrier();
        const int nbt = nb - 4 * mt;                 // pixel blocks of this m-tile (>= 4: a full tile)
        const int mine = nbt - 2 * wm;               // ... of which this wave owns min(2, mine)
        const unsigned char* pa = la + sa * RG_ASTAGE_B;
        const unsigned char* pb = lb + sb * RG_BSTAGE_B;
        if (ks == nk - 1) {
            rg_bias_issue(bs, a.bias + n * RG_BN + wn * 64);      // in flight under the tile's last K-step
            if constexpr (RES) res_issue(n, mt);
        }
        // ONE code path: a wave that owns one block of a partial tile multiplies the loader's zeros for the other (a second, 1-block
        // path makes the accumulators PHI values: 32 v_mov per K-step behind the MFMAs - measured +9 us per launch); a wave that owns
        // none skips the step, which leaves its SIMD to the other wave: a partial tile costs about half a tile either way
        if (mine >= 1 && !RG_ABL(4)) rg_compute<2>(acc, pa, pb, fswa, fswb, fkh);
        if (++ks == nk) {
            // ---- tile epilogue: + bias, ReLU, fp16, WHOLE 128-byte lines.  Stores straight from the accumulator layout are 16-byte
            // pieces 512 B apart - 64 write requests per instruction, measured at 17 of the launch's 78 us (profiles/r05_ring_abl_v1_piece_stores.txt).
            // After the end-of-tile barrier nobody reads the weight slot of the tile's last K-step any more, and its loader refills it
            // only behind the NEXT step's barrier, which the consumers reach after this epilogue: each wave transposes its 32 pixels x
            // 64 channels through a private 4 KiB patch of that slot (chunk c of row r in slot c ^ (r & 7)) and stores 8 whole lines
            // per instruction.
            __builtin_amdgcn_s_barrier();
            unsigned char* patch = smem + RG_BBASE + sb * RG_BSTAGE_B + wave * 4096;
            const int mrow0 = (b0 + 4 * mt) * 32;
            int vrows = (nbt < 4 ? nbt : 4) * 32;
            vrows = vrows < a.M - mrow0 ? vrows : a.M - mrow0;
            if (RG_ABL(8)) vrows = 0;
            const int chw = n * RG_BN + wn * 64;                    // this wave's first output channel
            rg_bias_wait(bs);
            const int rr = lane >> 3, rp = lane & 7;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    half8 h0, h1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float b0 = fkh ? bs.v[j * 4 + 2][e] : bs.v[j * 4 + 0][e];
                        const float b1 = fkh ? bs.v[j * 4 + 3][e] : bs.v[j * 4 + 1][e];
                        float x0 = acc[i][j][e] + b0, x1 = acc[i][j][e + 8] + b1;
                        if constexpr (RES) {      // (not unconditionally: -0.f + 0.f would change a sign bit)
                            x0 += (float)rv[i][j][0][e];
                            x1 += (float)rv[i][j][1][e];
                        }
                        if (RELU) { x0 = pe::relu_nan(x0); x1 = pe::relu_nan(x1); }
                        h0[e] = (_Float16)x0;
                        h1[e] = (_Float16)x1;
                    }
                    const int c0 = j * 4 + fkh * 2;                 // 16-byte chunk of this lane's first 8 channels within the wave's 128 B
                    *reinterpret_cast<half8*>(patch + frow * 128 + ((c0 ^ (frow & 7)) << 4)) = h0;
                    *reinterpret_cast<half8*>(patch + frow * 128 + (((c0 + 1) ^ (frow & 7)) << 4)) = h1;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = t * 8 + rr;                       // row of the patch; slot rp holds chunk rp ^ (r & 7) = rp ^ (rr & 7)
                    const half8 v = *reinterpret_cast<const half8*>(patch + r * 128 + (rp << 4));
                    const int row = wm * 64 + i * 32 + r;
                    // rows beyond this tile's pixels (another workgroup's, or beyond M) get an out-of-range offset: the store is dropped
                    const unsigned off = row < vrows ? (unsigned)(((mrow0 + row) * a.out_stride + chw + ((rp ^ (rr & 7)) << 3)) * 2) : RG_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rout, off, 0, 0);
                }
            }
            ks = 0;
            if (++n == n0 + nseq) { n = n0; ++mt; }
        }
        sa = (sa + 1) & 3;
        sb = sb == 2 ? 0 : sb + 1;
    }
}

}  // namespace

namespace pe {
std::atomic<int> g_ring_wgs{256};
#ifdef PE_LAB
std::atomic<int> g_ring_abl{0};
#endif

// true when the ring kernel can take the launch (the caller has checked: 1x1, stride 1, no residual, fp16 output)
bool conv1x1_ring_eligible(int M, int K, int Cout, int cout_store, int out_stride, long long in_pixels) {
    return Cout % RG_BN == 0 && cout_store == Cout && K % 64 == 0 && in_pixels * K * 2 < (1ll << 31) &&
           (long long)Cout * K * 2 < (1ll << 31) && (long long)M * out_stride * 2 < (1ll << 31);
}

int conv1x1_ring_launch(const void* in, const void* wgt, const float* bias, const void* res, int res_mode, int resH, int resW, void* out, int N,
                        int H, int W, int Ho, int Wo, int stride, int M, int K, int Cout, int out_stride, int relu, hipStream_t st) {
    RingArgs a{};
    a.in = (const _Float16*)in; a.wgt = (const _Float16*)wgt; a.bias = bias; a.out = (_Float16*)out;
    a.M = M; a.K = K; a.Cout = Cout; a.out_stride = out_stride; a.relu = relu;
    a.stride = stride; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.in_bytes = (unsigned)((long long)N * H * W * K * 2);
    a.res = res_mode ? (const _Float16*)res : nullptr; a.res_mode = res_mode; a.resH = resH; a.resW = resW;
    a.res_bytes = res_mode == 1 ? (unsigned)((long long)M * Cout * 2) : res_mode == 2 ? (unsigned)((long long)N * resH * resW * Cout * 2) : 0u;
    a.nblk = pe::ceil_div(M, 32);
    a.tiles_n = Cout / RG_BN;
#ifdef PE_LAB
    a.abl = g_ring_abl.load(std::memory_order_relaxed);
#endif
    const int wgs = g_ring_wgs.load(std::memory_order_relaxed);
    // Cout > 256: tiles_n workgroups per run when every run still has at least one m-tile's worth of blocks
    int grid = a.nblk < wgs ? a.nblk : wgs;
    if (a.tiles_n > 1 && wgs % (8 * a.tiles_n) == 0 && a.nblk >= wgs / a.tiles_n) grid = wgs;
#define PE_RING_LAUNCH(RELU, RES)                                                                                     \
    do {                                                                                                                  \
        PE_ENSURE_LDS((conv1x1_ring_kernel<RELU, RES>), (size_t)RG_LDS, "pe_conv2d_nhwc_f16(1x1 ring)");                    \
        hipLaunchKernelGGL((conv1x1_ring_kernel<RELU, RES>), dim3(grid), dim3(RG_THREADS), (size_t)RG_LDS, st, a);       \
    } while (0)
    if (relu && res_mode) PE_RING_LAUNCH(true, true);
    else if (relu) PE_RING_LAUNCH(true, false);
    else if (res_mode) PE_RING_LAUNCH(false, true);
    else PE_RING_LAUNCH(false, false);
#undef PE_RING_LAUNCH
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16(1x1 ring)");
    return PE_OK;
}
}  // namespace pe

// measurement hook (csrc/test_hooks.h): workgroups of the persistent 1x1 kernel (8 .. 256)
extern "C" int pe_test_set_ring_wgs(int wgs) {
    if (wgs >= 8 && wgs <= 1024) pe::g_ring_wgs = wgs;
    return PE_OK;
}
#ifdef PE_LAB
// ablation bits of the ring kernel (RingArgs::abl; results are WRONG for any non-zero value - scripts/archive/r05_ring_abl.py with the lab library only)
extern "C" int pe_test_set_ring_ablation(int bits) {
    pe::g_ring_abl = bits;
    return PE_OK;
}
#endif
