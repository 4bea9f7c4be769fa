// Weights-direct 3x3 convolution, fourth generation (round 4): one wave per SIMD, 256 accumulators in the accumulation half of
// the unified register file, persistent workgroups, the pixel slab filled by LDS-DMA.
//
// Same reference rows as csrc/conv_wd.h (layers/wrappers.py:62-98 Conv2d.forward + layers/batch_norm.py:45-65 folded + relu_;
// backbone/fpn.py:127-137 output convs, proposal_generator/rpn.py:74-85) and the same packed weight records
// (pe_conv_wd_pack_weights) in the same K order and summation order (accumulators start at zero, bias added in the epilogue), so
// results are BIT-IDENTICAL to conv3x3_wd_kernel<1,4,4,4>.
//
// What changes against conv_wd.h and why:
//   * a wave owns TPX = 8 pixel blocks x 64 output channels = 256 fp32 accumulators - twice the pixels per weight record, i.e.
//     HALF the L2 -> VGPR weight stream per MFMA, the biggest consumer of the CU's vector-memory issue path (bare loop: 1524-1569 vs
//     1350 TFLOP/s).  The accumulators are the AGPRs a[0:255], addressed LITERALLY from inline-asm MFMAs: the register allocator
//     never sees them (left to it, the 16 accumulator tuples crossing the tile loop cost 450 AGPR-to-AGPR moves and a round trip
//     through scratch memory per tile), and the 256 architectural VGPRs stay free for fragments, weight records and addresses;
//   * 256 accumulators mean one 4-wave workgroup per CU and nobody to cover a workgroup's dispatch / prologue / epilogue, so the
//     workgroup is PERSISTENT: <= 256 of them walk the tiles; the slab ring and the weight ring run on across the tile boundary;
//   * the slab (pixels + explicit zero halo) reaches LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write,
//     no vmcnt wait in the loop): LDS-DMA writes lane-linearly, so rows are dense 128 B and bank conflicts are avoided by the
//     XOR swizzle 16-byte-chunk c of entry e sits in slot c ^ ((e >> 1) & 7) - applied to the SOURCE address of the DMA and to
//     the fragment reads (the same involution on both sides).  Image-row segments start at multiples of 16 entries, so the
//     swizzle term of a lane is the same for all its pixel blocks and the block offset stays an immediate of ds_read_b128;
//   * out-of-image entries (halo columns, rows above / below the image, rows beyond M) use an out-of-range buffer offset: the
//     hardware bounds check returns zeros, which the DMA writes like data;
//   * the DMAs are issued from inline asm (the compiler would otherwise drain them with vmcnt(0) at every weight-record use) and
//     retired by one counted `s_waitcnt vmcnt` in front of the barrier that publishes the slab, a whole group after their issue;
//   * every asm statement is `volatile`: the compiler keeps memory operations on their side of it, so the MFMA / ds_read /
//     weight-load interleave of a K-step is the source order below, not a scheduler's choice.
#pragma once
#include <type_traits>
#include <utility>

#include "conv_wd.h"

namespace wd9 {
using wd::float16v;
using wd::half8;
typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

constexpr int WN = 4, THREADS = 256;

template <int SEGL, int TPX>
struct Geo {
    static constexpr int SEG = 1 << SEGL;          // image width = one slab segment
    static constexpr int BPX = TPX * 32;           // pixels per tile (whole image rows)
    static constexpr int NSEG = BPX / SEG;
    static constexpr int SEGP = SEG + 16;          // entries per segment: halo, SEG pixels, halo, 14 never-read pad entries
    static constexpr int E = (NSEG * SEGP + 31) / 32 * 32;   // entries incl. never-read padding: every wave moves whole 1 KiB pieces
    static constexpr int SLAB = E * 128;           // bytes per ring slot
    static constexpr int NPIECE = E / 8;           // 1 KiB pieces per slab, a multiple of 4
    static constexpr int PPW = (NPIECE + 3) / 4;   // pieces per wave and group, one per K-step
    static_assert(BPX % SEG == 0 && SEG >= 32 && NSEG >= 1, "tiles are whole image rows");
    static_assert(PPW <= 11, "one DMA piece per K-step, K-steps 0 .. 10 (the group's barrier sits in front of K-step 11)");
    static_assert(3 * SLAB + 16384 <= 160 * 1024, "three ring slots + the epilogue patches must fit the CU's LDS");
};

// ---- the accumulator file: a[LO : LO + 15] = accumulator block (blk, i) with LO = (blk * TPX + i) * 16 ----
template <int LO>
__device__ __forceinline__ void mfma_acc(half8 w, half8 p) {
    asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(w), "v"(p), "n"(LO), "n"(LO + 15));
}
template <int LO>
__device__ __forceinline__ void mfma_zero(half8 w, half8 p) {      // first K-step of a tile: C = 0
    asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, 0" ::"v"(w), "v"(p), "n"(LO), "n"(LO + 15));
}
// eight consecutive accumulators -> VGPRs (the caller has waited out the MFMA write latency)
template <int LO>
__device__ __forceinline__ void acc_read8(float (&x)[8]) {
    asm volatile(
        "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%9]\n\tv_accvgpr_read_b32 %2, a[%10]\n\tv_accvgpr_read_b32 %3, a[%11]\n\t"
        "v_accvgpr_read_b32 %4, a[%12]\n\tv_accvgpr_read_b32 %5, a[%13]\n\tv_accvgpr_read_b32 %6, a[%14]\n\tv_accvgpr_read_b32 %7, a[%15]"
        : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7])
        : "n"(LO), "n"(LO + 1), "n"(LO + 2), "n"(LO + 3), "n"(LO + 4), "n"(LO + 5), "n"(LO + 6), "n"(LO + 7));
}

// an MFMA whose accumulator is 16 VGPRs, spelled out (a compiler-placed MFMA could land in the accumulation file, which belongs to the
// statements above), and the wait states between the last MFMA of such a chain and the first VALU / DS read of its result
__device__ __forceinline__ void mfma_vgpr(float16v& c, half8 w, half8 p) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(p));
}
// first MFMA of such a chain: C = 0 (no VALU-initialised accumulator: a VALU write right in front of an MFMA that reads the register
// as SrcC is a hazard the compiler cannot see here - measured: the last v_mov of a zero-init arrived too late), and two wait states
// for operand registers a VALU instruction may have written just before
__device__ __forceinline__ void mfma_vgpr_zero(float16v& c, half8 w, half8 p) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(w), "v"(p));
}
__device__ __forceinline__ void mfma_vgpr_settle(float16v& c, half8 p0, half8 p1, half8 p2, half8 p3) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(c) : "v"(p0), "v"(p1), "v"(p2), "v"(p3));
}

// one LDS-DMA piece: 64 lanes x 16 B, lane-linear at LDS byte address `lds` (wave-uniform); `voff` = per-lane byte offset into the
// buffer or an out-of-range value (-> zeros).  M0 is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void dma16(unsigned voff, unsigned lds, int4v rsrc) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(lds), "s"(rsrc));
}

__device__ __forceinline__ int4v make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    int4v r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

template <int V>
using ic = std::integral_constant<int, V>;
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_impl(F& f) {
    if constexpr (I < N) {
        f(ic<I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {      // f(ic<0>{}), ..., f(ic<N - 1>{}): indices usable as template arguments
    static_for_impl<0, N>(f);
}

// VAR = SM + 4 * EP.  SM, how the slab reaches LDS: 0 = LDS-DMA issued behind the K-step's fragment reads, 1 = LDS-DMA late in the
//   K-step (behind accumulator block (1, 4), when the fragment reads have returned), 2 = register-staged (buffer_load a group ahead,
//   ds_write_b128 in the next group).  EP, the epilogue: 0 = stores straight from the accumulator layout (a lane owns 64 B of a
//   pixel: four 16-byte pieces), 1 = through a wave-private LDS patch as whole 128-byte lines, 2 = the fused RPN HEAD of
//   conv_wd.h's HEAD == 1 (t = relu(conv + bias) never leaves the registers: the accumulator layout IS the MFMA B-fragment layout
//   of a pre-permuted head weight, pe_conv_wd_pack_head; 16 fp32 outputs per pixel, cross-wave sum through 32 KiB of LDS in the
//   two-wave kernel's order - same bits).
// ABL (measurement builds, results wrong): 1 = no output stores, 2 = no slab traffic, 4 = no weight loads in the loop
// DBG: wave 0 of every workgroup writes s_memtime stamps (kernel start, and per tile: loop start, loop end, epilogue end)
template <int SEGL, int TPX, int DEPTH, int RELU, int VAR = 0, int ABL = 0, int DBG = 0>
__global__ __launch_bounds__(THREADS, 1) void conv3x3_wd9_kernel(pe::ConvWdArgs a, unsigned long long* dbg) {
    using G_ = Geo<SEGL, TPX>;
    constexpr int SEG = G_::SEG, BPX = G_::BPX, SEGP = G_::SEGP, SLAB = G_::SLAB, NPIECE = G_::NPIECE, PPW = G_::PPW, NSEG = G_::NSEG;
    constexpr int SM = VAR & 3, EP = (VAR >> 2) & 3;
    static_assert(EP != 2 || TPX == 8, "fused head: two halves of four pixel blocks");
    static_assert(12 % DEPTH == 0, "weight prefetch depth must divide the 12 K-steps of a group");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned smem_base = (unsigned)(unsigned long long)(lds_byte*)smem;    // LDS byte address of the ring
    // the whole accumulation file belongs to the asm statements below (this makes the kernel descriptor allocate it)
    asm volatile("" ::: "a0", "a255");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = 3 * (a.Cin / 64);
    const int KSEQ = G * 12;
    const int R = a.M >> SEGL;                  // image rows in the batch

    // ---- this workgroup's tiles: XCD x (= blockIdx % 8) owns a contiguous run of tiles, its workgroups stride through it ----
    const int ntile = a.tiles_m * a.tiles_n;
    int t_first, t_step, t_end;
    if (gridDim.x % 8 == 0) {
        const int q = ntile / 8, r = ntile % 8, xcd = blockIdx.x % 8, idx = blockIdx.x / 8;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        t_end = start + (xcd < r ? q + 1 : q);
        t_first = start + idx; t_step = gridDim.x / 8;
    } else {
        t_first = blockIdx.x; t_step = gridDim.x; t_end = ntile;
    }
    if (t_first >= t_end) return;
    const int nmine = (t_end - t_first + t_step - 1) / t_step;

    // ---- DMA pieces of this lane: piece k of a group is slab piece q = 4 k + wn (1 KiB = 8 entries), this lane = entry
    // e = 8 q + (lane >> 3), slot lane & 7 = chunk c ^ ((e >> 1) & 7).  Tile-independent part of the source offset (bytes,
    // relative to the tile's first pixel, channel chunk 0, centre kernel row) or -1 for entries that are never image pixels
    // (halo columns - SEG == W, so a segment is a whole image row -, pad entries, pieces beyond the slab).
    int rel[PPW], sg[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int q = k * 4 + wn;
        const int e = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((e >> 1) & 7);
        const int s = e / SEGP, jj = e - s * SEGP - 1;
        sg[k] = s;
        rel[k] = (q < NPIECE && s < NSEG && (unsigned)jj < (unsigned)SEG) ? ((s * SEG + jj) * a.Cin + c * 8) * 2 : -1;
    }
    const int4v rin = make_rsrc(a.in, (unsigned)a.M * (unsigned)a.Cin * 2u);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.out_stride * 2, 0x00020000);
    // Tile being LOADED (two groups ahead of the one being computed; it changes once per tile, at group G - 2): per piece the
    // centre-row source offset incl. the tile base, and which kernel rows fall outside the image (bit 0: kh = 0, bit 1: kh = 2,
    // bit 2: never an image pixel).  Invalid offsets are 2 GiB (buffers are < 2 GiB: the C-ABI checks), so they stay out of range
    // when the group's channel-chunk shift is added.
    auto tile_m0 = [&](int tile) { return (tile / a.tiles_n) * BPX; };
    int voffc[PPW], badk[PPW];
    auto describe = [&](int row0) {       // row0: first image row (global index n * H + h) of the tile; >= R: no such tile
        const int h0 = row0 % a.H;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            int hs = h0 + sg[k];
            hs = hs >= a.H ? hs - a.H : hs;
            const bool never = (rel[k] < 0) | (row0 + sg[k] >= R);
            voffc[k] = rel[k] + (row0 << SEGL) * a.Cin * 2;
            badk[k] = (never ? 4 : 0) | (hs == 0 ? 1 : 0) | (hs == a.H - 1 ? 2 : 0);
        }
    };
    const __amdgpu_buffer_rsrc_t rin_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, a.M * a.Cin * 2, 0x00020000);
    half8 sreg[SM == 2 ? PPW : 1];               // SM 2: the pieces of the slab after the next, in flight for a whole group
    auto piece_off = [&](int k, int shift, int khbits) { return (badk[k] & khbits) ? 0x80000000u : (unsigned)(voffc[k] + shift); };
    auto dma_piece = [&](int k, int slot, int shift, int khbits) {   // shift = ((kh - 1) * W * Cin + cc * 64) * 2, khbits = 4 | (kh == 0) | 2 (kh == 2)
        dma16(piece_off(k, shift, khbits), smem_base + slot * SLAB + (k * 4 + wn) * 1024, rin);
    };
    auto reg_load = [&](int k, int shift, int khbits) {
        sreg[SM == 2 ? k : 0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rin_c, piece_off(k, shift, khbits), 0, 0));
    };
    auto reg_store = [&](int k, int slot) {
        *reinterpret_cast<half8*>(smem + slot * SLAB + (k * 4 + wn) * 1024 + lane * 16) = sreg[SM == 2 ? k : 0];
    };
    auto group_shift = [&](int g) { const int cc = g / 3, kh = g - cc * 3; return ((kh - 1) * SEG * a.Cin + cc * 64) * 2; };
    auto group_khbits = [&](int g) { const int kh = g % 3; return 4 | (kh == 0 ? 1 : 0) | (kh == 2 ? 2 : 0); };

    // ---- fragment read addresses (ring slot 0, pixel block 0): tap kw, K-step ks; block i adds an immediate ----
    int fa[3][4];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int e = (lane & 31) + kw;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[kw][ks] = e * 128 + (((ks * 2 + (lane >> 5)) ^ ((e >> 1) & 7)) * 16);
    }
    auto blk_off = [](int i) constexpr { return (((i * 32) >> SEGL) * SEGP + ((i * 32) & (SEG - 1))) * 128; };

    // ---- weight stream ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wpk), 0, a.Cout * a.Cin * 18, 0x00020000);
    auto wbase_of = [&](int tile) { return ((tile % a.tiles_n) * KSEQ * WN + wn) * 2048; };
    int w_cur = wbase_of(t_first), w_nxt = nmine > 1 ? wbase_of(t_first + t_step) : w_cur;
    half8 wf[DEPTH][2];
    auto w_load1 = [&](int slot, int blk, int so) {
        wf[slot][blk] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + blk * 1024, so, 0));
    };

    // Accumulators start every tile at ZERO - the tile's first K-step is an MFMA with a constant-zero C operand - and the bias
    // (this wave's 64 channels in accumulator layout, 32 registers) is added in the epilogue.  conv_wd.h sums in the same order.
    float16v bias_v[2];
    auto load_bias = [&](int tile) {
        const float* bp = a.bias + (tile % a.tiles_n) * (WN * 64) + wn * 64 + (lane >> 5) * 32;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = *reinterpret_cast<const float4*>(bp + blk * 16 + r4 * 4);
                bias_v[blk][r4 * 4 + 0] = v.x; bias_v[blk][r4 * 4 + 1] = v.y; bias_v[blk][r4 * 4 + 2] = v.z; bias_v[blk][r4 * 4 + 3] = v.w;
            }
    };

    unsigned long long* dbgp = nullptr;
    if (DBG) {
        dbgp = dbg + (size_t)blockIdx.x * 128;
        if (tid == 0) dbgp[0] = __builtin_readcyclecounter();
    }

    // ---- prologue (once per workgroup): slabs 0 and 1 of the first tile, weight ring primed ----
    describe(tile_m0(t_first) >> SEGL);
    if (!(ABL & 2)) {
        if constexpr (SM == 2) {
#pragma unroll
            for (int k = 0; k < PPW; ++k) reg_load(k, group_shift(0), group_khbits(0));
#pragma unroll
            for (int k = 0; k < PPW; ++k) reg_store(k, 0);
#pragma unroll
            for (int k = 0; k < PPW; ++k) reg_load(k, group_shift(1), group_khbits(1));      // written to slot 1 during group 0
        } else {
#pragma unroll
            for (int k = 0; k < PPW; ++k) dma_piece(k, 0, group_shift(0), group_khbits(0));
#pragma unroll
            for (int k = 0; k < PPW; ++k) dma_piece(k, 1, group_shift(1), group_khbits(1));
        }
    }
    load_bias(t_first);
    half8 hw[EP == 2 ? 4 : 1];      // fused head: K-step j covers accumulator registers (blk j >> 1, half j & 1) of this wave's 64 channels
    if constexpr (EP == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) hw[j] = *reinterpret_cast<const half8*>(a.head_w + ((wn * 4 + j) * 64 + lane) * 8);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        w_load1(d, 0, w_cur + d * (WN * 2048));
        w_load1(d, 1, w_cur + d * (WN * 2048));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float4 hb0 = make_float4(0.f, 0.f, 0.f, 0.f), hb1 = hb0;
    if constexpr (EP == 2) {
        hb0 = *reinterpret_cast<const float4*>(a.head_b + (tid & 1) * 8);
        hb1 = *reinterpret_cast<const float4*>(a.head_b + (tid & 1) * 8 + 4);
    }
    half8 pf[2][TPX];
    int cur = 0;
#pragma unroll
    for (int i = 0; i < TPX; ++i) pf[0][i] = *reinterpret_cast<const half8*>(smem + fa[0][0] + blk_off(i));

    for (int j = 0; j < nmine; ++j) {
        const int tile = t_first + j * t_step;
        if (DBG && tid == 0 && j < 40) dbgp[1 + j * 3] = __builtin_readcyclecounter();
        // the tile after this one (its first two slab groups are loaded under this tile's last two groups)
        const bool has_next = j + 1 < nmine;
        const int nt_row0 = has_next ? tile_m0(tile + t_step) >> SEGL : R;     // R: every row out of range -> zeros, never read
        for (int g = 0; g < G; ++g) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            const int nn = nxt == 2 ? 0 : nxt + 1;
            if (g == G - 2) describe(nt_row0);          // from here on the slab DMAs belong to the next tile
            const int gl = g + 2 >= G ? g + 2 - G : g + 2;
            const int l_shift = group_shift(gl), l_kh = group_khbits(gl);
            const int sb = cur * SLAB, sn = nxt * SLAB;
            // weight records of K-step t + DEPTH: this group's stream, then the next group's (the next tile's first group after the last)
            const int wo_g = w_cur + g * 12 * (WN * 2048);
            const int wo_n = g + 1 < G ? wo_g + 12 * (WN * 2048) : w_nxt;
            // One K-step = 16 MFMAs.  Issue order (the asm statements pin it): accumulator block (0, i) then the next step's pixel
            // fragment i in its shadow; the DMA piece; accumulator blocks (1, i).  The weight records of K-step t + DEPTH go into the
            // ring slot this step is consuming: record 0 is requested once (0, 7) has issued, record 1 after (1, 7).
            auto kstep = [&](auto tc, auto firstc) {
                constexpr int t = decltype(tc)::value;
                constexpr bool FIRST = decltype(firstc)::value;      // the tile's first K-step: C = 0
                constexpr int slot = t % DEPTH;
                constexpr int kw1 = ((t + 1) % 12) / 4, ks1 = ((t + 1) % 12) % 4;
                const unsigned char* src = smem + (t + 1 < 12 ? sb : sn) + fa[kw1][ks1];
                const int wso = t + DEPTH < 12 ? wo_g + (t + DEPTH) * (WN * 2048) : wo_n + (t + DEPTH - 12) * (WN * 2048);
                static_for<TPX>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (FIRST) mfma_zero<i * 16>(wf[slot][0], pf[t & 1][i]);
                    else mfma_acc<i * 16>(wf[slot][0], pf[t & 1][i]);
                    pf[(t + 1) & 1][i] = *reinterpret_cast<const half8*>(src + blk_off(i));
                });
                if constexpr (!(ABL & 2) && t < PPW && SM == 0) dma_piece(t, nn, l_shift, l_kh);
                if constexpr (!(ABL & 2) && t < PPW && SM == 2) {        // slab g + 1 (loaded a group ago) -> its ring slot; slab g + 2 -> registers
                    reg_store(t, nxt);
                    reg_load(t, l_shift, l_kh);
                }
                static_for<TPX>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (FIRST) mfma_zero<(TPX + i) * 16>(wf[slot][1], pf[t & 1][i]);
                    else mfma_acc<(TPX + i) * 16>(wf[slot][1], pf[t & 1][i]);
                    if constexpr (i == 0) { if (!(ABL & 4)) w_load1(slot, 0, wso); }     // record 0 of the slot: free since (0, TPX - 1)
                    if constexpr (i == (TPX > 4 ? 4 : TPX - 1) && !(ABL & 2) && t < PPW && SM == 1) dma_piece(t, nn, l_shift, l_kh);
                });
                if (!(ABL & 4)) w_load1(slot, 1, wso);                                     // record 1: free since (1, TPX - 1)
            };
            if (g == 0) kstep(ic<0>{}, std::true_type{});
            else kstep(ic<0>{}, std::false_type{});
            kstep(ic<1>{}, std::false_type{});
            kstep(ic<2>{}, std::false_type{});
            kstep(ic<3>{}, std::false_type{});
            kstep(ic<4>{}, std::false_type{});
            kstep(ic<5>{}, std::false_type{});
            kstep(ic<6>{}, std::false_type{});
            kstep(ic<7>{}, std::false_type{});
            kstep(ic<8>{}, std::false_type{});
            kstep(ic<9>{}, std::false_type{});
            kstep(ic<10>{}, std::false_type{});
            // Publish slab g + 1 before K-step 11 prefetches the next group's first fragments from it: its DMAs were issued during
            // group g - 1, and everything this wave issued since - at least the 22 weight records of this group's K-steps 0 .. 10 -
            // may stay in flight.  All reads of slab g (the last were step 11's fragments, read in step 10) are behind the barrier's
            // lgkmcnt(0), so group g + 1 may overwrite its slot with slab g + 3.
            if constexpr (SM != 2) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
            __syncthreads();
            kstep(ic<11>{}, std::false_type{});
            cur = nxt;
        }
        if (DBG && tid == 0 && j < 40) dbgp[2 + j * 3] = __builtin_readcyclecounter();
        // ---- tile epilogue: + bias, ReLU, fp16, 64 contiguous bytes per lane and pixel block (buffer stores: rows beyond M are
        // dropped by the bounds check, no branch) ----
        {
            const int m0 = tile_m0(tile), n0 = (tile % a.tiles_n) * (WN * 64);
            const unsigned ob = (unsigned)(((m0 + (lane & 31)) * a.out_stride + n0 + wn * 64 + (lane >> 5) * 32) * 2);
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");      // the last MFMAs' results must have reached the register file
            if constexpr (EP == 2) {
                float* red = reinterpret_cast<float*>(smem + 3 * SLAB);     // [4 waves][128 px][16] fp32 behind the ring
                static_for<2>([&](auto hf_) {
                    constexpr int hf = decltype(hf_)::value;
                    static_for<4>([&](auto ii_) {
                        constexpr int i = hf * 4 + decltype(ii_)::value;
                        float16v hd;
                        // all four B fragments first, then the four MFMAs back to back, then the wait states: an MFMA reads its
                        // operand registers for several cycles after it has issued, and nothing tells the compiler (the MFMAs are
                        // asm statements) not to recycle them at once - the settle statement keeps them alive
                        half8 bf[4];
                        static_for<4>([&](auto j_) {
                            constexpr int j = decltype(j_)::value, blk = j >> 1, hh = j & 1;
                            float x[8];
                            acc_read8<(blk * TPX + i) * 16 + hh * 8>(x);
#pragma unroll
                            for (int e = 0; e < 8; ++e) bf[j][e] = (_Float16)pe::relu_nan(x[e] + bias_v[blk][hh * 8 + e]);
                        });
                        mfma_vgpr_zero(hd, hw[0], bf[0]);
                        mfma_vgpr(hd, hw[1], bf[1]);
                        mfma_vgpr(hd, hw[2], bf[2]);
                        mfma_vgpr(hd, hw[3], bf[3]);
                        mfma_vgpr_settle(hd, bf[0], bf[1], bf[2], bf[3]);
                        // rows (= head outputs) held by this lane: 4 h + (r & 3) + 8 (r >> 2); rows 16 .. 31 are padding
                        float* dst = red + ((wn * 128 + (i & 3) * 32 + (lane & 31)) * 16) + (lane >> 5) * 4;
                        *reinterpret_cast<float4*>(dst) = make_float4(hd[0], hd[1], hd[2], hd[3]);
                        *reinterpret_cast<float4*>(dst + 8) = make_float4(hd[4], hd[5], hd[6], hd[7]);
                    });
                    __syncthreads();
                    {
                        const int px = tid >> 1, q = tid & 1;
                        const int m = m0 + hf * 128 + px;
                        float4 s0 = hb0, s1 = hb1;      // head bias: loaded once per workgroup (a load here would wait out the next tile's slab DMAs)
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const float4 x0 = *reinterpret_cast<const float4*>(red + (w * 128 + px) * 16 + q * 8);
                            const float4 x1 = *reinterpret_cast<const float4*>(red + (w * 128 + px) * 16 + q * 8 + 4);
                            s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
                            s1.x += x1.x; s1.y += x1.y; s1.z += x1.z; s1.w += x1.w;
                        }
                        if (m < a.M && !(ABL & 1)) {
                            float* o = a.head_out + (size_t)m * 16 + q * 8;
                            *reinterpret_cast<float4*>(o) = s0;
                            *reinterpret_cast<float4*>(o + 4) = s1;
                        }
                    }
                    __syncthreads();      // `red` is rewritten by the next half / the next tile
                });
            } else if constexpr (EP == 0) {
                static_for<TPX * 4>([&](auto q_) {
                    constexpr int q = decltype(q_)::value, i = q >> 2, blk = (q >> 1) & 1, hh = q & 1;
                    float x[8];
                    acc_read8<(blk * TPX + i) * 16 + hh * 8>(x);
                    half8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (_Float16)(x[e] + bias_v[blk][hh * 8 + e]);
                    if (RELU) v = __builtin_elementwise_maximum(v, (half8)(_Float16)0.f);
                    if (!(ABL & 1))
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rout, ob + (blk * 16 + hh * 8) * 2, i * 32 * a.out_stride * 2, 0);
                    else if (v[0] == (_Float16)12345.f) dbg[0] = 1;
                });
            } else {
                // a pixel's 64 channels of this wave are one 128-byte line of the output: transpose the pixel block through a
                // wave-private 4 KiB patch (chunk c of row px in slot c ^ (px & 7)) and store 8 whole lines per instruction
                unsigned char* patch = smem + 3 * SLAB + wn * 4096;
                const int px = lane & 31, hq = lane >> 5;
                const int rrow = lane >> 3, rc = (lane & 7) ^ (rrow & 7);      // read side: row (+ 8 r), the channel chunk that slot lane & 7 holds
                const unsigned ob2 = (unsigned)(((m0 + rrow) * a.out_stride + n0 + wn * 64 + rc * 8) * 2);
                static_for<TPX>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    static_for<4>([&](auto q_) {
                        constexpr int q = decltype(q_)::value, blk = q >> 1, hh = q & 1;
                        float x[8];
                        acc_read8<(blk * TPX + i) * 16 + hh * 8>(x);
                        half8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (_Float16)(x[e] + bias_v[blk][hh * 8 + e]);
                        if (RELU) v = __builtin_elementwise_maximum(v, (half8)(_Float16)0.f);
                        *reinterpret_cast<half8*>(patch + px * 128 + (((hq * 4 + blk * 2 + hh) ^ (px & 7)) * 16)) = v;
                    });
                    static_for<4>([&](auto r_) {
                        constexpr int r = decltype(r_)::value;
                        const half8 v = *reinterpret_cast<const half8*>(patch + (r * 8) * 128 + lane * 16);
                        if (!(ABL & 1))
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rout, ob2, (i * 32 + r * 8) * a.out_stride * 2, 0);
                        else if (v[0] == (_Float16)12345.f) dbg[0] = 1;
                    });
                });
            }
            if (a.tiles_n > 1 && has_next) load_bias(tile + t_step);
        }
        if (has_next) {
            w_cur = w_nxt;
            w_nxt = j + 2 < nmine ? wbase_of(tile + 2 * t_step) : w_cur;
        }
        if (DBG && tid == 0 && j < 40) dbgp[3 + j * 3] = __builtin_readcyclecounter();
    }
}

// geometry: W a power of two in [32, 256] that divides the tile, H at least the rows of a tile
inline bool geometry_ok(int H, int W, int TPX) {
    if (W != 32 && W != 64 && W != 128 && W != 256) return false;
    const int bpx = TPX * 32;
    return bpx % W == 0 && H >= bpx / W;
}

template <int SEGL, int TPX, int DEPTH, int RELU, int VAR = 0, int ABL = 0, int DBG = 0>
inline int launch_r(pe::ConvWdArgs a, hipStream_t st, int workgroups, unsigned long long* dbg) {
    using G_ = Geo<SEGL, TPX>;
    a.seg = G_::SEG; a.nseg = G_::NSEG;
    a.tiles_m = pe::ceil_div(a.M, G_::BPX);
    a.tiles_n = a.Cout / (WN * 64);
    constexpr size_t lds = (size_t)3 * G_::SLAB + (((VAR >> 2) & 3) == 2 ? 32768 : 16384);
    static_assert(lds <= 160 * 1024, "ring + epilogue area must fit the CU's LDS");
    PE_ENSURE_LDS((conv3x3_wd9_kernel<SEGL, TPX, DEPTH, RELU, VAR, ABL, DBG>), lds, "conv3x3_wd9");
    const int ntile = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((conv3x3_wd9_kernel<SEGL, TPX, DEPTH, RELU, VAR, ABL, DBG>), dim3(ntile < workgroups ? ntile : workgroups), dim3(THREADS), lds, st, a, dbg);
    return PE_OK;
}

template <int SEGL, int TPX, int DEPTH, int VAR = 0, int ABL = 0, int DBG = 0>
inline int launch_t(pe::ConvWdArgs a, hipStream_t st, int workgroups, unsigned long long* dbg) {
    return a.relu ? launch_r<SEGL, TPX, DEPTH, 1, VAR, ABL, DBG>(a, st, workgroups, dbg) : launch_r<SEGL, TPX, DEPTH, 0, VAR, ABL, DBG>(a, st, workgroups, dbg);
}

template <int TPX, int DEPTH, int VAR = 0, int ABL = 0, int DBG = 0>
inline int launch(pe::ConvWdArgs a, hipStream_t st, int workgroups = 256, unsigned long long* dbg = nullptr) {
    if (!geometry_ok(a.H, a.W, TPX)) return PE_ERR_UNSUPPORTED;
    switch (a.W) {
        case 256: if constexpr (TPX == 8) return launch_t<8, TPX, DEPTH, VAR, ABL, DBG>(a, st, workgroups, dbg); else return PE_ERR_UNSUPPORTED;
        case 128: if constexpr (TPX % 4 == 0) return launch_t<7, TPX, DEPTH, VAR, ABL, DBG>(a, st, workgroups, dbg); else return PE_ERR_UNSUPPORTED;
        case 64: if constexpr (TPX % 2 == 0) return launch_t<6, TPX, DEPTH, VAR, ABL, DBG>(a, st, workgroups, dbg); else return PE_ERR_UNSUPPORTED;
        case 32: if constexpr (TPX <= 6) return launch_t<5, TPX, DEPTH, VAR, ABL, DBG>(a, st, workgroups, dbg); else return PE_ERR_UNSUPPORTED;
    }
    return PE_ERR_UNSUPPORTED;
}

// fused RPN head (EP = 2): ReLU is part of the head, widths 64 / 128 / 256, one 256-channel tile column
template <int TPX, int DEPTH, int VAR>
inline int launch_head(pe::ConvWdArgs a, hipStream_t st, int workgroups = 256) {
    static_assert(((VAR >> 2) & 3) == 2, "launch_head: VAR must select the head epilogue");
    if (!geometry_ok(a.H, a.W, TPX) || a.Cout != WN * 64 || !a.head_w || !a.head_b || !a.head_out) return PE_ERR_UNSUPPORTED;
    switch (a.W) {
        case 256: return launch_r<8, TPX, DEPTH, 1, VAR>(a, st, workgroups, nullptr);
        case 128: return launch_r<7, TPX, DEPTH, 1, VAR>(a, st, workgroups, nullptr);
        case 64: return launch_r<6, TPX, DEPTH, 1, VAR>(a, st, workgroups, nullptr);
    }
    return PE_ERR_UNSUPPORTED;
}

}  // namespace wd9
