// "Weights-direct" convolution kernels for gfx950 (third generation).
//
// Replaces, per layer, the same reference rows as csrc/conv_igemm2.hip (layers/wrappers.py:62-98 Conv2d.forward +
// layers/batch_norm.py:45-65 FrozenBatchNorm2d folded + relu_; backbone/resnet.py:205-221, backbone/fpn.py:127-137,
// proposal_generator/rpn.py:74-85).
//
// What the round-1 measurements said: every LDS-staged variant ends LDS-port bound - per 256x256x64
// step the LDS absorbs 64 KiB of DMA writes at ~64 B/clk plus 192 KiB of fragment reads against 2048 clk of MFMA.
// The weight tile is the larger half of that traffic although weights are STATIC.  So here:
//   * weights never touch LDS.  They are pre-packed ONCE (pe_conv_wd_pack_weights) in MFMA A-fragment order:
//     one 1 KiB record per (32 output channels, 16 K) = exactly what one wavefront `global_load_dwordx4` reads,
//     perfectly coalesced, straight into the VGPRs the MFMA consumes, DEPTH K-steps ahead (L2-resident stream);
//   * only the pixels go through LDS, as a "slab" with EXPLICIT zero halo entries: entry e of an image-row segment
//     holds input pixel (col - 1 + e), so the three kw taps are the SAME slab read at entry offsets 0 / 1 / 2 with
//     no per-lane edge masks, and one slab serves a whole (kh, 64-channel chunk) group = 12 K-steps of 16;
//   * slab rows are padded to 144 B (36 dwords): every ds_read_b128 of a fragment is bank-conflict-free and
//     (kw, ks) become IMMEDIATE offsets of the read - no address VALU in the loop;
//   * slab filling is register-staged (global_load_dwordx4 -> VGPR -> ds_write_b128), a group ahead, into a
//     three-deep LDS ring: one barrier per 12 K-steps, and it is never on the read path;
//   * the MFMA is issued as D[cout][pixel] = W[cout][k] * X[k][pixel] with the 32 output channels of a record
//     PERMUTED so that a lane's 32 accumulator registers are 32 CONSECUTIVE output channels of one pixel: the
//     epilogue is bias (folded into the accumulator init) + ReLU + four 16-byte stores per pixel block, straight
//     from registers - no LDS transposition, no epilogue barriers.
// Per wave: 128 pixels x 64 output channels (TPX = 4 pixel blocks x 2 channel blocks, 128 accumulator VGPRs), per
// K-step of 16: 2 weight loads + 4 ds_read_b128 + 8 MFMA 32x32x16.  LDS traffic per MFMA is 1/4 of the 256x256
// LDS-staged tile's; the texture path carries 32 B/clk/CU of weight stream instead.
#pragma once
#include <hip/hip_fp16.h>

#include <type_traits>

#include "common.h"

namespace pe {
struct ConvWdArgs {
    const _Float16* in;    // NHWC fp16
    const _Float16* wpk;   // packed weights (pe_conv_wd_pack_weights)
    const float* bias;     // [Cout] or null
    const _Float16* res;   // residual (1x1 kernels), NHWC fp16, or null
    _Float16* out;
    int N, H, W, Cin, Cout;
    int M;           // N * H * W output pixels (3x3 stride 1 / 1x1 stride 1)
    int relu;
    int out_stride;  // halfs between output pixels
    int seg, nseg;   // image-row segment length of a block tile and segments per tile
    int tiles_m, tiles_n;
    // fused RPN head (HEAD builds): out[m][0..15] = head_b + head_w[16 x Cout] * relu(conv3x3(in))[m]
    const _Float16* head_w;   // packed by pack_head_kernel
    const float* head_b;      // [16]
    float* head_out;          // [M, 16] fp32
    // fused bottleneck tail (TAIL builds): tail_out = relu(tail_b + tail_w[tail_cout x 256] * relu(conv3x3(in)) + tail_res)
    const _Float16* tail_w;   // packed by pack_tail_kernel
    const float* tail_b;      // [tail_cout]
    const _Float16* tail_res; // [M, tail_cout] fp16 or null
    _Float16* tail_out;       // [M, tail_cout] fp16
    int tail_cout;            // multiple of 256
    // 1x1 lab kernel only (scripts/lab/conv_wd_1x1.h)
    int stride, Ho, Wo;          // output grid (stride 1 | 2)
    int res_mode, resH, resW;    // 0 none, 1 residual has the output's shape, 2 residual [N,resH,resW,Cout] read at (oh/2, ow/2)
};
}  // namespace pe

namespace wd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int SLAB_ROW_B = 144;  // 128 B of channels + 16 B pad: rows r .. r+15 hit 16 distinct 4-bank groups

__device__ __attribute__((aligned(16))) static unsigned int g_zero16[4] = {0, 0, 0, 0};

// Host-side mirror of the packed layout (used by the packing kernel and by tests):
// record index = ((tile_n * KSEQ + kseq) * WN + wn) * 2 + blk, 512 halfs each; lane l holds
// W[cout(l & 31)][k0 + 8 * (l >> 5) .. + 8] with cout = tile_n * WN * 64 + wn * 64 + perm(blk, l & 31).
__host__ __device__ inline int cout_perm(int blk, int rho) {
    return ((rho >> 2) & 1) * 32 + blk * 16 + (rho >> 3) * 4 + (rho & 3);
}
// K order of the 3x3 kernel: group g = (channel chunk cc outer, kernel row kh inner), then kw, then ks (16 wide)
__host__ __device__ inline int k3x3_of_kseq(int kseq, int Cin, int e) {
    const int g = kseq / 12, t = kseq - g * 12;
    const int cc = g / 3, kh = g - cc * 3, kw = t / 4, ks = t - kw * 4;
    return (kh * 3 + kw) * Cin + cc * 64 + ks * 16 + e;
}

// ------------------------------------------------------------------------------------------------------
// 3x3, stride 1, pad 1.  Block = WM x WN waves; wave (wm, wn) owns pixels [wm*TPX*32, +TPX*32) x channels [wn*64, +64).
// ------------------------------------------------------------------------------------------------------
// ABL (measurement builds only, results wrong; instantiated by scripts/ probes and the -DPE_LAB library, never by the product): 1 = no weight
// loads in the loop, 2 = no LDS fragment reads in the loop, 4 = no slab traffic and no barrier in the loop, 8 (HEAD == 2) = every conv3
// chunk's K-loop runs twice and 64 KiB more lines are stored per tile: what a fused NEXT conv1 (1024 -> 256) would add in MFMAs, weight
// records, fragment reads and stores if its 128 accumulators and its 64 KiB exchange tile were free
// HEAD: the StandardRPNHead form (proposal_generator/rpn.py:74-85): the ReLU'd 3x3 output t never goes to memory; each wave
// multiplies its 64 channels of t - the accumulators, converted to fp16, ARE MFMA B fragments when the head weight's K order
// is packed to match - with the 15 x 256 objectness / delta weights, the four partial [128 px x 16] sums are added through LDS
// and only the fp32 head rows (64 B per pixel instead of 512 B of t, and no second launch that re-reads t) are stored.
// HEAD == 2 ("tail"): the second half of a BottleneckBlock (backbone/resnet.py:205-221) in the same launch: t = relu(conv2(x))
// (128 px x 256 channels per workgroup) is written as fp16 into the idle slab ring, then every wave computes 256 of the
// tail_cout outputs of conv3 in chunks of 64 - t fragments from LDS, conv3 weights L2 -> VGPR in fragment order, no barrier in
// the whole phase -, adds the shortcut (one channel quarter per 4 K-steps, requested 3 K-steps ahead, 16 registers live) and ReLU, and stores.
// Neither t nor a second launch's re-read of it touches HBM, and the latency-bound 1x1 kernel is gone from the block.
typedef float float8v __attribute__((ext_vector_type(8)));
// 2 x 16 consecutive floats at two wave-uniform addresses through the scalar cache, one wait
__device__ __forceinline__ void wd_sload4(float8v& a0, float8v& a1, float8v& b0, float8v& b1, const float* pa, const float* pb) {
    auto uni = [](const float* p) {
        const unsigned long long v = reinterpret_cast<unsigned long long>(p);
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    asm volatile("s_load_dwordx8 %0, %4, 0x0\n\ts_load_dwordx8 %1, %4, 0x20\n\ts_load_dwordx8 %2, %5, 0x0\n\ts_load_dwordx8 %3, %5, 0x20\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(a0), "=&s"(a1), "=&s"(b0), "=&s"(b1)
                 : "s"(uni(pa)), "s"(uni(pb)));
}
// Diagnosis switches of the fused tail (variant libraries built by scripts/lab/build_variant.py only - the product object is
// compiled with both at 0 and contains none of these branches; results are WRONG with any of them set).
// PE_EXP_TAIL_NORES: 1 = shortcut loads all out of range (no traffic); 2 = every shortcut load reads a 16 KiB window (L1 hits); 3 = loads
// issued, never added (no explicit wait); 4 = addresses folded into a 2 MiB window (L1 misses, L2 hits).  PE_EXP_TAIL_NOSTORE: stores dropped.
#ifndef PE_EXP_TAIL_NORES
#define PE_EXP_TAIL_NORES 0
#endif
#ifndef PE_EXP_TAIL_NOSTORE
#define PE_EXP_TAIL_NOSTORE 0
#endif
template <int WM, int WN, int TPX, int DEPTH, int ABL = 0, int HEAD = 0>
__global__ __launch_bounds__(64 * WM * WN, (TPX <= 4 ? 2 : 1)) void conv3x3_wd_kernel(pe::ConvWdArgs a) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int BPX = WM * TPX * 32;                  // pixels per block tile
    constexpr int EMAX = BPX + BPX / 16;                // slab entries when every segment is 32 pixels
    constexpr int NP = (EMAX * 8 + THREADS - 1) / THREADS;  // 16-byte slab pieces per thread
    static_assert(12 % DEPTH == 0, "weight prefetch depth must divide the 12 K-steps of a group");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {   // bijective XCD remap: each XCD (private L2) gets a contiguous run of tiles
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BPX, n0 = tile_n * (WN * 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int segp = a.seg + 2;
    const int E = a.nseg * segp;
    const int slab_bytes = E * SLAB_ROW_B;

    // ---- slab pieces owned by this thread: BYTE offset of the centre-row source (or -1) and its image row ----
    // (pieces beyond the slab write to a private 16-byte dummy slot behind the ring: no divergent branches;
    //  invalid sources - halo columns, rows above / below the image, pixels beyond M - use an out-of-range buffer
    //  offset: the hardware bounds check of buffer_load returns zeros)
    int p_off[NP], p_h[NP], p_lds[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int q = tid + n * THREADS;
        const int e = q >> 3, c = q & 7;
        p_off[n] = -1; p_h[n] = 0; p_lds[n] = 3 * slab_bytes + tid * 16;
        if (e < E) {
            const int s = e / segp, jj = e - s * segp - 1;
            const int P0 = m0 + s * a.seg;
            const int row = P0 / a.W, col = P0 - row * a.W + jj;
            p_lds[n] = e * SLAB_ROW_B + c * 16;
            p_h[n] = row % a.H;
            if (P0 < a.M && (unsigned)col < (unsigned)a.W) p_off[n] = ((P0 + jj) * a.Cin + c * 8) * 2;
        }
    }
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, a.M * a.Cin * 2, 0x00020000);
    half8 sreg[NP];
    auto slab_load1 = [&](int n, int g) {   // piece n of group g = (cc, kh): global -> registers
        const int cc = g / 3, kh = g - cc * 3;
        const int shift = ((kh - 1) * a.W * a.Cin + cc * 64) * 2;
        const bool ok = p_off[n] >= 0 && (unsigned)(p_h[n] + kh - 1) < (unsigned)a.H;
        const unsigned vo = ok ? (unsigned)(p_off[n] + shift) : 0xFFFFFFF0u;
        sreg[n] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rin, vo, 0, 0));
    };
    auto slab_store1 = [&](int n, int buf) {  // registers -> LDS ring slot (the dummy slots are never read)
        *reinterpret_cast<half8*>(smem + (p_lds[n] < 3 * slab_bytes ? buf * slab_bytes : 0) + p_lds[n]) = sreg[n];
    };

    // ---- fragment read bases: pixel block i of this wave, entry of tap kw = 0 ----
    int fb[TPX];
#pragma unroll
    for (int i = 0; i < TPX; ++i) {
        const int p = (wm * TPX + i) * 32;
        const int s = p / a.seg, j0 = p - s * a.seg;
        fb[i] = (s * segp + j0 + (lane & 31)) * SLAB_ROW_B + (lane >> 5) * 16;
    }

    // ---- weight stream: record pair (blk 0, 1) of K-step kseq for this wave: scalar offset + lane * 16 ----
    const int G = 3 * (a.Cin / 64);
    const int KSEQ = G * 12;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wpk), 0, a.Cout * a.Cin * 18, 0x00020000);
    const int w_base = (tile_n * KSEQ * WN + wn) * 2048;
    half8 wf[DEPTH][2];
    auto w_load = [&](int slot, int kseq) {
        const int ks = kseq < KSEQ ? kseq : KSEQ - 1;   // tail prefetches re-read the last record (unused)
        const int so = w_base + ks * (WN * 2048);
        wf[slot][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, so, 0));
        wf[slot][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + 1024, so, 0));
    };

    // ---- accumulators start at zero; the bias is added in the epilogue (conv_wd9.h, which takes the large launches of the same
    // layers, sums in the same order: results must not depend on which of the two kernels a batch size selects) ----
    float16v acc[2][TPX];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < TPX; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][i][r] = 0.f;

    // ---- prologue: slabs 0 and 1 into the ring, weight ring primed ----
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_load1(n, 0);
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_store1(n, 0);
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_load1(n, G > 1 ? 1 : 0);
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_store1(n, 1);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) w_load(d, d);
    __syncthreads();

    half8 pf[2][TPX];
    int cur = 0;  // ring slot of group g
#pragma unroll
    for (int i = 0; i < TPX; ++i) pf[0][i] = *reinterpret_cast<const half8*>(smem + fb[i]);

    static_assert(NP <= 12, "slab pieces are spread over the 12 K-steps of a group: piece n is loaded in step n and stored in step 12 - NP + n");
    for (int g = 0; g < G; ++g) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        const int nn = nxt == 2 ? 0 : nxt + 1;
        // slab g+2: piece n is loaded global -> registers in K-step n and stored registers -> ring slot nn in K-step
        // 12 - NP + n of the same group (one basic block: the compiler counts vmcnt exactly).  Slot nn was last read in
        // group g-1 and every wave has passed that barrier.
        const int gl = g + 2 < G ? g + 2 : G - 1;
        const unsigned char* sb = smem + cur * slab_bytes;
        const unsigned char* sn = smem + nxt * slab_bytes;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            if (!(ABL & 4) && t < NP) slab_load1(t, gl);
            // next step's pixel fragments (step 0 of the next group comes from the next ring slot, published by the
            // barrier at the end of group g-1)
            if (ABL & 2) {
            } else if (t + 1 < 12) {
                const int kw1 = (t + 1) / 4, ks1 = (t + 1) - kw1 * 4;
#pragma unroll
                for (int i = 0; i < TPX; ++i)
                    pf[(t + 1) & 1][i] = *reinterpret_cast<const half8*>(sb + fb[i] + kw1 * SLAB_ROW_B + ks1 * 32);
            } else {
#pragma unroll
                for (int i = 0; i < TPX; ++i) pf[(t + 1) & 1][i] = *reinterpret_cast<const half8*>(sn + fb[i]);
            }
            const int slot = t % DEPTH;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int i = 0; i < TPX; ++i)
                    acc[blk][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][blk], pf[t & 1][i], acc[blk][i], 0, 0, 0);
            if (!(ABL & 1)) w_load(slot, g * 12 + t + DEPTH);
            if (!(ABL & 4) && t >= 12 - NP) slab_store1(t - (12 - NP), nn);
            // issue order inside the step: the MFMAs of blk 0 carry the next step's LDS reads in their shadows, the
            // MFMAs of blk 1 carry the weight loads (whose ring slot blk 0 / blk 1 have just released), the slab piece
            // load and the slab piece store
#pragma unroll
            for (int i = 0; i < TPX; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // (slab piece load, steps 0 .. NP-1)
            __builtin_amdgcn_sched_group_barrier(0x008, TPX - 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // (slab piece store, last NP steps)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(ABL & 4)) __syncthreads();  // publishes slab g+2, retires the reads of slab g
        cur = nxt;
    }

    // ---- epilogue: + bias (lane holds channels n0 + wn*64 + (lane>>5)*32 + blk*16 + r), ReLU, fp16, 64 contiguous bytes per lane
    // and pixel block ----
    {
        const float* bp = a.bias + n0 + wn * 64 + (lane >> 5) * 32;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float16v b;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = *reinterpret_cast<const float4*>(bp + blk * 16 + r4 * 4);
                b[r4 * 4 + 0] = v.x; b[r4 * 4 + 1] = v.y; b[r4 * 4 + 2] = v.z; b[r4 * 4 + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < TPX; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[blk][i][r] += b[r];
        }
    }
    if (a.relu || HEAD) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < TPX; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[blk][i][r] = pe::relu_nan(acc[blk][i][r]);
    }
    if constexpr (HEAD == 1) {
        static_assert(WM == 1 && WN == 4 && TPX == 4, "fused head: one 128-pixel x 256-channel tile per workgroup");
        // head weight fragments of this wave: K-step j covers channels wn*64 + (lane>>5)*32 + j*8 .. +8
        half8 hw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) hw[j] = *reinterpret_cast<const half8*>(a.head_w + ((wn * 4 + j) * 64 + lane) * 8);
        float* red = reinterpret_cast<float*>(smem);     // [4 waves][128 px][16] fp32 = 32 KiB over the (now idle) slab ring
#pragma unroll
        for (int i = 0; i < TPX; ++i) {
            float16v hd;
#pragma unroll
            for (int r = 0; r < 16; ++r) hd[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half8 bf;
#pragma unroll
                for (int e = 0; e < 8; ++e) bf[e] = (_Float16)acc[j >> 1][i][(j & 1) * 8 + e];
                hd = __builtin_amdgcn_mfma_f32_32x32x16_f16(hw[j], bf, hd, 0, 0, 0);
            }
            // rows (= head outputs) held by this lane: 4h + (r & 3) + 8 (r >> 2): r 0..3 -> o = 4h + r, r 4..7 -> o = 8 + 4h + (r - 4)
            float* dst = red + ((wn * 128 + i * 32 + (lane & 31)) * 16) + (lane >> 5) * 4;
            *reinterpret_cast<float4*>(dst) = make_float4(hd[0], hd[1], hd[2], hd[3]);
            *reinterpret_cast<float4*>(dst + 8) = make_float4(hd[4], hd[5], hd[6], hd[7]);
        }
        __syncthreads();
        {
            const int px = tid >> 1, q = tid & 1;
            const int m = m0 + px;
            float4 s0 = *reinterpret_cast<const float4*>(a.head_b + q * 8), s1 = *reinterpret_cast<const float4*>(a.head_b + q * 8 + 4);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float4 x0 = *reinterpret_cast<const float4*>(red + (w * 128 + px) * 16 + q * 8);
                const float4 x1 = *reinterpret_cast<const float4*>(red + (w * 128 + px) * 16 + q * 8 + 4);
                s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
                s1.x += x1.x; s1.y += x1.y; s1.z += x1.z; s1.w += x1.w;
            }
            if (m < a.M) {
                float* o = a.head_out + (size_t)m * 16 + q * 8;
                *reinterpret_cast<float4*>(o) = s0;
                *reinterpret_cast<float4*>(o + 4) = s1;
            }
        }
        return;
    }
    if constexpr (HEAD == 2) {
        static_assert(WM == 1 && WN == 4 && TPX == 4 && DEPTH == 4, "fused tail: one 128-pixel x 256-channel tile per workgroup");
        constexpr int TROW = 528;                       // 256 halfs + 16 B pad: rows r .. r+15 hit 16 distinct bank groups
        // ---- t -> LDS (the slab ring is idle: every wave passed the loop's last barrier) ----
#pragma unroll
        for (int i = 0; i < TPX; ++i) {
            unsigned char* dst = smem + (i * 32 + (lane & 31)) * TROW + (wn * 64 + (lane >> 5) * 32) * 2;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    half8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (_Float16)acc[blk][i][hh * 8 + e];
                    *reinterpret_cast<half8*>(dst + (blk * 16 + hh * 8) * 2) = v;
                }
        }
        __syncthreads();
        const int NCH = a.tail_cout / 256;               // 64-output chunks per wave
        const int NS = NCH * 16;                         // K-steps of the tail phase (K = 256)
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.tail_w), 0, a.tail_cout * 256 * 2, 0x00020000);
        const int t_base = wn * NCH * 16 * 2048;
        auto t_load = [&](int slot, int step) {
            const int so = t_base + (step < NS ? step : NS - 1) * 2048;
            wf[slot][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rt, lane * 16, so, 0));
            wf[slot][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rt, lane * 16 + 1024, so, 0));
        };
        t_load(0, 0);     // four K-steps of weight prefetch, like the 3x3 phase (round 6; two until then)
        t_load(1, 1);
        t_load(2, 2);
        t_load(3, 3);
        int tb[TPX];
#pragma unroll
        for (int i = 0; i < TPX; ++i) tb[i] = (i * 32 + (lane & 31)) * TROW + (lane >> 5) * 16;
#pragma unroll
        for (int i = 0; i < TPX; ++i) pf[0][i] = *reinterpret_cast<const half8*>(smem + tb[i]);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.tail_res ? a.tail_res : a.in), 0,
                                                                            a.tail_res ? a.M * a.tail_cout * 2 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rto = __builtin_amdgcn_make_buffer_rsrc(a.tail_out, 0, a.M * a.tail_cout * 2, 0x00020000);
        unsigned rbase[TPX];                              // byte offset of this lane's 64 residual bytes for chunk 0, or out of range
#pragma unroll
        for (int i = 0; i < TPX; ++i) {
            const int m = m0 + i * 32 + (lane & 31);
            rbase[i] = (a.tail_res && m < a.M && PE_EXP_TAIL_NORES != 1) ? (unsigned)(((size_t)m * a.tail_cout + wn * (NCH * 64) + (lane >> 5) * 32) * 2) : 0xFFFFFFF0u;
        }
        // shortcut quarters (8 of the lane's 32 outputs, all four pixel blocks) run as a rolling two-deep pipeline across the
        // chunks: global quarter g = 4 c + q is requested at K-step 4 g - 4 and added at 4 g + 3 (7 K-steps of HBM latency
        // slack, two quarters = 32 registers in flight).  Splitting by channel - not by pixel block - keeps the fp32 summation
        // order of an output independent of where its pixel sits in the tile (results do not depend on the batch composition).
        half8 rv[2][TPX];
        auto r_load = [&](int set, int g) {      // g < 4 * NCH, else a harmless out-of-range request
#pragma unroll
            for (int i = 0; i < TPX; ++i)
                rv[set][i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(
                    rr, (rbase[i] == 0xFFFFFFF0u || g >= 4 * NCH) ? 0xFFFFFFF0u : (PE_EXP_TAIL_NORES == 2 ? (unsigned)(lane * 16 + i * 1024 + wn * 4096) : PE_EXP_TAIL_NORES == 4 ? ((rbase[i] + (unsigned)((g >> 2) * 128 + (g & 3) * 16)) & 0x1FFFFFu) : rbase[i] + (unsigned)((g >> 2) * 128 + (g & 3) * 16)), 0, 0));
        };
        r_load(0, 0);
        for (int c = 0; c < NCH; ++c) {
            {
                // Round 5: the chunk's bias through the SCALAR cache (the address of the wave's 64 outputs is uniform; a lane picks its
                // 32 by lane >> 5).  As vector loads these were the YOUNGEST vector-memory operations at the chunk's first MFMA, so waiting
                // for them was `vmcnt(0)`: every chunk opened by draining the previous chunk's 16 line stores (write acknowledgements under
                // full HBM load) and the prefetched shortcut quarter.  Without them the first wait is a counted one and the stores stay in flight.
                // Same bits; -1..-5 % per launch, +0.2 % in the pipeline (profiles/r05_tail_bias_ab.txt; the A/B hook and the old form are gone).
                const float* bw = a.tail_b + wn * (NCH * 64) + c * 64;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    float8v s0, s1, s2, s3;      // outputs [blk * 16, +16) of lane half 0 (s0, s1) and of lane half 1 (s2, s3)
                    wd_sload4(s0, s1, s2, s3, bw + blk * 16, bw + 32 + blk * 16);
                    float16v b;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        b[e] = (lane >> 5) ? s2[e] : s0[e];
                        b[e + 8] = (lane >> 5) ? s3[e] : s1[e];
                    }
#pragma unroll
                    for (int i = 0; i < TPX; ++i) acc[blk][i] = b;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the chunk's 16 K-steps.  SC (compile time): with the shortcut pipeline - false only in the LAB pricing build (ABL & 8), which
            // runs the K-loop a second time on the same fragments to price the MFMA / weight-stream / fragment-read work of a fused NEXT conv1
            auto chunk_ksteps = [&](auto sc_tag, int wbase) {
                constexpr bool SC = decltype(sc_tag)::value;
                // Round 6 (profiles/r06_tail_tcp_diagnosis.txt): the shortcut's HBM misses sit in the same in-order vector-memory path as the
                // weight records, so whatever is issued behind a shortcut request waits for it.  The K-step therefore (1) keeps FOUR K-steps of
                // weights in flight (all of wf, as in the 3x3 phase), (2) issues the shortcut request AFTER its own weight loads - the first
                // record that can be held up by the request is needed five K-steps later -, and (3) pays for the two extra weight slots with a
                // SINGLE set of pixel fragments: pixel block i's two MFMAs are issued together and its next fragment is read right behind them
                // (six MFMAs = 192 cycles before it is needed).  Same MFMAs on the same accumulators in the same K order, same shortcut adds at
                // the same K-steps: same bits (tests/test_ops_gpu.py tail tests); -1.5 % per launch, +0.4 % in the pipeline.
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
                    for (int i = 0; i < TPX; ++i) {
                        acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks & 3][0], pf[0][i], acc[0][i], 0, 0, 0);
                        acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks & 3][1], pf[0][i], acc[1][i], 0, 0, 0);
                        pf[0][i] = *reinterpret_cast<const half8*>(smem + tb[i] + ((ks + 1) & 15) * 32);
                    }
                    t_load(ks & 3, wbase + ks + 4);
                    if (SC && (ks & 3) == 0) r_load(((ks >> 2) + 1) & 1, c * 4 + (ks >> 2) + 1);   // the NEXT quarter (may belong to the next chunk)
#pragma unroll
                    for (int i = 0; i < TPX; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                    if (SC && (ks & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (SC && (ks & 3) == 3 && (PE_EXP_TAIL_NORES != 3 || a.N < 0)) {
                        const int q = ks >> 2;
#pragma unroll
                        for (int i = 0; i < TPX; ++i)
#pragma unroll
                            for (int e = 0; e < 8; ++e) acc[q >> 1][i][(q & 1) * 8 + e] += (float)rv[q & 1][i][e];
                    }
                }
            };
            if constexpr ((ABL & 8) != 0) chunk_ksteps(std::false_type{}, c * 16);      // LAB: a second pass over the chunk (results wrong)
            chunk_ksteps(std::true_type{}, c * 16);
            // chunk epilogue: ReLU, fp16.  A lane owns 64 B of a pixel's 128-byte line (this wave's 64 outputs of the chunk); stored
            // straight from that layout every instruction scatters 16-byte pieces over 64 lines (r02 / r03).  Now half a pixel block
            // (16 lines) at a time goes through a wave-private 2 KiB LDS patch (piece p of row px in slot p ^ (px & 7)) and leaves
            // as WHOLE lines, 8 per instruction, with the non-temporal hint: the 210 MB of output streaming through the L2 were what
            // evicted the shortcut lines between their four quarter reads (the 1.36 x traffic of r03), and whole lines can carry
            // `nt` without the write amplification partial ones get (`nt` on the 16-byte pieces: 2.7 x the write traffic).
            // profiles/r04_tail_store_ab.txt: 0.2374 -> 0.2034 ms per launch, 904 -> 931 pairs/s.
            unsigned char* patch = smem + 128 * TROW + wn * 2048;
            const int px = lane & 31, hq = lane >> 5;
            const int rrow = lane >> 3, rc = (lane & 7) ^ (rrow & 7);
            const unsigned lb = (unsigned)((wn * (NCH * 64) + c * 64 + rc * 8) * 2);
#pragma unroll
            for (int i = 0; i < TPX; ++i)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    if ((px >> 4) == h2) {
#pragma unroll
                        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
                                half8 v;
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (_Float16)pe::relu_nan(acc[blk][i][hh * 8 + e]);
                                *reinterpret_cast<half8*>(patch + (px & 15) * 128 + (((hq * 4 + blk * 2 + hh) ^ (px & 7)) * 16)) = v;
                            }
                    }
                    // Half the lanes wrote, all lanes read what OTHER lanes wrote: without a convergent operation in between the
                    // compiler may (and did) give the non-writing lanes their own copy of the reads on the other side of the
                    // divergent branch, which the hardware can run BEFORE the writers' side.  LDS itself is in order per wave.
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const half8 v = *reinterpret_cast<const half8*>(patch + r * 1024 + lane * 16);
                        const int m = m0 + i * 32 + h2 * 16 + r * 8 + rrow;      // rows >= M: beyond the buffer's records, dropped
                        const unsigned off = PE_EXP_TAIL_NOSTORE ? 0xFFFFFFF0u : (unsigned)m * (unsigned)(a.tail_cout * 2) + lb;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rto, off, 0, 2);      // aux 2 = nt
                        if constexpr ((ABL & 8) != 0) {      // LAB: the NEXT conv1's output lines (128 px x 512 B per tile), here a copy of chunk wn
                            if (c == wn && a.out != nullptr && m < a.M)
                                *reinterpret_cast<uint4v*>(reinterpret_cast<unsigned char*>(a.out) + (size_t)m * 512 + wn * 128 + rc * 16) = __builtin_bit_cast(uint4v, v);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();      // the next half's writers overwrite rows the other lanes have just read
                }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TPX; ++i) {
        const int m = m0 + (wm * TPX + i) * 32 + (lane & 31);
        if (m >= a.M) continue;
        _Float16* o = a.out + (size_t)m * a.out_stride + n0 + wn * 64 + (lane >> 5) * 32;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                half8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)acc[blk][i][hh * 8 + e];
                *reinterpret_cast<half8*>(o + blk * 16 + hh * 8) = v;
            }
    }
}

// conv3 weights [tail_cout][256] fp16 -> fragment records in the tail phase's stream order:
// record (wn, c, ks, blk), lane l: output = wn*(NCH*64) + c*64 + cout_perm(blk, l & 31), 8 halfs = k ks*16 + (l>>5)*8 + e
static __global__ void pack_tail_kernel(const _Float16* w, _Float16* out, int tail_cout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per lane slot
    const int NCH = tail_cout / 256;
    if (idx >= 4 * NCH * 16 * 2 * 64) return;
    const int lane = idx & 63;
    int rec = idx >> 6;
    const int blk = rec & 1; rec >>= 1;
    const int ks = rec & 15; rec >>= 4;
    const int c = rec % NCH, wn = rec / NCH;
    const int o = wn * (NCH * 64) + c * 64 + cout_perm(blk, lane & 31);
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[(size_t)o * 256 + ks * 16 + (lane >> 5) * 8 + e];
    *reinterpret_cast<half8*>(out + (size_t)idx * 8) = v;
}

// head weights [rows <= 16][256] fp16 -> A fragments in the K order the accumulator layout dictates:
// record (wn, j), lane l: row = l & 31 (rows >= `rows` are zero), 8 halfs = channels wn*64 + (l>>5)*32 + j*8 + e
static __global__ void pack_head_kernel(const _Float16* w, _Float16* out, int rows, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // one thread per lane slot: 4 wn x 4 j x 64 lanes
    if (idx >= 4 * 4 * 64) return;
    const int lane = idx & 63, j = (idx >> 6) & 3, wn = idx >> 8;
    const int row = lane & 31;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = wn * 64 + (lane >> 5) * 32 + j * 8 + e;
        v[e] = row < rows ? w[(size_t)row * C + ch] : (_Float16)0.f;
    }
    *reinterpret_cast<half8*>(out + (size_t)idx * 8) = v;
}

// packing: [Cout][3][3][Cin] (or [Cout][K] for 1x1 with order = 0) -> fragment records
static __global__ void pack_weights_kernel(const _Float16* w, _Float16* out, int Cout, int K, int Cin, int WN, int is3x3) {
    // one thread per 16-byte lane slot
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KSEQ = K / 16;
    const long long total = (long long)(Cout / 32) * KSEQ * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    long long rec = idx >> 6;
    const int blk = (int)(rec & 1); rec >>= 1;
    const int wn = (int)(rec % WN); rec /= WN;
    const int kseq = (int)(rec % KSEQ);
    const int tile_n = (int)(rec / KSEQ);
    const int cout = tile_n * WN * 64 + wn * 64 + cout_perm(blk, lane & 31);
    const int e0 = (lane >> 5) * 8;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = is3x3 ? k3x3_of_kseq(kseq, Cin, e0 + e) : kseq * 16 + e0 + e;
        v[e] = w[(size_t)cout * K + k];
    }
    *reinterpret_cast<half8*>(out + idx * 8) = v;
}

// geometry the 3x3 kernel supports for a block tile of BPX pixels
inline bool wd3x3_geometry(int W, int BPX, int* seg, int* nseg) {
    if (W % 32) return false;
    const int s = W < BPX ? W : BPX;
    if (W % s || BPX % s) return false;
    *seg = s; *nseg = BPX / s;
    return true;
}

template <int WM, int WN, int TPX, int DEPTH, int ABL = 0, int HEAD = 0>
int launch_conv3x3_wd(pe::ConvWdArgs a, hipStream_t st) {
    constexpr int BPX = WM * TPX * 32;
    if (!wd3x3_geometry(a.W, BPX, &a.seg, &a.nseg)) return PE_ERR_UNSUPPORTED;
    a.tiles_m = pe::ceil_div(a.M, BPX);
    a.tiles_n = a.Cout / (WN * 64);
    size_t lds = (size_t)3 * a.nseg * (a.seg + 2) * SLAB_ROW_B + (size_t)64 * WM * WN * 16;
    if (HEAD == 1 && lds < 32768) lds = 32768;        // the partial head sums reuse the slab ring
    if (HEAD == 2 && lds < 128 * 528 + 4 * 2048) lds = 128 * 528 + 4 * 2048;  // so does the fp16 copy of t; + the line-store patches
    PE_ENSURE_LDS((conv3x3_wd_kernel<WM, WN, TPX, DEPTH, ABL, HEAD>), lds, "conv3x3_wd");
    hipLaunchKernelGGL((conv3x3_wd_kernel<WM, WN, TPX, DEPTH, ABL, HEAD>), dim3(a.tiles_m * a.tiles_n), dim3(64 * WM * WN), lds, st, a);
    return PE_OK;
}

}  // namespace wd
