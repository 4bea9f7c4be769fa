// LAB NOTEBOOK (not built into libproben_hip.so): the weights-direct skeleton of csrc/conv_wd.h applied to the 1x1
// convolutions / FC GEMMs.  Correct on every shape tried (scripts/conv_wd_probe -1), but measured on MI355X (r02) it
// only wins where K is long and there is no residual - res5 conv1 2048->512: 0.059 vs 0.076 ms, fc2: 0.076 vs 0.088 ms,
// fc1 K=12544: 0.792 vs 0.807 ms - and LOSES on the HBM-bound layers that dominate the class: res4 conv3 + residual
// 0.152 vs 0.127 ms, res2 conv3 0.470 vs 0.347 ms, res3 conv3 0.276 vs 0.206 ms, FPN lateral 0.217 vs 0.187 ms.
// Why: the register epilogue reads / writes 16-byte pieces 64 B apart per lane (32 couts of ONE pixel per lane), while
// the LDS-transposed epilogue of conv_igemm2.hip moves full 256-byte rows per 16 lanes; with K = 64 .. 256 these
// layers are all epilogue.  The compute-bound 3x3 layers do not care (csrc/conv_wd.h), these do.
#pragma once
#include "conv_wd.h"

namespace wd {

// ------------------------------------------------------------------------------------------------------
// 1x1 (stride 1 | 2) / GEMM with the same skeleton: weights L2 -> VGPR in fragment order, pixels through a padded
// LDS ring (3 slots of BPX pixels x 64 channels, register-staged: chunk c+3 is loaded global -> VGPR and chunk c+2 is
// written VGPR -> LDS while chunk c feeds the MFMAs; one barrier per chunk = 4 K-steps), accumulators hold 32
// consecutive output channels per lane, and the epilogue - residual add (plain or nearest-2x upsampled FPN top-down),
// ReLU, fp16 - runs straight from registers with 64-byte contiguous residual loads and stores per lane.
// These layers are HBM-bound (res4 conv3: 472 MB per launch for 54 GFLOP): what matters is bytes in flight and no
// LDS transposition / extra barriers in the short K loop (K = 256: 4 chunks).
// ------------------------------------------------------------------------------------------------------
template <int WM, int WN, int TPX, int DEPTH>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv1x1_wd_kernel(pe::ConvWdArgs a) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int BPX = WM * TPX * 32;
    constexpr int NP = BPX * 8 / THREADS;          // 16-byte slab pieces per thread and chunk
    constexpr int SLOT_BYTES = BPX * SLAB_ROW_B;
    static_assert(BPX * 8 % THREADS == 0 && NP % 4 == 0 && DEPTH == 4, "pieces are spread evenly over the 4 K-steps of a chunk");
    constexpr int PPS = NP / 4;                    // pieces per K-step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;   // the n tiles of one pixel tile run together (L2)
    const int m0 = tile_m * BPX, n0 = tile_n * (WN * 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    // ---- slab pieces: byte offset of the source pixel's channel 0 (+ 16 B column), or out of range ----
    unsigned p_off[NP];
    int p_lds[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
        const int q = tid + n * THREADS;
        const int e = q >> 3, c = q & 7;
        const int m = m0 + e;
        p_lds[n] = e * SLAB_ROW_B + c * 16;
        p_off[n] = 0xFFFFFFF0u;
        if (m < a.M) {
            const int ow = m % a.Wo, t = m / a.Wo;
            const int oh = t % a.Ho, img = t / a.Ho;
            p_off[n] = (unsigned)(((img * a.H + oh * a.stride) * a.W + ow * a.stride) * a.Cin * 2 + c * 16);
        }
    }
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, a.N * a.H * a.W * a.Cin * 2, 0x00020000);
    const int NC = a.Cin / 64;
    half8 sreg[2][NP];
    auto slab_load1 = [&](int set, int n, int chunk) {
        const unsigned vo = (p_off[n] != 0xFFFFFFF0u && chunk < NC) ? p_off[n] + (unsigned)chunk * 128u : 0xFFFFFFF0u;
        sreg[set][n] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rin, vo, 0, 0));
    };
    auto slab_store1 = [&](int set, int n, int slot) {
        *reinterpret_cast<half8*>(smem + slot * SLOT_BYTES + p_lds[n]) = sreg[set][n];
    };

    int fb[TPX];
#pragma unroll
    for (int i = 0; i < TPX; ++i) fb[i] = ((wm * TPX + i) * 32 + (lane & 31)) * SLAB_ROW_B + (lane >> 5) * 16;

    const int KSEQ = a.Cin / 16;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wpk), 0, a.Cout * a.Cin * 2, 0x00020000);
    const int w_base = (tile_n * KSEQ * WN + wn) * 2048;
    half8 wf[DEPTH][2];
    auto w_load = [&](int slot, int kseq) {
        const int ks = kseq < KSEQ ? kseq : KSEQ - 1;
        const int so = w_base + ks * (WN * 2048);
        wf[slot][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, so, 0));
        wf[slot][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + 1024, so, 0));
    };

    // ---- prologue: chunks 0, 1 into slots 0, 1; chunk 2 in register set 0; weight ring primed.  All of these loads
    // and the residual's are in flight together; the accumulators start as bias + residual, so the epilogue has no
    // memory read left ----
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_load1(0, n, 0);
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_load1(1, n, 1);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) w_load(d, d);
    const int cb = n0 + wn * 64 + (lane >> 5) * 32;   // this lane's 32 consecutive output channels
    float16v acc[2][TPX];
    {
        const float* bp = a.bias + cb;
        float16v b[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = *reinterpret_cast<const float4*>(bp + blk * 16 + r4 * 4);
                b[blk][r4 * 4 + 0] = v.x; b[blk][r4 * 4 + 1] = v.y; b[blk][r4 * 4 + 2] = v.z; b[blk][r4 * 4 + 3] = v.w;
            }
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<_Float16*>(a.res ? a.res : a.in), 0,
            a.res_mode == 1 ? a.M * a.Cout * 2 : (a.res_mode == 2 ? a.N * a.resH * a.resW * a.Cout * 2 : 0), 0x00020000);
#pragma unroll
        for (int i = 0; i < TPX; ++i)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) acc[blk][i] = b[blk];
        // (two passes so that all residual loads are issued before the first conversion waits on one)
        half8 rv[TPX][4];
#pragma unroll
        for (int i = 0; i < TPX; ++i) {
            const int m = m0 + (wm * TPX + i) * 32 + (lane & 31);
            unsigned ro = 0xFFFFFFF0u;
            if (a.res_mode && m < a.M) {
                unsigned rp = (unsigned)m;
                if (a.res_mode == 2) {
                    const int ow = m % a.Wo, t = m / a.Wo;
                    const int oh = t % a.Ho, img = t / a.Ho;
                    rp = (unsigned)((img * a.resH + (oh >> 1)) * a.resW + (ow >> 1));
                }
                ro = (rp * (unsigned)a.Cout + (unsigned)cb) * 2u;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                rv[i][q] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rres, ro + (ro == 0xFFFFFFF0u ? 0u : (unsigned)q * 16u), 0, 0));
        }
#pragma unroll
        for (int i = 0; i < TPX; ++i)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[blk][i][hh * 8 + e] += (float)rv[i][blk * 2 + hh][e];
    }
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_store1(0, n, 0);
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_store1(1, n, 1);
#pragma unroll
    for (int n = 0; n < NP; ++n) slab_load1(0, n, 2);
    __syncthreads();

    half8 pf[2][TPX];
#pragma unroll
    for (int i = 0; i < TPX; ++i) pf[0][i] = *reinterpret_cast<const half8*>(smem + fb[i]);

    int cur = 0;   // ring slot of chunk c
    // two chunks per trip so that the register-set index is static
    for (int c0 = 0; c0 < NC; c0 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + h;
            if (c < NC) {   // wave-uniform
                const int nxt = cur == 2 ? 0 : cur + 1;
                const int nn = nxt == 2 ? 0 : nxt + 1;
                const unsigned char* sb = smem + cur * SLOT_BYTES;
                const unsigned char* sn = smem + nxt * SLOT_BYTES;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // chunk c+3 -> register set h^1 EARLY (step 0; that set's chunk c+1 was written out one chunk ago);
                    // chunk c+2 (register set h, requested a chunk ago) -> ring slot nn LATE (step 3): ~7 K-steps of slack
                    // for the HBM round trip.  Slot nn was last read during chunk c-1, every wave passed that barrier.
                    if (t == 0) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) slab_load1(h ^ 1, j, c + 3);
                    }
                    if (t == 3) {
#pragma unroll
                        for (int j = 0; j < NP; ++j) slab_store1(h, j, nn);
                    }
                    if (t + 1 < 4) {
#pragma unroll
                        for (int i = 0; i < TPX; ++i) pf[(t + 1) & 1][i] = *reinterpret_cast<const half8*>(sb + fb[i] + (t + 1) * 32);
                    } else {
#pragma unroll
                        for (int i = 0; i < TPX; ++i) pf[(t + 1) & 1][i] = *reinterpret_cast<const half8*>(sn + fb[i]);
                    }
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                        for (int i = 0; i < TPX; ++i)
                            acc[blk][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t][blk], pf[t & 1][i], acc[blk][i], 0, 0, 0);
                    w_load(t, c * 4 + t + DEPTH);
#pragma unroll
                    for (int i = 0; i < TPX; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (t == 0) __builtin_amdgcn_sched_group_barrier(0x020, NP, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TPX - 3, 0);
                    if (t == 3) __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();   // publishes chunk c+2, retires the reads of chunk c
                cur = nxt;
            }
        }
    }

    // ---- epilogue from registers: ReLU, fp16; 64 contiguous bytes per lane and pixel block ----
    if (a.relu) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int i = 0; i < TPX; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[blk][i][r] = fmaxf(acc[blk][i][r], 0.f);
    }
#pragma unroll
    for (int i = 0; i < TPX; ++i) {
        const int m = m0 + (wm * TPX + i) * 32 + (lane & 31);
        if (m >= a.M) continue;
        _Float16* o = a.out + (size_t)m * a.out_stride + cb;
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

template <int WM, int WN, int TPX, int DEPTH>
int launch_conv1x1_wd(pe::ConvWdArgs a, hipStream_t st) {
    constexpr int BPX = WM * TPX * 32;
    a.tiles_m = pe::ceil_div(a.M, BPX);
    a.tiles_n = a.Cout / (WN * 64);
    const size_t lds = (size_t)3 * BPX * SLAB_ROW_B;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_wd_kernel<WM, WN, TPX, DEPTH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
    hipLaunchKernelGGL((conv1x1_wd_kernel<WM, WN, TPX, DEPTH>), dim3(a.tiles_m * a.tiles_n), dim3(64 * WM * WN), lds, st, a);
    return PE_OK;
}


}  // namespace wd
