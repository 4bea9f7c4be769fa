// Weights-direct 3x3 convolution, fourth generation (round 3): 256 pixels x 64 channels per wave (TPX = 8, 256 accumulators, ONE
// wave per SIMD) in a PERSISTENT workgroup that software-pipelines across tiles.
//
// Why: the weight stream (L2 -> VGPR) is the kernel's biggest consumer of the CU's vector-memory path (2 KiB per wave and
// K-step = 32 B/clk/CU at 8 MFMA per K-step).  Twice the pixels per wave halve it per MFMA: the bare K-loop runs 1524-1569
// TFLOP/s at TPX = 8 against 1350-1364 at TPX = 4 (scripts/tpx_probe.hip).  But 256 accumulators mean one workgroup per CU, and
// the non-persistent TPX = 8 build of conv_wd.h loses all of that and more (1060 vs 1115 TFLOP/s) because nothing covers a
// workgroup's dispatch, prologue (first slabs from HBM, bias) and epilogue any more.  Hence: the workgroup stays resident and
// walks its tiles; the slab ring and the weight ring simply keep running across the tile boundary (the loads for the next
// tile's first two slab groups and first DEPTH weight K-steps are issued under the last groups of the current tile), and only
// the accumulator epilogue (ReLU + fp16 + 32 stores per lane + bias re-initialisation) is exposed once per tile.
// Everything else - packed weight layout, slab layout with explicit halo entries, fragment addressing, the K order - is
// csrc/conv_wd.h's (same pe_conv_wd_pack_weights records).
#pragma once
#include "conv_wd.h"

namespace wd8 {
using wd::float16v;
using wd::half8;
using wd::SLAB_ROW_B;

constexpr int TPX = 8, WN = 4, DEPTH = 4, THREADS = 256, BPX = 256;
constexpr int EMAX = BPX + BPX / 16;
constexpr int NP = (EMAX * 8 + THREADS - 1) / THREADS;   // 9 slab pieces (16 B) per thread and group

// ABL (measurement builds, wrong results): 1 no epilogue stores, 2 no slab traffic in the loop, 4 no weight loads in the loop, 8 no per-tile describe
template <int ABL = 0>
__global__ __launch_bounds__(THREADS, 1) void conv3x3_wd8_kernel(pe::ConvWdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int segp = a.seg + 2;
    const int E = a.nseg * segp;
    const int slab_bytes = E * SLAB_ROW_B;
    const int G = 3 * (a.Cin / 64);
    const int KSEQ = G * 12;

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

    // ---- slab pieces.  Piece n of this thread is 16-byte chunk (tid & 7) of slab entry e = (tid >> 3) + 32 n.  Its LDS target is
    // tile independent; its source (byte offset of the centre-row pixel or -1, image row) depends on the tile and lives in LDS -
    // two descriptor sets (the tile being loaded, the one after it), rewritten once per tile - because 256 accumulators leave no
    // room for 36 descriptor registers: the K-loop reads one 8-byte descriptor per K-step, a step ahead of its use.
    const int e0 = tid >> 3, c8 = (tid & 7) * 8;
    const int dummy = 3 * slab_bytes + tid * 16;
    int2* desc = reinterpret_cast<int2*>(smem + 3 * slab_bytes + THREADS * 16);      // [2 sets][NP][THREADS]
    auto describe = [&](int set, int tile) {   // tile < 0: no such tile -> every piece invalid (zeros are loaded and never used)
        const int m0 = tile >= 0 ? (tile / a.tiles_n) * BPX : 0;
#pragma unroll
        for (int n = 0; n < NP; ++n) {
            const int e = e0 + n * 32;
            int off = -1, hrow = 0;
            if (tile >= 0 && e < E) {
                const int sg = e / segp, jj = e - sg * segp - 1;
                const int P0 = m0 + sg * a.seg;
                const int row = P0 / a.W, col = P0 - row * a.W + jj;
                hrow = row % a.H;
                if (P0 < a.M && (unsigned)col < (unsigned)a.W) off = ((P0 + jj) * a.Cin + c8) * 2;
            }
            desc[(set * NP + n) * THREADS + tid] = make_int2(off, hrow);
        }
    };
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.in), 0, a.M * a.Cin * 2, 0x00020000);
    half8 sreg[4];                              // piece n is loaded in K-step n and stored in K-step n + 3: four in flight at most
    auto slab_load1 = [&](int2 d, int n, int g) {
        const int cc = g / 3, kh = g - cc * 3;
        const int shift = ((kh - 1) * a.W * a.Cin + cc * 64) * 2;
        const bool ok = d.x >= 0 && (unsigned)(d.y + kh - 1) < (unsigned)a.H;
        const unsigned vo = ok ? (unsigned)(d.x + shift) : 0xFFFFFFF0u;
        sreg[n & 3] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rin, vo, 0, 0));
    };
    auto slab_store1 = [&](int n, int buf) {
        const int lds = e0 + n * 32 < E ? buf * slab_bytes + (e0 + n * 32) * SLAB_ROW_B + (tid & 7) * 16 : dummy;
        *reinterpret_cast<half8*>(smem + lds) = sreg[n & 3];
    };

    int fb[TPX];
#pragma unroll
    for (int i = 0; i < TPX; ++i) {
        const int p = i * 32;
        const int s = p / a.seg, j0 = p - s * a.seg;
        fb[i] = (s * segp + j0 + (lane & 31)) * SLAB_ROW_B + (lane >> 5) * 16;
    }

    // ---- weight stream ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.wpk), 0, a.Cout * a.Cin * 18, 0x00020000);
    auto wbase_of = [&](int tile) { return ((tile % a.tiles_n) * KSEQ * WN + wn) * 2048; };
    int w_cur = wbase_of(t_first), w_nxt = nmine > 1 ? wbase_of(t_first + t_step) : w_cur;
    half8 wf[DEPTH][2];
    auto w_load = [&](int slot, int kseq) {   // kseq >= KSEQ: the next tile's stream (or a harmless re-read at the very end)
        const int so = kseq < KSEQ ? w_cur + kseq * (WN * 2048) : w_nxt + (kseq - KSEQ) * (WN * 2048);
        wf[slot][0] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, so, 0));
        wf[slot][1] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + 1024, so, 0));
    };

    float16v acc[2][TPX];
    auto init_acc = [&](int tile) {
        const float* bp = a.bias + (tile % a.tiles_n) * (WN * 64) + wn * 64 + (lane >> 5) * 32;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float16v b;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float4 v = *reinterpret_cast<const float4*>(bp + blk * 16 + r4 * 4);
                b[r4 * 4 + 0] = v.x; b[r4 * 4 + 1] = v.y; b[r4 * 4 + 2] = v.z; b[r4 * 4 + 3] = v.w;
            }
#pragma unroll
            for (int i = 0; i < TPX; ++i) acc[blk][i] = b;
        }
    };

    // ---- prologue (once per workgroup): descriptors, slabs 0 and 1 of the first tile, weight ring primed ----
    int dset = 0;                               // descriptor set of the tile whose K-loop is running
    describe(0, t_first);
    describe(1, nmine > 1 ? t_first + t_step : -1);
    init_acc(t_first);
    __syncthreads();
    for (int g0 = 0; g0 < 2; ++g0) {
#pragma unroll
        for (int n0 = 0; n0 < NP; n0 += 4) {
#pragma unroll
            for (int n = n0; n < n0 + 4 && n < NP; ++n) slab_load1(desc[n * THREADS + tid], n, G > 1 ? g0 : 0);
#pragma unroll
            for (int n = n0; n < n0 + 4 && n < NP; ++n) slab_store1(n, g0);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) w_load(d, d);
    __syncthreads();

    half8 pf[2][TPX];
    int cur = 0;
#pragma unroll
    for (int i = 0; i < TPX; ++i) pf[0][i] = *reinterpret_cast<const half8*>(smem + fb[i]);

    for (int j = 0; j < nmine; ++j) {
        const int tile = t_first + j * t_step;
        for (int g = 0; g < G; ++g) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            const int nn = nxt == 2 ? 0 : nxt + 1;
            // slab two groups ahead: group g + 2 of this tile, or group g + 2 - G of the next one (descriptor set 1)
            const bool ahead_next = g + 2 >= G;
            const int gl = ahead_next ? g + 2 - G : g + 2;
            const int2* dg = desc + ((ahead_next ? dset ^ 1 : dset) * NP) * THREADS + tid;
            int2 dnow = dg[0];
            const unsigned char* sb = smem + cur * slab_bytes;
            const unsigned char* sn = smem + nxt * slab_bytes;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                if (!(ABL & 2) && t < NP) slab_load1(dnow, t, gl);
                if (!(ABL & 2) && t + 1 < NP) dnow = dg[(t + 1) * THREADS];
                if (t + 1 < 12) {
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
                if (!(ABL & 4)) w_load(slot, g * 12 + t + DEPTH);
                if (!(ABL & 2) && t >= 12 - NP) slab_store1(t - (12 - NP), nn);
#pragma unroll
                for (int i = 0; i < TPX; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // (the next piece's descriptor)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TPX - 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            cur = nxt;
        }
        // ---- tile epilogue: ReLU + fp16, 64 contiguous bytes per lane and pixel block; then the next tile's bias ----
        {
            const int m0 = (tile / a.tiles_n) * BPX, n0 = (tile % a.tiles_n) * (WN * 64);
#pragma unroll
            for (int i = 0; i < TPX; ++i) {
                const int m = m0 + i * 32 + (lane & 31);
                if (m >= a.M) continue;
                _Float16* o = a.out + (size_t)m * a.out_stride + n0 + wn * 64 + (lane >> 5) * 32;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        half8 v;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float x = acc[blk][i][hh * 8 + e];
                            v[e] = (_Float16)(a.relu ? pe::relu_nan(x) : x);
                        }
                        if (!(ABL & 1) || v[0] == (_Float16)12345.f) *reinterpret_cast<half8*>(o + blk * 16 + hh * 8) = v;
                    }
            }
        }
        if (j + 1 < nmine) {
            const int t1 = tile + t_step;
            init_acc(t1);
            w_cur = w_nxt;
            w_nxt = j + 2 < nmine ? wbase_of(t1 + t_step) : w_cur;
            // the finished tile's descriptor set is free (its last slab loads were issued two groups ago; this tile's loads of
            // the running groups read the OTHER set): describe the tile after the next into it
            if (!(ABL & 8)) describe(dset, j + 2 < nmine ? t1 + t_step : -1);
            dset ^= 1;
            __syncthreads();
        }
    }
}

template <int ABL = 0>
inline int launch(pe::ConvWdArgs a, hipStream_t st, int workgroups = 256) {
    if (!wd::wd3x3_geometry(a.W, BPX, &a.seg, &a.nseg)) return PE_ERR_UNSUPPORTED;
    a.tiles_m = pe::ceil_div(a.M, BPX);
    a.tiles_n = a.Cout / (WN * 64);
    const size_t lds = (size_t)3 * a.nseg * (a.seg + 2) * SLAB_ROW_B + (size_t)THREADS * 16 + (size_t)2 * NP * THREADS * 8;
    PE_ENSURE_LDS(conv3x3_wd8_kernel<ABL>, lds, "conv3x3_wd8");
    const int ntile = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL(conv3x3_wd8_kernel<ABL>, dim3(ntile < workgroups ? ntile : workgroups), dim3(THREADS), lds, st, a);
    return PE_OK;
}

}  // namespace wd8
