// Batched (per image, class-aware) greedy NMS for gfx950.
//
// Replaces detectron2.layers.batched_nms (layers/nms.py:20-37) -> torchvision.ops.boxes.batched_nms /
// nms (torchvision 0.13.0, not vendored), as called from find_top_rpn_proposals
// (proposal_generator/rpn_outputs.py:147), fast_rcnn_inference_single_image (roi_heads/fast_rcnn.py:130)
// and nms_1 (demo/FLIR/demo_probEn.py:64).  float32 IoU = inter / (a + b - inter), suppress on IoU > thr,
// no "+1"; PE_NMS_TRICK adds idx * (max_coord + 1) to the coordinates first (torchvision's
// "coordinate trick"), PE_NMS_CLASS compares class ids instead (== one nms per class, "vanilla").
// Order rule: score descending, ties by input index ascending (stable descending sort).
//
// Suppression only ever happens inside a class (mode 1 by definition; mode 0 because boxes of different classes are moved
// max_coord + 1 apart and cannot intersect), so the work is organised per class:
//   1. sort:  one 1024-thread block per image, bitonic sort of 64-bit keys (class | ~score | index) in LDS: boxes end up
//             grouped by class, score-descending inside a class; the class segments are listed
//   2. mask:  64 x 64 tiles of the sorted order -> suppression bit matrix; tiles whose row and column class ranges are
//             disjoint (4 of every 5 for the five-level RPN input) are zero-filled without a single IoU
//   3. scan + order: one 1024-thread block per image; every wavefront runs the greedy scan of whole class segments
//             (<= 16 chunks for an RPN level instead of 73 for the image, early exit at max_out per class), then the block
//             re-sorts the survivors by (~score | index) and emits the first max_out.
#include "common.h"
#include "test_hooks.h"

namespace {

constexpr int kSortThreads = 1024;
__device__ int g_presorted_path = 1;     // test hook (pe_test_set_nms_presorted): 0 = always run the sorting network

__device__ __forceinline__ unsigned ordered_desc(float s) {
    s += 0.0f;  // -0 -> +0
    unsigned u = __float_as_uint(s);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending order for floats
    return ~u;                                  // descending
}

struct NmsArgs {
    const float* boxes;    // [B, n_max, 4]
    const float* scores;   // [B, n_max]
    const int32_t* idxs;   // [B, n_max] class / level ids (may be null: single class)
    const int32_t* counts; // [B] rows used per image (may be null: n_max)
    const uint8_t* valid;  // [B, n_max] optional row mask (rows with 0 are ignored)
    int B, n_max, n_pad;   // n_pad = power of two >= n_max
    float thr;
    int mode, max_out;
    // scratch
    float* sboxes;         // [B, n_max, 4] sorted (+ offset) boxes
    int32_t* sidx;         // [B, n_max] sorted -> input row
    int32_t* scls;         // [B, n_max]
    int32_t* nvalid;       // [B]
    unsigned* sord;        // [B, n_max] order-preserving score bits of the sorted rows
    int32_t* seg_start;    // [B, n_max] first sorted position of every class segment (any order)
    int32_t* nseg;         // [B]
    unsigned long long* mask;  // [B, n_max, words]
    int words;
    int32_t* out_keep;     // [B, max_out]
    int32_t* out_counts;   // [B]
    int merge;             // scan + order: the LDS holds the survivors' keys behind the kept lists (merge by rank instead of a sort)
};

__global__ __launch_bounds__(kSortThreads) void nms_sort_kernel(NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ float red[kSortThreads / 64], red_mn[kSortThreads / 64];
    __shared__ int cnt_s, seg_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.counts ? min(a.counts[b], a.n_max) : a.n_max;
    const float* sc = a.scores + (size_t)b * a.n_max;
    const float* bx = a.boxes + (size_t)b * a.n_max * 4;
    if (tid == 0) { cnt_s = 0; seg_s = 0; }
    __syncthreads();
    // pass 1: extrema of the coordinates of the live rows.  max -> the coordinate trick's per-class offset; min decides whether
    // the per-class organisation is legal for mode 0: boxes + idx * (max + 1) keeps classes apart only while every coordinate
    // is >= 0 (a box with coordinates below -1 reaches into the previous class's band and torchvision's trick lets the two
    // suppress each other).  Such an image ("generic") is handled as ONE segment with all pairs compared on the shifted boxes.
    float mx = -INFINITY, mn = INFINITY;
    for (int i = tid; i < n; i += kSortThreads) {
        if (a.valid && !a.valid[(size_t)b * a.n_max + i]) continue;
        const float lo = fminf(fminf(bx[i * 4], bx[i * 4 + 1]), fminf(bx[i * 4 + 2], bx[i * 4 + 3]));
        const float hi = fmaxf(fmaxf(bx[i * 4], bx[i * 4 + 1]), fmaxf(bx[i * 4 + 2], bx[i * 4 + 3]));
        mx = fmaxf(mx, hi); mn = fminf(mn, lo);
    }
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); mn = fminf(mn, __shfl_xor(mn, o)); }
    if ((tid & 63) == 0) { red[tid >> 6] = mx; red_mn[tid >> 6] = mn; }
    __syncthreads();
    mx = red[0]; mn = red_mn[0];
    for (int w = 1; w < kSortThreads / 64; ++w) { mx = fmaxf(mx, red[w]); mn = fminf(mn, red_mn[w]); }
    const bool generic = a.mode == 0 && a.idxs && mn < 0.f;
    // pass 2: keys
    int local_cnt = 0;
    for (int i = tid; i < a.n_pad; i += kSortThreads) {
        unsigned long long k = ~0ull;
        if (i < n && (!a.valid || a.valid[(size_t)b * a.n_max + i])) {
            const unsigned long long c = (a.idxs && !generic) ? ((unsigned)a.idxs[(size_t)b * a.n_max + i] & 0x3FFFFu) : 0u;   // class: 18 bits
            k = (c << 46) | ((unsigned long long)ordered_desc(sc[i]) << 14) | (unsigned)i;                       // index: 14 bits (n_max <= 16384)
            ++local_cnt;
        }
        keys[i] = k;
    }
    atomicAdd(&cnt_s, local_cnt);
    __syncthreads();
    // Already in order?  The RPN hands over every level's top-k in score order, the levels one after the other: the live keys
    // (class | ~score | row) are strictly increasing as they stand and the 91 passes of the network below (n_pad = 8192) would only
    // move the dead rows' ~0 keys to the end.  Each thread takes n_pad / 1024 CONSECUTIVE keys into registers and checks its own
    // stretch; a prefix maximum over the threads' last live keys checks the seams (an out-of-order pair across a seam has the later
    // key <= the maximum of everything before it); a prefix sum of the live counts gives the compaction offsets.  In order: the
    // live keys are written back closed up, the rest filled with ~0 - exactly the network's result (the keys are unique).
    bool presorted = false;
    if (g_presorted_path) {
        constexpr int EMAX = 16;                   // n_pad <= 16384 = 1024 threads x 16
        const int E = a.n_pad / kSortThreads;      // 0 when n_pad < 1024: not worth it, sort
        __shared__ unsigned long long wave_max[kSortThreads / 64];
        __shared__ int wave_sum[kSortThreads / 64];
        __shared__ int bad_s;
        if (tid == 0) bad_s = 0;
        unsigned long long mine[EMAX];
        unsigned long long first = ~0ull, last = 0ull;   // live keys are < ~0 and, past the first row, > 0 (row index bits)
        int live = 0;
        bool bad = false, any = false;
        if (E > 0) {
#pragma unroll
            for (int j = 0; j < EMAX; ++j) {
                unsigned long long k = ~0ull;
                if (j < E) k = keys[tid * E + j];
                mine[j] = k;
                if (k != ~0ull) {
                    if (any && k <= last) bad = true;
                    if (!any) first = k;
                    last = k; any = true; ++live;
                }
            }
        }
        // inclusive scans over the block: live counts (sum) and last live keys (max; 0 = none so far)
        int ps = live;
        unsigned long long pm = any ? last : 0ull;
        for (int o = 1; o < 64; o <<= 1) {
            const int vs = __shfl_up(ps, o);
            const unsigned long long vm = __shfl_up(pm, o);
            if ((tid & 63) >= o) { ps += vs; pm = vm > pm ? vm : pm; }
        }
        if ((tid & 63) == 63) { wave_sum[tid >> 6] = ps; wave_max[tid >> 6] = pm; }
        __syncthreads();
        int off = ps - live;                                  // exclusive within the wave
        unsigned long long before = __shfl_up(pm, 1);         // maximum of the lanes before this one (wave-local)
        if ((tid & 63) == 0) before = 0ull;
        for (int w = 0; w < (tid >> 6); ++w) { off += wave_sum[w]; before = wave_max[w] > before ? wave_max[w] : before; }
        if (any && before != 0ull && first <= before) bad = true;
        // (a live key of 0 can only be row 0 of class 0 with the best possible score: it is the very first key or out of order;
        //  `before` = 0 then means "nothing before" only for thread 0, and a real 0 before a later key never violates the order)
        if (bad) bad_s = 1;
        __syncthreads();
        presorted = E > 0 && bad_s == 0;
        if (presorted) {
            const int nvv = cnt_s;
#pragma unroll
            for (int j = 0; j < EMAX; ++j)
                if (j < E && mine[j] != ~0ull) keys[off++] = mine[j];      // every thread read its stretch before the barrier above
            for (int i = nvv + tid; i < a.n_pad; i += kSortThreads) keys[i] = ~0ull;
            __syncthreads();
        }
    }
    // bitonic sort ascending
    if (!presorted)
    for (int k = 2; k <= a.n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < a.n_pad; i += kSortThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    const int nv = cnt_s;
    if (tid == 0) a.nvalid[b] = nv;
    const float off1 = mx + 1.0f;
    for (int p = tid; p < nv; p += kSortThreads) {
        const int i = (int)(keys[p] & 0x3FFFu);
        const int c = a.idxs ? a.idxs[(size_t)b * a.n_max + i] : 0;
        a.sord[(size_t)b * a.n_max + p] = (unsigned)(keys[p] >> 14);
        if (p == 0 || (keys[p] >> 46) != (keys[p - 1] >> 46)) a.seg_start[(size_t)b * a.n_max + atomicAdd(&seg_s, 1)] = p;
        const float off = a.mode == 0 ? (float)c * off1 : 0.f;
        float* o = a.sboxes + ((size_t)b * a.n_max + p) * 4;
        o[0] = bx[i * 4] + off; o[1] = bx[i * 4 + 1] + off; o[2] = bx[i * 4 + 2] + off; o[3] = bx[i * 4 + 3] + off;
        a.sidx[(size_t)b * a.n_max + p] = i;
        a.scls[(size_t)b * a.n_max + p] = generic ? 0 : c;   // generic: one segment, every pair compared on the shifted boxes
    }
    __syncthreads();
    if (tid == 0) a.nseg[b] = seg_s;
}

// grid (col_tiles, row_tiles, B), block 64: thread t handles sorted row (row_tile*64 + t) against 64 columns.
// (Round 5 tried one 256-thread workgroup per row tile walking only the column tiles it needs - 2 336 workgroups instead of
// 170 000, 4 of 5 of which only find out that they have nothing to do: 92 -> 134 us for the RPN's launch.  The 21 800 real tiles are
// ~65 us of IoU arithmetic at full occupancy, and a workgroup that takes its tiles one after the other exposes every LDS round trip
// of the inner loop; one tiny workgroup per tile is what keeps 8 waves per SIMD in flight.)
__global__ __launch_bounds__(64) void nms_mask_kernel(NmsArgs a) {
    const int b = blockIdx.z;
    const int nv = a.nvalid[b];
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (row0 >= nv || col0 >= nv || col0 + 63 < row0) return;  // only the upper triangle matters
    __shared__ float cb[64][4];
    __shared__ int cc[64];
    const int t = threadIdx.x;
    const float* sb = a.sboxes + (size_t)b * a.n_max * 4;
    {   // sorted by (masked) class: the tile's columns all come after its rows; disjoint class ranges -> nothing to suppress
        const int* cls = a.scls + (size_t)b * a.n_max;
        const unsigned row_last = (unsigned)cls[min(row0 + 63, nv - 1)] & 0x3FFFFu, col_first = (unsigned)cls[col0] & 0x3FFFFu;
        if (col_first > row_last) {
            if (row0 + t < nv) a.mask[((size_t)b * a.n_max + row0 + t) * a.words + blockIdx.x] = 0ull;
            return;
        }
    }
    if (col0 + t < nv) {
        cb[t][0] = sb[(col0 + t) * 4]; cb[t][1] = sb[(col0 + t) * 4 + 1];
        cb[t][2] = sb[(col0 + t) * 4 + 2]; cb[t][3] = sb[(col0 + t) * 4 + 3];
        cc[t] = a.scls[(size_t)b * a.n_max + col0 + t];
    }
    __syncthreads();
    const int r = row0 + t;
    if (r >= nv) return;
    const float x1 = sb[r * 4], y1 = sb[r * 4 + 1], x2 = sb[r * 4 + 2], y2 = sb[r * 4 + 3];
    const float ar = (x2 - x1) * (y2 - y1);
    const int rc = a.scls[(size_t)b * a.n_max + r];
    unsigned long long bits = 0;
    const int jn = min(64, nv - col0);
    for (int j = (col0 == row0 ? t + 1 : 0); j < jn; ++j) {
        if (col0 + j <= r) continue;
        if (cc[j] != rc) continue;   // mode 1 by definition; mode 0: shifted apart by (max_coord + 1) per class -> IoU 0 (images with
                                     // negative coordinates carry class 0 everywhere here: nms_sort_kernel's "generic" case)
        const float w = fmaxf(0.f, fminf(x2, cb[j][2]) - fmaxf(x1, cb[j][0]));
        const float h = fmaxf(0.f, fminf(y2, cb[j][3]) - fmaxf(y1, cb[j][1]));
        const float inter = w * h;
        const float aj = (cb[j][2] - cb[j][0]) * (cb[j][3] - cb[j][1]);
        const float iou = inter / (ar + aj - inter);
        if (iou > a.thr) bits |= 1ull << j;
    }
    a.mask[((size_t)b * a.n_max + r) * a.words + blockIdx.x] = bits;
}

// One 1024-thread block per image.  Phase 1: wavefront w runs the greedy scan of class segments w, w + 16, ... ("pull" form,
// 64 sorted boxes per step):
//   1. the removed word of chunk c = OR over the boxes of this segment kept so far of their mask word c: the kept list
//      lives in LDS, lanes stride over it with independent loads (one memory latency per chunk), then a wave OR;
//   2. lane l fetches the DIAGONAL word of row 64c+l and the chunk is resolved in registers with scalar readlanes;
//   3. survivors are appended to the segment's kept list and flagged; a segment stops at max_out survivors (no class can
//      place more than that in the final top max_out).
// Phase 2: the survivors leave in (~score | input index) order, the first max_out of them.  A segment's kept list already is in
// that order, so with few segments (<= 16: the RPN's five levels, the FLIR heads' three classes) every survivor's place is its
// place in its own list plus, per other segment, the number of that segment's survivors in front of it - a binary search over the
// survivors' keys in LDS (~10 probes x 4 lists) instead of the 91 passes of a sorting network over n_pad keys.  More segments (the
// 80-class head) or no room for the keys: the flagged rows' keys are bitonic-sorted by the whole block.
constexpr int kScanThreads = 1024;
constexpr int kMergeSegs = 16;
__global__ __launch_bounds__(kScanThreads) void nms_scan_order_kernel(NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // phase 1 layout: kept_pos [n_max] u16 | flags [n_pad / 32] u32;  phase 2: keys [n_pad] u64 over the same bytes
    unsigned short* kept_pos = reinterpret_cast<unsigned short*>(smem);
    unsigned* flags = reinterpret_cast<unsigned*>(smem + (((size_t)a.n_max * 2 + 15) & ~(size_t)15));
    __shared__ int total_s;
    __shared__ int seg_p0_s[kMergeSegs], seg_kept_s[kMergeSegs];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = a.nvalid[b];
    const int nseg = a.nseg[b];
    const unsigned long long* mk = a.mask + (size_t)b * a.n_max * a.words;
    for (int i = tid; i < a.n_pad / 32; i += kScanThreads) flags[i] = 0u;
    if (tid == 0) total_s = 0;
    __syncthreads();
    for (int sgi = wave; sgi < nseg; sgi += kScanThreads / 64) {
        const int p0 = a.seg_start[(size_t)b * a.n_max + sgi];
        const int cls0 = a.scls[(size_t)b * a.n_max + p0];
        // segment end: the class changes (segments are listed in no particular order, so walk the class array)
        int p1;
        {
            int lo = p0, hi = nv;   // first position >= p0 whose class differs (classes are sorted: binary search on equality run)
            const unsigned key0 = (unsigned)cls0 & 0x3FFFFu;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (((unsigned)a.scls[(size_t)b * a.n_max + mid] & 0x3FFFFu) == key0) lo = mid + 1; else hi = mid;
            }
            p1 = lo;
        }
        unsigned short* kp = kept_pos + p0;
        int kept = 0;
        for (int c = p0 >> 6; c <= ((p1 - 1) >> 6) && kept < a.max_out; ++c) {
            const int i0 = c * 64;
            const int row = i0 + lane;
            const bool mine = row >= p0 && row < p1;
            const unsigned long long diag = mine ? mk[(size_t)row * a.words + c] : 0ull;     // in flight together with the gather below
            unsigned long long rem = 0;
            constexpr int GB = 16;      // a lane's loads of one batch: all issued before the first is waited for (a segment that stops
            for (int k0 = 0; k0 < kept; k0 += 64 * GB) {     // at max_out <= 1024 survivors is ONE batch = one memory latency per chunk;
                int pp[GB];                                  // one load per loop trip cost sixteen of them: 27 k cycles per chunk)
#pragma unroll
                for (int u = 0; u < GB; ++u) {
                    const int k = k0 + u * 64 + lane;
                    pp[u] = k < kept ? (int)kp[k] : 0x7FFFFFFF;
                }
                unsigned long long v[GB];
#pragma unroll
                for (int u = 0; u < GB; ++u) v[u] = (pp[u] >> 6) < c ? mk[(size_t)pp[u] * a.words + c] : 0ull;
#pragma unroll
                for (int u = 0; u < GB; ++u) rem |= v[u];
            }
            for (int o = 32; o > 0; o >>= 1) rem |= __shfl_xor(rem, o);
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            const int b_lo = max(p0 - i0, 0), b_hi = min(p1 - i0, 64);
            unsigned long long keepmask = 0;
            int kept_here = 0;
            for (int bit = b_lo; bit < b_hi; ++bit) {
                if ((rem >> bit) & 1ull) continue;          // wave-uniform
                if (kept + kept_here >= a.max_out) break;
                keepmask |= 1ull << bit;
                ++kept_here;
                const unsigned lo = __builtin_amdgcn_readlane(dlo, bit), hi = __builtin_amdgcn_readlane(dhi, bit);
                rem |= ((unsigned long long)hi << 32) | lo;
            }
            if ((keepmask >> lane) & 1ull) {
                kp[kept + __popcll(keepmask & pe::lanemask_lt())] = (unsigned short)row;
                atomicOr(&flags[row >> 5], 1u << (row & 31));
            }
            kept += kept_here;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the kept list is re-read by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            atomicAdd(&total_s, kept);
            if (sgi < kMergeSegs) { seg_p0_s[sgi] = p0; seg_kept_s[sgi] = kept; }
        }
    }
    __syncthreads();
    if (a.merge && nseg <= kMergeSegs) {
        // ---- phase 2, merge by rank ----
        unsigned long long* key2 = reinterpret_cast<unsigned long long*>(smem + (((((size_t)a.n_max * 2 + 15) & ~(size_t)15) + (size_t)a.n_pad / 8 + 15) & ~(size_t)15));
        int off[kMergeSegs + 1];
        off[0] = 0;
#pragma unroll
        for (int s = 0; s < kMergeSegs; ++s) off[s + 1] = off[s] + (s < nseg ? seg_kept_s[s] : 0);
        const int total = total_s;
        for (int g = tid; g < total; g += kScanThreads) {
            int s = 0;
            while (g >= off[s + 1]) ++s;
            const int p = kept_pos[seg_p0_s[s] + (g - off[s])];
            key2[g] = ((unsigned long long)a.sord[(size_t)b * a.n_max + p] << 32) | (unsigned)a.sidx[(size_t)b * a.n_max + p];
        }
        __syncthreads();
        for (int g = tid; g < total; g += kScanThreads) {
            int s = 0;
            while (g >= off[s + 1]) ++s;
            const unsigned long long x = key2[g];
            int rank = g - off[s];
            for (int t = 0; t < nseg; ++t) {
                if (t == s) continue;
                int lo = off[t], hi = off[t + 1];         // first key of list t that is not < x (keys are unique)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (key2[mid] < x) lo = mid + 1; else hi = mid;
                }
                rank += lo - off[t];
            }
            if (rank < a.max_out) a.out_keep[(size_t)b * a.max_out + rank] = (int)(x & 0xFFFFFFFFu);
        }
        if (tid == 0) a.out_counts[b] = min(total, a.max_out);
        return;
    }
    // ---- phase 2: keys of the flagged rows (flags are read into registers before the key array overwrites them) ----
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    constexpr int KPT = 16;   // n_pad <= 16384 = 1024 threads x 16
    unsigned long long mykeys[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int p = tid + j * kScanThreads;
        unsigned long long k = ~0ull;
        if (p < nv && ((flags[p >> 5] >> (p & 31)) & 1u))
            k = ((unsigned long long)(a.sord[(size_t)b * a.n_max + p] & 0xFFFFFFFFu) << 32) | (unsigned)a.sidx[(size_t)b * a.n_max + p];
        mykeys[j] = k;
    }
    const int total = total_s;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int p = tid + j * kScanThreads;
        if (p < a.n_pad) keys[p] = mykeys[j];
    }
    __syncthreads();
    for (int k = 2; k <= a.n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < a.n_pad; i += kScanThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    const int nout = min(total, a.max_out);
    for (int i = tid; i < nout; i += kScanThreads) a.out_keep[(size_t)b * a.max_out + i] = (int)(keys[i] & 0xFFFFFFFFu);
    if (tid == 0) a.out_counts[b] = nout;
}

}  // namespace

extern "C" int pe_test_set_nms_presorted(int on) {
    const int v = on ? 1 : 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_presorted_path), &v, sizeof(int));
}

extern "C" size_t pe_nms_scratch_bytes(int32_t B, int32_t n_max) {
    const size_t words = ((size_t)n_max + 63) / 64;
    size_t s = 0;
    s += (size_t)B * n_max * 4 * sizeof(float);    // sboxes
    s += (size_t)B * n_max * sizeof(int32_t) * 2;  // sidx, scls
    s += (size_t)B * sizeof(int32_t) * 2 + 128;    // nvalid, nseg
    s += (size_t)B * n_max * sizeof(int32_t) * 2;  // sord, seg_start
    s += (size_t)B * n_max * words * 8;            // mask
    return s + 512;
}

extern "C" int pe_nms_batched(const float* boxes, const float* scores, const int32_t* idxs, const int32_t* counts,
                              const uint8_t* valid, int32_t B, int32_t n_max, float iou_thresh, int32_t mode,
                              int32_t max_out, int32_t* out_keep, int32_t* out_counts, void* scratch,
                              size_t scratch_bytes, void* stream) {
    PE_CHECK_ARG(B >= 0 && n_max >= 0, "pe_nms_batched: negative sizes");
    if (B == 0) return PE_OK;
    PE_CHECK_ARG(out_keep && out_counts, "pe_nms_batched: null output");
    PE_CHECK_ARG(mode == 0 || mode == 1, "pe_nms_batched: mode %d", mode);
    PE_CHECK_ARG(max_out >= 1, "pe_nms_batched: max_out < 1");
    hipStream_t st = (hipStream_t)stream;
    if (n_max == 0) {
        (void)hipMemsetAsync(out_counts, 0, sizeof(int32_t) * B, st);
        return PE_OK;
    }
    PE_CHECK_ARG(boxes && scores && scratch, "pe_nms_batched: null pointer");
    PE_CHECK_ARG(n_max <= 16384, "pe_nms_batched: n_max %d > 16384", n_max);
    PE_CHECK_ARG(scratch_bytes >= pe_nms_scratch_bytes(B, n_max), "pe_nms_batched: scratch too small (%zu < %zu)",
                 scratch_bytes, pe_nms_scratch_bytes(B, n_max));
    NmsArgs a{};
    a.boxes = boxes; a.scores = scores; a.idxs = idxs; a.counts = counts; a.valid = valid;
    a.B = B; a.n_max = n_max; a.thr = iou_thresh; a.mode = mode; a.max_out = max_out;
    a.n_pad = 64;
    while (a.n_pad < n_max) a.n_pad <<= 1;
    a.words = (n_max + 63) / 64;
    unsigned char* p = (unsigned char*)scratch;
    auto carve = [&](size_t bytes) { void* q = p; p += (bytes + 63) & ~(size_t)63; return q; };
    a.mask = (unsigned long long*)carve((size_t)B * n_max * a.words * 8);
    a.sboxes = (float*)carve((size_t)B * n_max * 16);
    a.sidx = (int32_t*)carve((size_t)B * n_max * 4);
    a.scls = (int32_t*)carve((size_t)B * n_max * 4);
    a.nvalid = (int32_t*)carve((size_t)B * 4);
    a.nseg = (int32_t*)carve((size_t)B * 4);
    a.sord = (unsigned*)carve((size_t)B * n_max * 4);
    a.seg_start = (int32_t*)carve((size_t)B * n_max * 4);
    a.out_keep = out_keep; a.out_counts = out_counts;
    const size_t lds = (size_t)a.n_pad * 8;
    PE_ENSURE_LDS(nms_sort_kernel, lds + 512, "pe_nms_batched(sort)");   // + the kernel's static LDS (~332 B of reduction / segment scratch)
    hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(kSortThreads), lds, st, a);
    PE_CHECK_LAUNCH("pe_nms_batched(sort)");
    hipLaunchKernelGGL(nms_mask_kernel, dim3(a.words, a.words, B), dim3(64), 0, st, a);
    PE_CHECK_LAUNCH("pe_nms_batched(mask)");
    const size_t lists = (((((size_t)n_max * 2 + 15) & ~(size_t)15) + (size_t)a.n_pad / 8 + 15) & ~(size_t)15);   // kept lists + flags
    size_t lds2 = std::max((size_t)a.n_pad * 8, lists + 16);
    a.merge = lists + (size_t)n_max * 8 + 512 <= 160 * 1024;       // + the survivors' keys (at most one per row)
    if (a.merge) lds2 = std::max(lds2, lists + (size_t)n_max * 8);
    PE_ENSURE_LDS(nms_scan_order_kernel, lds2 + 512, "pe_nms_batched(scan + order)");   // + the kernel's static LDS (~132 B)
    hipLaunchKernelGGL(nms_scan_order_kernel, dim3(B), dim3(kScanThreads), lds2, st, a);
    PE_CHECK_LAUNCH("pe_nms_batched(scan + order)");
    return PE_OK;
}
