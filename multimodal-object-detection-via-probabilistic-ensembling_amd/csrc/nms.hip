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
// Three launches per batch, all images in flight at once:
//   1. sort:  one 1024-thread block per image, bitonic sort of 64-bit keys (~score | index) in LDS
//   2. mask:  64x64 box tiles -> suppression bit matrix [n][ceil(n/64)] (one wavefront row per block)
//   3. scan:  one wavefront per image walks the sorted boxes, OR-ing kept rows into a 64-lane register
//             mask (lane l owns words l, l+64, ...), early exit at max_out kept.
#include "common.h"

namespace {

constexpr int kSortThreads = 1024;

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
    unsigned long long* mask;  // [B, n_max, words]
    int words;
    int32_t* out_keep;     // [B, max_out]
    int32_t* out_counts;   // [B]
};

__global__ __launch_bounds__(kSortThreads) void nms_sort_kernel(NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    __shared__ float red[kSortThreads / 64];
    __shared__ int cnt_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.counts ? min(a.counts[b], a.n_max) : a.n_max;
    const float* sc = a.scores + (size_t)b * a.n_max;
    const float* bx = a.boxes + (size_t)b * a.n_max * 4;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    float mx = -INFINITY;
    int local_cnt = 0;
    for (int i = tid; i < a.n_pad; i += kSortThreads) {
        unsigned long long k = ~0ull;
        if (i < n && (!a.valid || a.valid[(size_t)b * a.n_max + i])) {
            k = ((unsigned long long)ordered_desc(sc[i]) << 32) | (unsigned)i;
            ++local_cnt;
            mx = fmaxf(mx, fmaxf(fmaxf(bx[i * 4], bx[i * 4 + 1]), fmaxf(bx[i * 4 + 2], bx[i * 4 + 3])));
        }
        keys[i] = k;
    }
    atomicAdd(&cnt_s, local_cnt);
    // block max of coordinates (for the coordinate trick)
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < kSortThreads / 64; ++w) mx = fmaxf(mx, red[w]);
    // bitonic sort ascending
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
        const int i = (int)(keys[p] & 0xFFFFFFFFu);
        const int c = a.idxs ? a.idxs[(size_t)b * a.n_max + i] : 0;
        const float off = a.mode == 0 ? (float)c * off1 : 0.f;
        float* o = a.sboxes + ((size_t)b * a.n_max + p) * 4;
        o[0] = bx[i * 4] + off; o[1] = bx[i * 4 + 1] + off; o[2] = bx[i * 4 + 2] + off; o[3] = bx[i * 4 + 3] + off;
        a.sidx[(size_t)b * a.n_max + p] = i;
        a.scls[(size_t)b * a.n_max + p] = c;
    }
}

// grid (col_tiles, row_tiles, B), block 64: thread t handles sorted row (row_tile*64 + t) against 64 columns.
__global__ __launch_bounds__(64) void nms_mask_kernel(NmsArgs a) {
    const int b = blockIdx.z;
    const int nv = a.nvalid[b];
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (row0 >= nv || col0 >= nv || col0 + 63 < row0) return;  // only the upper triangle matters
    __shared__ float cb[64][4];
    __shared__ int cc[64];
    const int t = threadIdx.x;
    const float* sb = a.sboxes + (size_t)b * a.n_max * 4;
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
        if (a.mode == 1 && cc[j] != rc) continue;
        const float w = fmaxf(0.f, fminf(x2, cb[j][2]) - fmaxf(x1, cb[j][0]));
        const float h = fmaxf(0.f, fminf(y2, cb[j][3]) - fmaxf(y1, cb[j][1]));
        const float inter = w * h;
        const float aj = (cb[j][2] - cb[j][0]) * (cb[j][3] - cb[j][1]);
        const float iou = inter / (ar + aj - inter);
        if (iou > a.thr) bits |= 1ull << j;
    }
    a.mask[((size_t)b * a.n_max + r) * a.words + blockIdx.x] = bits;
}

// One wavefront per image, 64 sorted boxes per step ("pull" form):
//   1. the removed word of chunk c = OR over ALL boxes kept so far of their mask word c: the kept list lives
//      in LDS, lanes stride over it with independent loads (one memory latency per chunk), then a wave OR;
//   2. lane l fetches the DIAGONAL word of row 64c+l and the chunk is resolved in registers with scalar
//      readlanes (64 short steps, no memory);
//   3. survivors are appended to the kept list / output in order; early exit at max_out.
__global__ __launch_bounds__(64) void nms_scan_kernel(NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short* kept_pos = reinterpret_cast<unsigned short*>(smem);  // sorted positions of kept boxes
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nv = a.nvalid[b];
    const int words = (nv + 63) / 64;
    int kept = 0;
    const unsigned long long* mk = a.mask + (size_t)b * a.n_max * a.words;
    for (int c = 0; c < words && kept < a.max_out; ++c) {
        const int i0 = c * 64;
        // ---- 1. pull: who among the kept boxes suppresses members of this chunk ----
        unsigned long long rem = 0;
        for (int k = lane; k < kept; k += 64) {
            const int p = kept_pos[k];
            if ((p >> 6) < c) rem |= mk[(size_t)p * a.words + c];  // rows of earlier chunks only (same chunk: step 2)
        }
        for (int o = 32; o > 0; o >>= 1) rem |= __shfl_xor(rem, o);
        // ---- 2. resolve the chunk in registers ----
        const int row = i0 + lane;
        const unsigned long long diag = row < nv ? mk[(size_t)row * a.words + c] : 0ull;
        const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
        const int nb = min(64, nv - i0);
        unsigned long long keepmask = 0;
        int kept_here = 0;
        for (int bit = 0; bit < nb; ++bit) {
            if ((rem >> bit) & 1ull) continue;          // wave-uniform
            if (kept + kept_here >= a.max_out) break;
            keepmask |= 1ull << bit;
            ++kept_here;
            const unsigned lo = __builtin_amdgcn_readlane(dlo, bit), hi = __builtin_amdgcn_readlane(dhi, bit);
            rem |= ((unsigned long long)hi << 32) | lo;
        }
        // ---- 3. emit the survivors in order ----
        if ((keepmask >> lane) & 1ull) {
            const int slot = kept + __popcll(keepmask & pe::lanemask_lt());
            kept_pos[slot] = (unsigned short)row;
            a.out_keep[(size_t)b * a.max_out + slot] = a.sidx[(size_t)b * a.n_max + row];
        }
        kept += kept_here;
        __syncthreads();
    }
    if (lane == 0) a.out_counts[b] = kept;
}

}  // namespace

extern "C" size_t pe_nms_scratch_bytes(int32_t B, int32_t n_max) {
    const size_t words = ((size_t)n_max + 63) / 64;
    size_t s = 0;
    s += (size_t)B * n_max * 4 * sizeof(float);    // sboxes
    s += (size_t)B * n_max * sizeof(int32_t) * 2;  // sidx, scls
    s += (size_t)B * sizeof(int32_t) + 64;         // nvalid
    s += (size_t)B * n_max * words * 8;            // mask
    return s + 256;
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
    a.out_keep = out_keep; a.out_counts = out_counts;
    const size_t lds = (size_t)a.n_pad * 8;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(kSortThreads), lds, st, a);
    PE_CHECK_LAUNCH("pe_nms_batched(sort)");
    hipLaunchKernelGGL(nms_mask_kernel, dim3(a.words, a.words, B), dim3(64), 0, st, a);
    PE_CHECK_LAUNCH("pe_nms_batched(mask)");
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), (size_t)std::min(max_out, n_max) * 2 + 16, st, a);
    PE_CHECK_LAUNCH("pe_nms_batched(scan)");
    return PE_OK;
}
