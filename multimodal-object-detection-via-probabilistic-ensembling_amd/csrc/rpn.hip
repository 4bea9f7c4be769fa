// RPN proposal selection for gfx950: per (image, level) radix-select top-k of the objectness logits,
// decode ONLY the survivors against analytically generated anchors, validity / clip / non-empty flags.
//
// Replaces DefaultAnchorGenerator.forward (modeling/anchor_generator.py:130-199), RPNOutputs.predict_proposals
// / predict_objectness_logits (proposal_generator/rpn_outputs.py:409-452, decoding all 204 624 anchors, twice),
// Box2BoxTransform.apply_deltas (box_regression.py:73-110, weights 1,1,1,1) and the per-level full sort +
// top-k + finite / clip / nonempty part of find_top_rpn_proposals (rpn_outputs.py:100-145).
// Order rule: logit descending, ties by anchor index (h, w, a) ascending (stable descending sort).
//
// Input per level: the fused RPN head output, fp32 [N*H*W, 16]: columns 0..2 objectness (a = 0..2),
// columns 3..14 deltas (a*4 + {dx,dy,dw,dh}).  One 1024-thread block per (image, level).
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxTopk = 1024;
constexpr int kA = 3;
constexpr int kSliceKeys = 16383;  // anchors per stage-1 slice: 5461 whole cells, 64 KiB of keys in LDS

struct RpnLevel {
    const float* head;  // [N*H*W, head_stride]
    int H, W, stride;
    int first_slice, num_slices;  // stage-1 slices of this level (num_slices <= 1: single-stage)
    int topk;           // min(pre_nms_topk, H*W*A)
    int cand_offset;    // offset of this level inside an image's candidate list
    float cell[kA][4];  // cell anchors (float32 of the float64 closed form)
};

struct RpnArgs {
    RpnLevel lv[8];
    int num_levels, N, head_stride;
    const int32_t* image_hw;  // [N,2] unpadded (h, w) used for clipping (Q9)
    int cand_per_image;
    float scale_clamp;
    float* cand_boxes;     // [N, cand_per_image, 4]
    float* cand_scores;    // [N, cand_per_image]
    int32_t* cand_level;   // [N, cand_per_image]
    uint8_t* cand_valid;   // [N, cand_per_image]
    // two-stage selection (optional scratch)
    unsigned long long* slice_out;  // [N, num_slices, 1024]
    int num_slices;
    int slice_level[64], slice_begin[64];
};

__device__ __forceinline__ unsigned ordered_desc(float s) {
    s += 0.0f;
    unsigned u = __float_as_uint(s);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ~u;
}

// Block-wide exact top-k (k <= 1024) of `total` 32-bit keys (smaller key = better) read through key_at(i):
// 4-pass 8-bit radix select of the k-th key, then the k winners - ties broken by LOWER index i - end up in
// cand[0..k) as (key << 32 | i), sorted ascending.  cand[k..1024) is padded with ~0.
template <typename KeyAt>
__device__ __forceinline__ void block_topk(int total, int k, KeyAt key_at, unsigned long long* cand, unsigned* hist,
                                           unsigned* wave_cnt, unsigned* s4) {
    const int tid = threadIdx.x;
    unsigned &s_prefix = s4[0], &s_cnt = s4[1], &s_base = s4[2];
    unsigned prefix = 0, prefix_mask = 0, below = 0;  // `below` = #keys with key < (prefix bucket so far)
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += kThreads) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < total; i += kThreads) {
            const unsigned key = key_at(i);
            if ((key & prefix_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = below, bin = 0;
            for (; bin < 256; ++bin) {
                if (acc + hist[bin] >= (unsigned)k) break;
                acc += hist[bin];
            }
            s_prefix = prefix | (bin << shift);
            s_cnt = acc;
        }
        __syncthreads();
        prefix = s_prefix;
        below = s_cnt;
        prefix_mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned T = prefix;               // k-th smallest key
    const unsigned need_eq = k - below;      // how many keys == T to take (lowest indices first)
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i = tid; i < total; i += kThreads) {  // keys < T, any order (sorted below)
        const unsigned key = key_at(i);
        if (key < T) cand[atomicAdd(&s_cnt, 1u)] = ((unsigned long long)key << 32) | (unsigned)i;
    }
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < total; c0 += kThreads) {  // keys == T in index order: ordered block compaction
        if (s_base >= need_eq) break;  // block-uniform (read after the barrier at the loop end)
        const int i = c0 + tid;
        const bool eq = i < total && key_at(i) == T;
        const unsigned long long m = __ballot(eq);
        if ((tid & 63) == 0) wave_cnt[tid >> 6] = __popcll(m);
        __syncthreads();
        unsigned before = s_base;
        for (int w = 0; w < (tid >> 6); ++w) before += wave_cnt[w];
        const unsigned my = before + __popcll(m & pe::lanemask_lt());
        if (eq && my < need_eq) cand[below + my] = ((unsigned long long)T << 32) | (unsigned)i;
        __syncthreads();
        if (tid == 0) {
            unsigned tot = 0;
            for (int w = 0; w < kThreads / 64; ++w) tot += wave_cnt[w];
            s_base += tot;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < kMaxTopk; i += kThreads)
        if (i >= k) cand[i] = ~0ull;
    __syncthreads();
    for (int kk = 2; kk <= kMaxTopk; kk <<= 1) {  // bitonic sort of the 1024 slots
        for (int j = kk >> 1; j > 0; j >>= 1) {
            const int i = tid, ixj = i ^ j;
            if (ixj > i) {
                const unsigned long long x = cand[i], y = cand[ixj];
                const bool up = (i & kk) == 0;
                if ((x > y) == up) { cand[i] = y; cand[ixj] = x; }
            }
            __syncthreads();
        }
    }
}

// The objectness keys of a slice / small level are read from the head rows ONCE (one float4 = the three logits of a
// cell per thread) into LDS; the four radix passes and the two collection passes of block_topk then run on LDS instead of
// going back to the 64-byte head rows six times.
__device__ __forceinline__ void stage_keys(const float* head, int head_stride, int cell0, int cells, unsigned* keys) {
    for (int c = threadIdx.x; c < cells; c += kThreads) {
        const float4 v = *reinterpret_cast<const float4*>(head + (size_t)(cell0 + c) * head_stride);   // rows are 16-byte aligned (stride % 4 == 0)
        keys[c * kA] = ordered_desc(v.x);
        keys[c * kA + 1] = ordered_desc(v.y);
        keys[c * kA + 2] = ordered_desc(v.z);
    }
    __syncthreads();
}

// Stage 1 (optional, for big levels): every slice of kSliceKeys anchors finds ITS top-k on its own workgroup, so the
// 153 600-key p2 level is spread over 10 CUs instead of being one workgroup's 5 passes.  Any global top-k element
// is in its slice's top-k, so stage 2 (below) stays exact.  Output: (key << 32 | global anchor index) per slice.
__global__ __launch_bounds__(kThreads) void rpn_slice_kernel(RpnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned keys[];   // kSliceKeys
    __shared__ unsigned hist[256];
    __shared__ unsigned long long cand[kMaxTopk];
    __shared__ unsigned s4[4];
    __shared__ unsigned wave_cnt[kThreads / 64];
    const int sl = blockIdx.x, n = blockIdx.y;
    const int L = a.slice_level[sl];
    const RpnLevel& lv = a.lv[L];
    const int total_l = lv.H * lv.W * kA;
    const int s0 = a.slice_begin[sl];                  // slices start at cell boundaries (kSliceKeys % kA == 0)
    const int total = min(kSliceKeys, total_l - s0);
    const int k = min(lv.topk, total);
    const float* head = lv.head + (size_t)n * lv.H * lv.W * a.head_stride;
    stage_keys(head, a.head_stride, s0 / kA, total / kA, keys);
    auto key_at = [&](int i) { return keys[i]; };
    block_topk(total, k, key_at, cand, hist, wave_cnt, s4);
    unsigned long long* out = a.slice_out + ((size_t)n * a.num_slices + sl) * kMaxTopk;
    const int tid = threadIdx.x;
    out[tid] = tid < k ? ((cand[tid] & 0xFFFFFFFF00000000ull) | (unsigned)(s0 + (int)(cand[tid] & 0xFFFFFFFFu))) : ~0ull;
}

__global__ __launch_bounds__(kThreads) void rpn_select_kernel(RpnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned keys[];   // kSliceKeys
    __shared__ unsigned hist[256];
    __shared__ unsigned long long cand[kMaxTopk];
    __shared__ unsigned s4[4];
    __shared__ unsigned wave_cnt[kThreads / 64];
    const int tid = threadIdx.x;
    const int L = blockIdx.x, n = blockIdx.y;
    const RpnLevel& lv = a.lv[L];
    const int k = lv.topk;
    const float* head = lv.head + (size_t)n * lv.H * lv.W * a.head_stride;
    if (a.slice_out && lv.num_slices > 1) {
        // stage 2: exact top-k of the slices' winners.  The compact list is ordered (slice, rank): among equal keys a
        // lower compact index is a lower anchor index, so the tie rule carries over.
        const unsigned long long* lst = a.slice_out + ((size_t)n * a.num_slices + lv.first_slice) * kMaxTopk;
        const int total = lv.num_slices * kMaxTopk;
        auto key_at = [&](int i) { return (unsigned)(lst[i] >> 32); };  // padding slots carry 0xFFFFFFFF: never selected
        block_topk(total, k, key_at, cand, hist, wave_cnt, s4);
        if (tid < k) cand[tid] = lst[(int)(cand[tid] & 0xFFFFFFFFu)];
        __syncthreads();
    } else {
        const int total = lv.H * lv.W * kA;
        if (total <= kSliceKeys) {
            stage_keys(head, a.head_stride, 0, lv.H * lv.W, keys);
            auto key_at = [&](int i) { return keys[i]; };
            block_topk(total, k, key_at, cand, hist, wave_cnt, s4);
        } else {   // no scratch for the two-stage route: the whole level from memory
            auto key_at = [&](int i) { return ordered_desc(head[(size_t)(i / kA) * a.head_stride + (i % kA)]); };
            block_topk(total, k, key_at, cand, hist, wave_cnt, s4);
        }
    }
    // ---- decode survivors ----
    if (tid < k) {
        const int i = (int)(cand[tid] & 0xFFFFFFFFu);
        const int cell = i / kA, an = i - cell * kA;
        const int h = cell / lv.W, w = cell - h * lv.W;
        const float* row = head + (size_t)cell * a.head_stride;
        const float score = row[an];
        const float sx = (float)(w * lv.stride), sy = (float)(h * lv.stride);
        const float ax1 = sx + lv.cell[an][0], ay1 = sy + lv.cell[an][1];
        const float ax2 = sx + lv.cell[an][2], ay2 = sy + lv.cell[an][3];
        const float wd = ax2 - ax1, ht = ay2 - ay1;
        const float cx = ax1 + 0.5f * wd, cy = ay1 + 0.5f * ht;
        const float dx = row[3 + an * 4], dy = row[3 + an * 4 + 1];
        const float dw = fminf(row[3 + an * 4 + 2], a.scale_clamp), dh = fminf(row[3 + an * 4 + 3], a.scale_clamp);
        const float pcx = dx * wd + cx, pcy = dy * ht + cy;
        const float pw = expf(dw) * wd, ph = expf(dh) * ht;
        float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
        bool ok = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(score);
        const float ih = (float)a.image_hw[n * 2], iw = (float)a.image_hw[n * 2 + 1];
        x1 = fminf(fmaxf(x1, 0.f), iw); y1 = fminf(fmaxf(y1, 0.f), ih);
        x2 = fminf(fmaxf(x2, 0.f), iw); y2 = fminf(fmaxf(y2, 0.f), ih);
        ok = ok && (x2 - x1) > 0.f && (y2 - y1) > 0.f;
        const size_t o = (size_t)n * a.cand_per_image + lv.cand_offset + tid;
        a.cand_boxes[o * 4] = x1; a.cand_boxes[o * 4 + 1] = y1; a.cand_boxes[o * 4 + 2] = x2; a.cand_boxes[o * 4 + 3] = y2;
        a.cand_scores[o] = score;
        a.cand_level[o] = L;
        a.cand_valid[o] = ok ? 1 : 0;
    }
}

__global__ void gather_rows_kernel(const float* boxes, const float* scores, const int32_t* keep, const int32_t* counts,
                                   int N, int n_in, int max_out, float* out_boxes, float* out_scores) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * max_out) return;
    const int n = i / max_out, p = i - n * max_out;
    float b0 = 0, b1 = 0, b2 = 0, b3 = 0, s = 0;
    if (p < counts[n]) {
        const int r = keep[i];
        const float* b = boxes + ((size_t)n * n_in + r) * 4;
        b0 = b[0]; b1 = b[1]; b2 = b[2]; b3 = b[3];
        s = scores[(size_t)n * n_in + r];
    }
    out_boxes[(size_t)i * 4] = b0; out_boxes[(size_t)i * 4 + 1] = b1;
    out_boxes[(size_t)i * 4 + 2] = b2; out_boxes[(size_t)i * 4 + 3] = b3;
    if (out_scores) out_scores[i] = s;
}

}  // namespace

extern "C" int pe_rpn_select_topk(const float* const* level_heads_host, const int32_t* level_hw_host,
                                  const int32_t* level_stride_host, const float* cell_anchors_host,
                                  int32_t num_levels, int32_t N, int32_t head_stride, int32_t pre_nms_topk,
                                  const int32_t* image_hw, float scale_clamp, float* cand_boxes, float* cand_scores,
                                  int32_t* cand_level, uint8_t* cand_valid, int32_t cand_per_image, void* scratch,
                                  size_t scratch_bytes, void* stream) {
    PE_CHECK_ARG(num_levels >= 1 && num_levels <= 8, "pe_rpn_select_topk: num_levels %d", num_levels);
    PE_CHECK_ARG(pre_nms_topk >= 1 && pre_nms_topk <= kMaxTopk, "pe_rpn_select_topk: pre_nms_topk %d not in [1,%d]",
                 pre_nms_topk, kMaxTopk);
    PE_CHECK_ARG(head_stride >= 16 && head_stride % 4 == 0, "pe_rpn_select_topk: head_stride %d (rows must be 16-byte aligned, >= 16 floats)", head_stride);
    PE_CHECK_ARG(level_heads_host && level_hw_host && level_stride_host && cell_anchors_host && image_hw,
                 "pe_rpn_select_topk: null pointer");
    PE_CHECK_ARG(cand_boxes && cand_scores && cand_level && cand_valid, "pe_rpn_select_topk: null output");
    if (N == 0) return PE_OK;
    RpnArgs a{};
    int off = 0;
    for (int l = 0; l < num_levels; ++l) {
        RpnLevel& lv = a.lv[l];
        lv.head = level_heads_host[l];
        lv.H = level_hw_host[2 * l]; lv.W = level_hw_host[2 * l + 1]; lv.stride = level_stride_host[l];
        const long long tot = (long long)lv.H * lv.W * kA;
        lv.topk = (int)std::min<long long>(pre_nms_topk, tot);
        lv.cand_offset = off;
        off += lv.topk;
        for (int i = 0; i < kA * 4; ++i) lv.cell[i / 4][i % 4] = cell_anchors_host[l * kA * 4 + i];
        PE_CHECK_ARG(lv.head != nullptr, "pe_rpn_select_topk: null level pointer");
    }
    PE_CHECK_ARG(off == cand_per_image, "pe_rpn_select_topk: cand_per_image %d != sum of per-level top-k %d",
                 cand_per_image, off);
    a.num_levels = num_levels; a.N = N; a.head_stride = head_stride; a.image_hw = image_hw;
    a.cand_per_image = cand_per_image; a.scale_clamp = scale_clamp;
    a.cand_boxes = cand_boxes; a.cand_scores = cand_scores; a.cand_level = cand_level; a.cand_valid = cand_valid;
    // two-stage selection for levels with more than one slice, when the caller provides scratch
    int ns = 0;
    for (int l = 0; l < num_levels; ++l) {
        const int tot = a.lv[l].H * a.lv[l].W * kA;
        const int cnt = (tot + kSliceKeys - 1) / kSliceKeys;
        a.lv[l].first_slice = ns;
        a.lv[l].num_slices = cnt > 1 ? cnt : 0;
        if (cnt > 1) {
            for (int c = 0; c < cnt && ns < 64; ++c, ++ns) { a.slice_level[ns] = l; a.slice_begin[ns] = c * kSliceKeys; }
        }
    }
    a.num_slices = ns;
    static_assert(kSliceKeys % kA == 0, "slices are cut at cell boundaries");
    // Dynamic LDS = the staged objectness keys, sized for what THIS call stages (a stage-2 merge and the from-memory route of
    // a big level without scratch stage nothing): small pyramids keep their launches under the 64 KiB default.
    auto key_bytes = [](int keys) { return ((size_t)keys * sizeof(unsigned) + 15) / 16 * 16; };
    const size_t need = (size_t)N * ns * kMaxTopk * sizeof(unsigned long long);
    const bool two_stage = scratch && ns > 0 && ns < 64 && scratch_bytes >= need;
    size_t lds_slice = 0, lds_select = 0;
    for (int l = 0; l < num_levels; ++l) {
        const int tot = a.lv[l].H * a.lv[l].W * kA;
        if (two_stage && a.lv[l].num_slices > 1) lds_slice = std::max(lds_slice, key_bytes(std::min(tot, kSliceKeys)));
        else if (tot <= kSliceKeys) lds_select = std::max(lds_select, key_bytes(tot));
    }
    constexpr size_t kStaticLds = 10 * 1024;   // hist + cand + counters of either kernel (9.3 KiB), rounded up
    if (two_stage) {
        a.slice_out = (unsigned long long*)scratch;
        PE_ENSURE_LDS(rpn_slice_kernel, lds_slice + kStaticLds, "pe_rpn_select_topk(slices)");
        hipLaunchKernelGGL(rpn_slice_kernel, dim3(ns, N), dim3(kThreads), lds_slice, (hipStream_t)stream, a);
        PE_CHECK_LAUNCH("pe_rpn_select_topk(slices)");
    } else {
        a.slice_out = nullptr;
    }
    PE_ENSURE_LDS(rpn_select_kernel, lds_select + kStaticLds, "pe_rpn_select_topk");
    hipLaunchKernelGGL(rpn_select_kernel, dim3(num_levels, N), dim3(kThreads), lds_select, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_rpn_select_topk");
    return PE_OK;
}

extern "C" size_t pe_rpn_scratch_bytes(const int32_t* level_hw_host, int32_t num_levels, int32_t N) {
    size_t ns = 0;
    for (int l = 0; l < num_levels; ++l) {
        const long long tot = (long long)level_hw_host[2 * l] * level_hw_host[2 * l + 1] * kA;
        const long long cnt = (tot + kSliceKeys - 1) / kSliceKeys;
        if (cnt > 1) ns += (size_t)cnt;
    }
    return (size_t)N * ns * kMaxTopk * sizeof(unsigned long long);
}

extern "C" int pe_gather_boxes(const float* boxes, const float* scores, const int32_t* keep, const int32_t* counts,
                               int32_t N, int32_t n_in, int32_t max_out, float* out_boxes, float* out_scores,
                               void* stream) {
    PE_CHECK_ARG(boxes && keep && counts && out_boxes, "pe_gather_boxes: null pointer");
    if (N == 0) return PE_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(pe::ceil_div((long long)N * max_out, 256)), dim3(256), 0,
                       (hipStream_t)stream, boxes, scores, keep, counts, N, n_in, max_out, out_boxes, out_scores);
    PE_CHECK_LAUNCH("pe_gather_boxes");
    return PE_OK;
}
