// Box-head post-processing for gfx950: softmax + per-class box decode + validity + clip + score threshold
// (ordered candidate compaction), then - after the shared batched NMS - top-k gather of every output field,
// rescale to the output resolution, clip, drop-empty.  One block per image, no host round trips.
//
// Replaces FastRCNNOutputs.predict_probs / _predict_boxes / inference and fast_rcnn_inference_single_image
// (modeling/roi_heads/fast_rcnn.py:43-147,345-360,417-452) and detector_postprocess + Boxes.scale/clip/nonempty
// (modeling/postprocessing.py:8-38, structures/boxes.py:177-206,261-267), with batch-1 semantics per image
// (quirk Q2 fixed for batching) and the reference's index quirks kept:
//   Q4: class_logits / variance are NOT filtered by the finite mask, but are indexed with post-filter row ids;
//   Q3: result.vars = variance[keep]  (keep indexes the CANDIDATE list)  unless fix_vars != 0.
//
// Head tensor: fp32 [N*per_image, head_stride]: columns [0,K] class logits, [K+1, 5K] deltas (class-major),
// column 5K+1 = var_pred (log-variance; exp applied here: fast_rcnn.py:541).
#include "common.h"

namespace {
constexpr int kThreads = 1024;
constexpr int kMaxK = 80;

struct BoxHeadArgs {
    const float* head;
    int head_stride, N, per_image, K;
    const int32_t* prop_counts;  // [N]
    const float* proposals;      // [N, per_image, 4]
    const int32_t* image_hw;     // [N,2] resized (h,w)
    float wx, wy, ww, wh, scale_clamp, score_thresh;
    int cand_max;
    // candidate outputs
    float* cand_boxes;     // [N, cand_max, 4]
    float* cand_scores;    // [N, cand_max]
    int32_t* cand_class;   // [N, cand_max]
    int32_t* cand_rows;    // [N, cand_max, 2] (filtered row id, original row id)
    int32_t* cand_counts;  // [N]
    int32_t* cand_total;   // [N] optional: candidates BEFORE the cand_max cap (overflow = cand_total > cand_max)
    float* probs;          // [N, per_image, K+1] scratch
};

__device__ __forceinline__ void decode(const float* d, float bx1, float by1, float bx2, float by2, const BoxHeadArgs& a,
                                       float* o) {
    const float wd = bx2 - bx1, ht = by2 - by1;
    const float cx = bx1 + 0.5f * wd, cy = by1 + 0.5f * ht;
    const float dx = d[0] / a.wx, dy = d[1] / a.wy;
    const float dw = fminf(d[2] / a.ww, a.scale_clamp), dh = fminf(d[3] / a.wh, a.scale_clamp);
    const float pcx = dx * wd + cx, pcy = dy * ht + cy;
    const float pw = expf(dw) * wd, ph = expf(dh) * ht;
    o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
}

// exclusive block scan of one int per thread (1024 threads)
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_tot, int& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < kThreads / 64; ++i) {
        if (i < w) base += wave_tot[i];
        tot += wave_tot[i];
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(kThreads) void boxhead_candidates_kernel(BoxHeadArgs a) {
    __shared__ int wave_tot[kThreads / 64];
    const int n = blockIdx.x, r = threadIdx.x;
    const int R = min(a.prop_counts ? a.prop_counts[n] : a.per_image, a.per_image);
    const int K = a.K;
    const float ih = (float)a.image_hw[n * 2], iw = (float)a.image_hw[n * 2 + 1];
    const float* row = a.head + ((size_t)n * a.per_image + r) * a.head_stride;
    float* pr = a.probs + ((size_t)n * a.per_image + r) * (K + 1);
    bool valid = false;
    int npass = 0;
    float bx1 = 0, by1 = 0, bx2 = 0, by2 = 0;
    if (r < R) {
        float mx = -INFINITY;
        for (int k = 0; k <= K; ++k) mx = fmaxf(mx, row[k]);
        float sum = 0.f;
        for (int k = 0; k <= K; ++k) sum += expf(row[k] - mx);
        valid = true;
        for (int k = 0; k <= K; ++k) {
            const float p = expf(row[k] - mx) / sum;
            pr[k] = p;
            valid = valid && isfinite(p);
        }
        const float* pb = a.proposals + ((size_t)n * a.per_image + r) * 4;
        bx1 = pb[0]; by1 = pb[1]; bx2 = pb[2]; by2 = pb[3];
        for (int k = 0; k < K; ++k) {
            float o[4];
            decode(row + K + 1 + 4 * k, bx1, by1, bx2, by2, a, o);
            valid = valid && isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(o[3]);
        }
        if (valid)
            for (int k = 0; k < K; ++k) npass += pr[k] > a.score_thresh ? 1 : 0;
    }
    int tot_valid, tot_cand;
    const int frow = block_exclusive_scan(valid ? 1 : 0, wave_tot, tot_valid);
    int cbase = block_exclusive_scan(npass, wave_tot, tot_cand);
    if (threadIdx.x == 0) {
        a.cand_counts[n] = min(tot_cand, a.cand_max);
        if (a.cand_total) a.cand_total[n] = tot_cand;
    }
    if (valid) {
        for (int k = 0; k < K; ++k) {
            if (!(pr[k] > a.score_thresh)) continue;
            if (cbase < a.cand_max) {
                float o[4];
                decode(row + K + 1 + 4 * k, bx1, by1, bx2, by2, a, o);
                const size_t c = (size_t)n * a.cand_max + cbase;
                a.cand_boxes[c * 4] = fminf(fmaxf(o[0], 0.f), iw);
                a.cand_boxes[c * 4 + 1] = fminf(fmaxf(o[1], 0.f), ih);
                a.cand_boxes[c * 4 + 2] = fminf(fmaxf(o[2], 0.f), iw);
                a.cand_boxes[c * 4 + 3] = fminf(fmaxf(o[3], 0.f), ih);
                a.cand_scores[c] = pr[k];
                a.cand_class[c] = k;
                a.cand_rows[c * 2] = frow;
                a.cand_rows[c * 2 + 1] = r;
            }
            ++cbase;
        }
    }
}

struct FinalArgs {
    const float* head;
    int head_stride, N, per_image, K, cand_max, max_det, fix_vars;
    const float* probs;
    const float* cand_boxes;
    const float* cand_scores;
    const int32_t* cand_class;
    const int32_t* cand_rows;
    const int32_t* keep;         // [N, max_det] candidate ids in score order
    const int32_t* keep_counts;  // [N]
    const int32_t* image_hw;     // [N,2] resized (h,w)
    const int32_t* out_hw;       // [N,2] output (h,w)
    float* det_boxes;     // [N, max_det, 4]
    float* det_scores;    // [N, max_det]
    int32_t* det_classes; // [N, max_det]
    float* det_logits;    // [N, max_det, K+1]
    float* det_probs;     // [N, max_det, K]
    float* det_vars;      // [N, max_det]
    int32_t* det_rows;    // [N, max_det] original proposal row of each detection
    int32_t* det_counts;  // [N]
};

__global__ __launch_bounds__(64) void boxhead_finalize_kernel(FinalArgs a) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const int kc = min(a.keep_counts[n], a.max_det);
    const float ih = (float)a.image_hw[n * 2], iw = (float)a.image_hw[n * 2 + 1];
    const float oh = (float)a.out_hw[n * 2], ow = (float)a.out_hw[n * 2 + 1];
    const float sx = (float)((double)ow / (double)iw), sy = (float)((double)oh / (double)ih);
    int written = 0;
    for (int base = 0; base < kc; base += 64) {
        const int j = base + lane;
        bool ok = false;
        float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        int c = 0;
        if (j < kc) {
            c = a.keep[(size_t)n * a.max_det + j];
            const float* b = a.cand_boxes + ((size_t)n * a.cand_max + c) * 4;
            x1 = fminf(fmaxf(b[0] * sx, 0.f), ow); y1 = fminf(fmaxf(b[1] * sy, 0.f), oh);
            x2 = fminf(fmaxf(b[2] * sx, 0.f), ow); y2 = fminf(fmaxf(b[3] * sy, 0.f), oh);
            ok = (x2 - x1) > 0.f && (y2 - y1) > 0.f;
        }
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const size_t o = (size_t)n * a.max_det + written + __popcll(m & pe::lanemask_lt());
            const size_t cc = (size_t)n * a.cand_max + c;
            const int frow = a.cand_rows[cc * 2], orow = a.cand_rows[cc * 2 + 1];
            a.det_boxes[o * 4] = x1; a.det_boxes[o * 4 + 1] = y1; a.det_boxes[o * 4 + 2] = x2; a.det_boxes[o * 4 + 3] = y2;
            a.det_scores[o] = a.cand_scores[cc];
            a.det_classes[o] = a.cand_class[cc];
            a.det_rows[o] = orow;
            const float* lrow = a.head + ((size_t)n * a.per_image + frow) * a.head_stride;  // Q4 index
            for (int k = 0; k <= a.K; ++k) a.det_logits[o * (a.K + 1) + k] = lrow[k];
            const float* prow = a.probs + ((size_t)n * a.per_image + orow) * (a.K + 1);
            for (int k = 0; k < a.K; ++k) a.det_probs[o * a.K + k] = prow[k];
            const int vrow = a.fix_vars ? orow : min(c, a.per_image - 1);                    // Q3 index
            a.det_vars[o] = expf(a.head[((size_t)n * a.per_image + vrow) * a.head_stride + 5 * a.K + 1]);
        }
        written += __popcll(m);
    }
    if (lane == 0) a.det_counts[n] = written;
}
}  // namespace

extern "C" int pe_boxhead_candidates(const float* head, int32_t head_stride, int32_t N, int32_t per_image,
                                     int32_t num_classes, const int32_t* prop_counts, const float* proposals,
                                     const int32_t* image_hw, const float* reg_weights_host, float scale_clamp,
                                     float score_thresh, int32_t cand_max, float* cand_boxes, float* cand_scores,
                                     int32_t* cand_class, int32_t* cand_rows, int32_t* cand_counts, int32_t* cand_total,
                                     float* probs, void* stream) {
    PE_CHECK_ARG(head && proposals && image_hw && reg_weights_host, "pe_boxhead_candidates: null pointer");
    PE_CHECK_ARG(cand_boxes && cand_scores && cand_class && cand_rows && cand_counts && probs,
                 "pe_boxhead_candidates: null output");
    PE_CHECK_ARG(num_classes >= 1 && num_classes <= kMaxK, "pe_boxhead_candidates: num_classes %d", num_classes);
    PE_CHECK_ARG(per_image >= 1 && per_image <= kThreads, "pe_boxhead_candidates: per_image %d not in [1,%d]",
                 per_image, kThreads);
    PE_CHECK_ARG(head_stride >= 5 * num_classes + 2, "pe_boxhead_candidates: head_stride %d < %d", head_stride,
                 5 * num_classes + 2);
    if (N == 0) return PE_OK;
    BoxHeadArgs a{};
    a.head = head; a.head_stride = head_stride; a.N = N; a.per_image = per_image; a.K = num_classes;
    a.prop_counts = prop_counts; a.proposals = proposals; a.image_hw = image_hw;
    a.wx = reg_weights_host[0]; a.wy = reg_weights_host[1]; a.ww = reg_weights_host[2]; a.wh = reg_weights_host[3];
    a.scale_clamp = scale_clamp; a.score_thresh = score_thresh; a.cand_max = cand_max;
    a.cand_boxes = cand_boxes; a.cand_scores = cand_scores; a.cand_class = cand_class; a.cand_rows = cand_rows;
    a.cand_counts = cand_counts; a.cand_total = cand_total; a.probs = probs;
    hipLaunchKernelGGL(boxhead_candidates_kernel, dim3(N), dim3(kThreads), 0, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_boxhead_candidates");
    return PE_OK;
}

extern "C" int pe_boxhead_finalize(const float* head, int32_t head_stride, int32_t N, int32_t per_image,
                                   int32_t num_classes, int32_t cand_max, int32_t max_det, int32_t fix_vars,
                                   const float* probs, const float* cand_boxes, const float* cand_scores,
                                   const int32_t* cand_class, const int32_t* cand_rows, const int32_t* keep,
                                   const int32_t* keep_counts, const int32_t* image_hw, const int32_t* out_hw,
                                   float* det_boxes, float* det_scores, int32_t* det_classes, float* det_logits,
                                   float* det_probs, float* det_vars, int32_t* det_rows, int32_t* det_counts,
                                   void* stream) {
    PE_CHECK_ARG(head && probs && cand_boxes && cand_scores && cand_class && cand_rows && keep && keep_counts &&
                     image_hw && out_hw, "pe_boxhead_finalize: null input");
    PE_CHECK_ARG(det_boxes && det_scores && det_classes && det_logits && det_probs && det_vars && det_rows && det_counts,
                 "pe_boxhead_finalize: null output");
    if (N == 0) return PE_OK;
    FinalArgs a{};
    a.head = head; a.head_stride = head_stride; a.N = N; a.per_image = per_image; a.K = num_classes;
    a.cand_max = cand_max; a.max_det = max_det; a.fix_vars = fix_vars; a.probs = probs;
    a.cand_boxes = cand_boxes; a.cand_scores = cand_scores; a.cand_class = cand_class; a.cand_rows = cand_rows;
    a.keep = keep; a.keep_counts = keep_counts; a.image_hw = image_hw; a.out_hw = out_hw;
    a.det_boxes = det_boxes; a.det_scores = det_scores; a.det_classes = det_classes; a.det_logits = det_logits;
    a.det_probs = det_probs; a.det_vars = det_vars; a.det_rows = det_rows; a.det_counts = det_counts;
    hipLaunchKernelGGL(boxhead_finalize_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_boxhead_finalize");
    return PE_OK;
}
