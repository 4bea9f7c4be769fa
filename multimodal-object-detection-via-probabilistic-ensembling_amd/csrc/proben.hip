// ProbEn late fusion for gfx950: ONE WORKGROUP PER IMAGE (one wavefront in rounds 1-4), whole per-image state in LDS.
//
// Replaces the reference's per-image NumPy loop (demo/FLIR/demo_probEn.py:92-187 nms_bayesian,
// :32-42 bayesian_fusion_multiclass, :24-30 bayesian_fusion, :73-77 weighted_box_fusion,
// :20-22 avg_bbox_fusion).  float64 throughout (the reference's NumPy dtype); compiled with
// -ffp-contract=off so IoU decisions round exactly like the reference's separate mul/add/div.
//
// Kernel shape: the greedy clustering is sequential in the pivot (<= N pivots per image) but its IoU tests are not - they
// use the rows' own geometry -, so parallelism comes from (a) the pair tests as ballots into bit matrices by all waves and a
// register-resident walk over them (or, when the matrices do not fit, 64 lanes scoring the pivot against 64 candidates per step),
// (b) per-row ranks / logs / geometry a thread per row, (c) the fusion formulas AFTER the clustering, a thread per cluster,
// (d) one workgroup per image, any number of images per launch.  Per image the HBM
// traffic is N*(4+1+K+1)*8 + N*4 bytes in and M*(32+4+4+4) bytes out; everything else stays in LDS.
#include "common.h"

namespace {

struct ProbenArgs {
    const double* boxes;
    const double* scores;
    const double* probs;
    const double* vars;
    const int32_t* classes;
    const int32_t* offsets;
    const int32_t* row_counts;
    const int32_t* passthrough;
    int32_t B, K, max_rows, score_mode, box_mode;
    double thr, fw, fh;
    double* out_boxes;
    float* out_scores;
    float* out_classes;
    int32_t* out_keep;
    int32_t* out_counts;
};

// Sort rule shared with oracle/proben.py: NaN first, score descending, ties by ORIGINAL index
// descending (== reversed stable ascending argsort, the reference's `argsort()[::-1]`).
__device__ __forceinline__ bool precedes(double sa, int ia, double sb, int ib) {
    const bool na = sa != sa, nb = sb != sb;
    if (na != nb) return na;
    if (!na && sa != sb) return sa > sb;
    return ia > ib;
}

// One 1024-thread workgroup per image (one per CU; a step has 32 of them).  The rows' original boxes, 1 / variance and class ids
// are copied into LDS (by sorted position) next to the geometry: nothing after the second phase touches global memory (PE_SCORE_MAX
// excepted).  Phases: 1 rank sort (a thread per row), 2 geometry + logs (a thread per row), 3 clustering, 4 fusion (a thread per
// cluster).  Clustering, BITS form: (a) all 16 waves fill two bit matrices over the sorted rows, match[p][q] = IoU > thr and
// kill[p][q] = !(IoU <= thr) for q > p (a ballot per 64 candidates - the IoU tests use the rows' own geometry, never a fused box,
// so they do not depend on the order the pivots are visited in); (b) wave 0 walks the rows in order with the alive set in registers
// (lane w = rows 64w .. 64w+63): a live row becomes a pivot, its cluster = match row & alive, alive &= ~kill row - four rows'
// matrix lines are fetched per LDS round trip.  The sequential form (below, when the matrices do not fit the LDS) computes the
// IoUs inside the walk: ~1 500 cycles per row against ~100; with the rank sort on one wave that was 0.26 ms per step at the END
// of the step, where nothing overlaps it (profiles/r05_proben_phases.txt).
constexpr int kFuseThreads = 1024;

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

template <bool BITS>
__global__ __launch_bounds__(kFuseThreads) void proben_fuse_kernel(ProbenArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int ncl_s;
    const int img = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beg = a.offsets[img];
    const int n = a.row_counts ? a.row_counts[img] : a.offsets[img + 1] - beg;
    if (n > a.max_rows || n < 0) {
        if (tid == 0) a.out_counts[img] = -1;
        return;
    }
    if (a.passthrough && a.passthrough[img]) {  // exactly one detector fired: rows pass through unchanged
        for (int r = tid; r < n; r += kFuseThreads) {
            const size_t o = (size_t)beg + r;
            for (int e = 0; e < 4; ++e) a.out_boxes[o * 4 + e] = a.boxes[o * 4 + e];
            a.out_scores[o] = (float)a.scores[o];
            a.out_classes[o] = (float)a.classes[o];
            a.out_keep[o] = r;
        }
        if (tid == 0) a.out_counts[img] = n;
        return;
    }
    const int R = a.max_rows;
    const int K = a.K;
    const int W = (R + 63) >> 6;   // 64-row words per matrix line
    // columns of log-probabilities kept per row (0 when the score mode does not need them)
    const int L = (a.score_mode == PE_SCORE_PROBEN) ? K + 1 : (a.score_mode == PE_SCORE_PROBEN_BINARY ? 2 : 0);
    // ---- LDS carve (all arrays indexed by SORTED position unless noted) ----
    unsigned long long* mbits = reinterpret_cast<unsigned long long*>(smem);      // BITS: match [R][W], then the clusters' member bits
    unsigned long long* kbits = mbits + (BITS ? (size_t)R * W : 0);               // BITS: kill [R][W]
    double* gx1 = reinterpret_cast<double*>(kbits + (BITS ? (size_t)R * W : 0));
    double* gy1 = gx1 + R;
    double* gx2 = gy1 + R;
    double* gy2 = gx2 + R;
    double* gar = gy2 + R;
    double* gsc = gar + R;   // score
    double* glog = gsc + R;  // [L][R]
    double* gob = glog + (size_t)L * R;                       // [4][R] original coordinates
    double* ginv = gob + 4 * (size_t)R;                       // 1 / variance (v-avg only)
    int* ord = reinterpret_cast<int*>(ginv + R);              // sorted position -> original row
    int* gcls = ord + R;                                      // class id
    unsigned short* members = reinterpret_cast<unsigned short*>(gcls + R);   // all clusters' matches, back to back
    unsigned short* cl_piv = members + R;      // per cluster: pivot position, first member, number of matches
    unsigned short* cl_beg = cl_piv + R;
    unsigned short* cl_cnt = cl_beg + R;
    unsigned char* alive = reinterpret_cast<unsigned char*>(cl_cnt + R);       // sequential form only

    // ---- 1. rank sort by score (scores staged through gar, indexed by ORIGINAL row) ----
    for (int r = tid; r < n; r += kFuseThreads) gar[r] = a.scores[beg + r];
    __syncthreads();
    for (int r = tid; r < n; r += kFuseThreads) {
        const double s = gar[r];
        int rank = 0;
#pragma unroll 8
        for (int q = 0; q < n; ++q) rank += precedes(gar[q], q, s, r) ? 1 : 0;
        ord[rank] = r;
        gsc[rank] = s;
    }
    __syncthreads();
    // ---- 2. per-row geometry (class-band shifted, legacy "+1" area) and log-probabilities ----
    for (int p = tid; p < n; p += kFuseThreads) {
        const int r = ord[p];
        const double c = (double)a.classes[beg + r];
        const double* b = a.boxes + (size_t)(beg + r) * 4;
        const double x1 = b[0] + c * a.fw, y1 = b[1] + c * a.fh;
        const double x2 = b[2] + c * a.fw, y2 = b[3] + c * a.fh;
        gx1[p] = x1; gy1[p] = y1; gx2[p] = x2; gy2[p] = y2;
        gar[p] = (x2 - x1 + 1.0) * (y2 - y1 + 1.0);
        if (!BITS) alive[p] = 1;
        gob[p] = b[0]; gob[R + p] = b[1]; gob[2 * (size_t)R + p] = b[2]; gob[3 * (size_t)R + p] = b[3];
        gcls[p] = a.classes[beg + r];
        if (a.box_mode == PE_BOX_VAVG) ginv[p] = 1.0 / a.vars[beg + r];
        if (a.score_mode == PE_SCORE_PROBEN) {
            const double* pr = a.probs + (size_t)(beg + r) * K;
            double sum = 0.0;
            for (int j = 0; j < K; ++j) {
                const double pj = pr[j];
                sum += pj;
                glog[(size_t)j * R + p] = log(pj);
            }
            glog[(size_t)K * R + p] = log(1.0 - sum);
        } else if (a.score_mode == PE_SCORE_PROBEN_BINARY) {
            const double s = gsc[p];
            glog[p] = log(s);
            glog[(size_t)R + p] = log(1.0 - s);
        }
    }
    __syncthreads();

    // ---- 3. greedy clustering: only the membership is recorded ----
    if (BITS) {
        // (a) the pair tests: one (row, 64 candidates) item per wave step
        const int Wn = (n + 63) >> 6;
        for (int item = wave; item < n * Wn; item += kFuseThreads / 64) {
            const int p = item / Wn, c = item - p * Wn;
            unsigned long long m = 0, kl = 0;
            if (c >= (p >> 6)) {
                const int q = c * 64 + lane;
                bool match = false, kill = false;
                if (q > p && q < n) {
                    const double px1 = gx1[p], py1 = gy1[p], px2 = gx2[p], py2 = gy2[p], par = gar[p];
                    const double w = fmax(0.0, fmin(px2, gx2[q]) - fmax(px1, gx1[q]) + 1.0);
                    const double h = fmax(0.0, fmin(py2, gy2[q]) - fmax(py1, gy1[q]) + 1.0);
                    const double inter = w * h;
                    const double ovr = inter / (par + gar[q] - inter);
                    match = ovr > a.thr;
                    kill = !(ovr <= a.thr);      // matched or NaN: leaves the pool
                }
                m = __ballot(match);
                kl = __ballot(kill);
            }
            if (lane == 0) { mbits[(size_t)p * W + c] = m; kbits[(size_t)p * W + c] = kl; }
        }
        __syncthreads();
        // (b) the walk
        if (wave == 0) {
            unsigned long long live = 0;       // lane w: rows 64w .. 64w + 63
            if (lane < Wn) live = (n - lane * 64 >= 64) ? ~0ull : ((1ull << (n - lane * 64)) - 1ull);
            int ncl = 0, cursor = 0;
            for (int p0 = 0; p0 < n; p0 += 4) {
                unsigned long long mrow[4], krow[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool ok = p0 + u < n && lane < Wn;
                    mrow[u] = ok ? mbits[(size_t)(p0 + u) * W + lane] : 0ull;
                    krow[u] = ok ? kbits[(size_t)(p0 + u) * W + lane] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = p0 + u;
                    if (p >= n) break;
                    if (!((readlane64(live, p >> 6) >> (p & 63)) & 1ull)) continue;   // wave-uniform
                    const unsigned long long mem = mrow[u] & live;
                    live &= ~krow[u];
                    int cnt = 0;
                    for (int w = 0; w < Wn; ++w) cnt += __popcll(readlane64(mem, w));
                    if (lane < Wn) mbits[(size_t)p * W + lane] = mem;       // the line now holds the cluster's members
                    if (lane == 0) { cl_piv[ncl] = (unsigned short)p; cl_beg[ncl] = (unsigned short)cursor; cl_cnt[ncl] = (unsigned short)cnt; }
                    cursor += cnt;
                    ++ncl;
                }
            }
            if (lane == 0) ncl_s = ncl;
        }
    } else if (wave == 0) {
        // sequential in the pivot, 64 candidates per step, IoUs computed on the way
        int ncl = 0, cursor = 0;
        for (int pos = 0; pos < n; ++pos) {
            if (!alive[pos]) continue;  // wave-uniform (LDS broadcast)
            const double px1 = gx1[pos], py1 = gy1[pos], px2 = gx2[pos], py2 = gy2[pos], par = gar[pos];
            int cnt = 0;
            for (int base = pos + 1; base < n; base += 64) {
                const int q = base + lane;
                bool match = false;
                if (q < n && alive[q]) {
                    const double w = fmax(0.0, fmin(px2, gx2[q]) - fmax(px1, gx1[q]) + 1.0);
                    const double h = fmax(0.0, fmin(py2, gy2[q]) - fmax(py1, gy1[q]) + 1.0);
                    const double inter = w * h;
                    const double ovr = inter / (par + gar[q] - inter);
                    match = ovr > a.thr;
                    if (!(ovr <= a.thr)) alive[q] = 0;  // matched or NaN: leaves the pool
                }
                const unsigned long long mask = __ballot(match);
                if (match) members[cursor + cnt + __popcll(mask & pe::lanemask_lt())] = (unsigned short)q;
                cnt += __popcll(mask);
            }
            if (lane == 0) { cl_piv[ncl] = (unsigned short)pos; cl_beg[ncl] = (unsigned short)cursor; cl_cnt[ncl] = (unsigned short)cnt; }
            cursor += cnt;
            ++ncl;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // alive[] is re-read by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) ncl_s = ncl;
    }
    __syncthreads();
    const int ncl = ncl_s;
    // ---- 4. fusion: one lane per cluster (cluster = matches in sorted order + the pivot LAST), output row = cluster index.
    // The per-cluster arithmetic is the sequence the reference runs per pivot (sums over the members in cluster order, the
    // normaliser summed over the columns in column order, first-maximum / first-NaN rules); it used to sit inside the pivot loop
    // with 4 (+4) of the 64 lanes working and every latency of its dependent chains exposed ~100 times per image. ----
    for (int k = tid; k < ncl; k += kFuseThreads) {
        const int pos = cl_piv[k], cnt = cl_cnt[k];
        unsigned short* mem = members + cl_beg[k];
        if (BITS) {     // the member bits in ascending (= sorted) order; this thread's own stretch of the list
            int i = 0;
            for (int w = pos >> 6; w < ((n + 63) >> 6); ++w)
                for (unsigned long long bits = mbits[(size_t)pos * W + w]; bits; bits &= bits - 1ull) mem[i++] = (unsigned short)(w * 64 + __builtin_ctzll(bits));
        }
        const int m = cnt + 1;
        const int piv_row = ord[pos];
        auto at = [&](int t) { return t < cnt ? (int)mem[t] : pos; };
        auto coord = [&](int c4, int p) { return gob[(size_t)c4 * R + p]; };
        double out_score = gsc[pos];
        double out_class = (double)gcls[pos];
        double out_coord[4];
        if (cnt == 0) {
            for (int c4 = 0; c4 < 4; ++c4) out_coord[c4] = coord(c4, pos);
        } else {
            // ---------- score fusion ----------
            if (L > 0) {
                auto column = [&](int j) {       // exp of the cluster's summed log-probability of column j
                    double acc = 0.0;
                    const double* col = glog + (size_t)j * R;
                    for (int t = 0; t < m; ++t) acc += col[at(t)];
                    return exp(acc);
                };
                double tot = 0.0;
                for (int j = 0; j < L; ++j) tot += column(j);
                if (a.score_mode == PE_SCORE_PROBEN) {
                    // np.max / np.argmax over the K+1 entries INCLUDING background; NaN wins, first NaN index
                    double best = column(0) / tot;
                    int bi = 0;
                    bool bnan = best != best;
                    for (int j = 1; j < L; ++j) {
                        const double v = column(j) / tot;
                        if (!bnan && (v != v || v > best)) { best = v; bi = j; bnan = v != v; }
                    }
                    out_score = best;
                    out_class = (double)bi;
                } else {
                    out_score = column(0) / tot;
                }
            } else if (a.score_mode == PE_SCORE_AVG) {
                double acc = 0.0;
                for (int t = 0; t < m; ++t) acc += gsc[at(t)];
                out_score = acc / (double)m;
            } else {  // PE_SCORE_MAX: max over the whole [m,K] probability matrix
                double best = -INFINITY;
                bool bnan = false;
                for (int t = 0; t < m; ++t) {
                    const int r = ord[at(t)];
                    for (int j = 0; j < K; ++j) {
                        const double v = a.probs[(size_t)(beg + r) * K + j];
                        if (v != v) bnan = true;
                        best = v > best ? v : best;
                    }
                }
                out_score = bnan ? NAN : best;
            }
            // ---------- box fusion ----------
            if (a.box_mode == PE_BOX_VAVG || a.box_mode == PE_BOX_SAVG) {
                auto weight = [&](int p) { return (a.box_mode == PE_BOX_VAVG) ? ginv[p] : gsc[p]; };
                double wsum = 0.0;
                for (int t = 0; t < m; ++t) wsum += weight(at(t));
                for (int c4 = 0; c4 < 4; ++c4) {
                    double acc = 0.0;
                    for (int t = 0; t < m; ++t) {
                        const int p = at(t);
                        acc += coord(c4, p) * (weight(p) / wsum);
                    }
                    out_coord[c4] = acc;
                }
            } else if (a.box_mode == PE_BOX_AVG) {
                for (int c4 = 0; c4 < 4; ++c4) {
                    double acc = 0.0;
                    for (int t = 0; t < m; ++t) acc += coord(c4, at(t));
                    out_coord[c4] = acc / (double)m;
                }
            } else {  // argmax: box of the first maximal score in cluster order
                int bp = at(0);
                double best = gsc[bp];
                bool bnan = best != best;
                for (int t = 1; t < m; ++t) {
                    const int p = at(t);
                    const double v = gsc[p];
                    if (!bnan && (v != v || v > best)) { best = v; bp = p; bnan = v != v; }
                }
                for (int c4 = 0; c4 < 4; ++c4) out_coord[c4] = coord(c4, bp);
            }
        }
        const size_t o = (size_t)beg + k;
        a.out_scores[o] = (float)out_score;
        a.out_classes[o] = (float)out_class;
        a.out_keep[o] = piv_row;
        for (int c4 = 0; c4 < 4; ++c4) a.out_boxes[o * 4 + c4] = out_coord[c4];
    }
    if (tid == 0) a.out_counts[img] = ncl;
}

struct PackArgs {
    const float* boxes[4];
    const float* scores[4];
    const int32_t* classes[4];
    const float* probs[4];
    const float* vars[4];
    const int32_t* counts[4];
    int nd, B, D, K, max_class, stride;
    double* ob;
    double* os;
    double* op;
    double* ov;
    int32_t* oc;
    int32_t* ooff;
    int32_t* ocnt;
    int32_t* osingle;
};

// one wavefront per image: ordered compaction of every detector's live rows (class <= max_class)
__global__ __launch_bounds__(64) void proben_pack_kernel(PackArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int written = 0, sources = 0;
    for (int d = 0; d < a.nd; ++d) {
        const int before = written;
        const int c = min(a.counts[d][b], a.D);
        for (int base = 0; base < c; base += 64) {
            const int j = base + lane;
            bool ok = false;
            int cls = 0;
            if (j < c) {
                cls = a.classes[d][(size_t)b * a.D + j];
                ok = cls <= a.max_class;
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const size_t src = (size_t)b * a.D + j;
                const size_t dst = (size_t)b * a.stride + written + __popcll(m & pe::lanemask_lt());
                for (int e = 0; e < 4; ++e) a.ob[dst * 4 + e] = (double)a.boxes[d][src * 4 + e];
                a.os[dst] = (double)a.scores[d][src];
                for (int k = 0; k < a.K; ++k) a.op[dst * a.K + k] = (double)a.probs[d][src * a.K + k];
                a.ov[dst] = (double)a.vars[d][src];
                a.oc[dst] = cls;
            }
            written += __popcll(m);
        }
        sources += written > before ? 1 : 0;
    }
    if (lane == 0) {
        a.ooff[b] = b * a.stride;
        a.ocnt[b] = written;
        if (a.osingle) a.osingle[b] = sources == 1 ? 1 : 0;
    }
}

}  // namespace

extern "C" int pe_proben_pack_detections(const float* const* det_boxes_host, const float* const* det_scores_host,
                                         const int32_t* const* det_classes_host, const float* const* det_probs_host,
                                         const float* const* det_vars_host, const int32_t* const* det_counts_host,
                                         int32_t num_detectors, int32_t num_images, int32_t det_stride,
                                         int32_t num_classes, int32_t max_class, int32_t row_stride,
                                         double* out_boxes, double* out_scores, double* out_probs, double* out_vars,
                                         int32_t* out_classes, int32_t* out_offsets, int32_t* out_counts,
                                         int32_t* out_single_source, void* stream) {
    PE_CHECK_ARG(num_detectors >= 1 && num_detectors <= 4, "pe_proben_pack_detections: num_detectors %d", num_detectors);
    PE_CHECK_ARG(row_stride >= num_detectors * det_stride, "pe_proben_pack_detections: row_stride %d < %d", row_stride,
                 num_detectors * det_stride);
    PE_CHECK_ARG(out_boxes && out_scores && out_probs && out_vars && out_classes && out_offsets && out_counts,
                 "pe_proben_pack_detections: null output");
    if (num_images == 0) return PE_OK;
    PackArgs a{};
    for (int d = 0; d < num_detectors; ++d) {
        a.boxes[d] = det_boxes_host[d]; a.scores[d] = det_scores_host[d]; a.classes[d] = det_classes_host[d];
        a.probs[d] = det_probs_host[d]; a.vars[d] = det_vars_host[d]; a.counts[d] = det_counts_host[d];
        PE_CHECK_ARG(a.boxes[d] && a.scores[d] && a.classes[d] && a.probs[d] && a.vars[d] && a.counts[d],
                     "pe_proben_pack_detections: null detector pointer");
    }
    a.nd = num_detectors; a.B = num_images; a.D = det_stride; a.K = num_classes; a.max_class = max_class;
    a.stride = row_stride; a.ob = out_boxes; a.os = out_scores; a.op = out_probs; a.ov = out_vars; a.oc = out_classes;
    a.ooff = out_offsets; a.ocnt = out_counts; a.osingle = out_single_source;
    hipLaunchKernelGGL(proben_pack_kernel, dim3(num_images), dim3(64), 0, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_proben_pack_detections");
    return PE_OK;
}

extern "C" int pe_proben_fuse_batch(const double* boxes, const double* scores, const double* probs,
                                    const double* variances, const int32_t* classes, const int32_t* offsets,
                                    const int32_t* row_counts, const int32_t* passthrough, int32_t num_images, int32_t num_classes, int32_t max_rows_per_image,
                                    int32_t score_mode, int32_t box_mode, double iou_thresh, double frame_w,
                                    double frame_h, double* out_boxes, float* out_scores, float* out_classes,
                                    int32_t* out_keep, int32_t* out_counts, void* stream) {
    PE_CHECK_ARG(num_images >= 0, "pe_proben_fuse_batch: num_images < 0");
    if (num_images == 0) return PE_OK;
    PE_CHECK_ARG(boxes && scores && variances && classes && offsets, "pe_proben_fuse_batch: null input pointer");
    PE_CHECK_ARG(out_boxes && out_scores && out_classes && out_keep && out_counts,
                 "pe_proben_fuse_batch: null output pointer");
    PE_CHECK_ARG(score_mode >= 0 && score_mode <= 3, "pe_proben_fuse_batch: bad score_mode %d", score_mode);
    PE_CHECK_ARG(box_mode >= 0 && box_mode <= 3, "pe_proben_fuse_batch: bad box_mode %d", box_mode);
    // K <= 62: a wavefront keeps a cluster's per-class log-odds in lanes (K + background in 64 lanes).  Enough for every fusion the
    // reference can run: prediction files of different class counts cannot be fused there either (prepare_data concatenates the
    // `probs` arrays, demo_probEn.py:79-90), and the 80-class rgb_only file is evaluated on its own.
    PE_CHECK_ARG(num_classes >= 1 && num_classes <= 62, "pe_proben_fuse_batch: num_classes %d not in [1,62]",
                 num_classes);
    PE_CHECK_ARG(probs || (score_mode != PE_SCORE_PROBEN && score_mode != PE_SCORE_MAX),
                 "pe_proben_fuse_batch: probs required for this score_mode");
    PE_CHECK_ARG(max_rows_per_image >= 1 && max_rows_per_image <= 2048,
                 "pe_proben_fuse_batch: max_rows_per_image %d not in [1,2048]", max_rows_per_image);
    const int R = (max_rows_per_image + 1) & ~1;  // keep the int/short/byte carves 8-byte aligned
    const int L = score_mode == PE_SCORE_PROBEN ? num_classes + 1 : (score_mode == PE_SCORE_PROBEN_BINARY ? 2 : 0);
    const size_t lds_seq = (size_t)R * (8 * (6 + L + 5) + 4 + 4 + 4 * 2 + 1) + 16;
    const size_t lds_bits = lds_seq + (size_t)R * ((R + 63) / 64) * 16;        // + the two bit matrices
    constexpr size_t kStatic = 512;                                            // the kernels' static __shared__ scratch (ncl_s, reductions)
    const bool bits = lds_bits + kStatic <= 160 * 1024;
    const size_t lds = bits ? lds_bits : lds_seq;
    if (lds + kStatic > 160 * 1024) {
        // a whole image's rows live in LDS (boxes, 1 / variance, class ids, log-odds, cluster tables: 8 (11 + L) + 17 bytes per row);
        // capacity at K = 3: 1 195 rows per image (probEn), 1 400 (other score modes) - a detector contributes at most 100
        pe::set_error("pe_proben_fuse_batch: %zu bytes of LDS needed (> 160 KiB): max_rows_per_image %d is above the per-image capacity of %zu rows "
                      "for this score mode / class count", lds + kStatic, max_rows_per_image,
                      (size_t)(160 * 1024 - kStatic - 16) / (size_t)(8 * (6 + L + 5) + 4 + 4 + 4 * 2 + 1));
        return PE_ERR_UNSUPPORTED;
    }
    ProbenArgs a{boxes, scores, probs, variances, classes, offsets, row_counts, passthrough, num_images, num_classes, R,
                 score_mode, box_mode, iou_thresh, frame_w, frame_h,
                 out_boxes, out_scores, out_classes, out_keep, out_counts};
    const void* fn = bits ? reinterpret_cast<const void*>(proben_fuse_kernel<true>) : reinterpret_cast<const void*>(proben_fuse_kernel<false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            pe::set_error("pe_proben_fuse_batch: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e));
            return PE_ERR_HIP;
        }
    }
    if (bits)
        hipLaunchKernelGGL(proben_fuse_kernel<true>, dim3(num_images), dim3(kFuseThreads), lds, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(proben_fuse_kernel<false>, dim3(num_images), dim3(kFuseThreads), lds, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_proben_fuse_batch");
    return PE_OK;
}
