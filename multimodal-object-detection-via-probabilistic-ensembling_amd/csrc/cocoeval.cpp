// Native COCO bbox evaluator (host, multithreaded): evaluateImg + accumulate of the reference's vendored
// COCOeval (detectron2/pycocotools/cocoeval.py:85-191 evaluate/computeIoU, :236-313 evaluateImg, :315-421
// accumulate, Params :500-536) and pycocotools 2.0.4 `_mask.iou` -> bbIou (third-party C, not in the tree;
// call site cocoeval.py:190) for boxes.  SURVEY 8(f) item 1: the NumPy evaluator costs as much wall time as the
// whole detector + ProbEn pass over FLIR-val on one MI355X; this one takes milliseconds.
//
// The work is tiny, branchy, float64 and sequential per (image, category) - a host job, not a GPU kernel:
// (image, category) pairs are matched in parallel by a pool of std::threads, then every (category, area range,
// maxDets) cell is accumulated in parallel.  Results are bit-identical to the NumPy restatement
// (proben_amd/evaluation.py, pinned to the reference by tests/golden/cocoeval_case.json): same stable sorts,
// same float64 expressions in the same order.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <vector>

#include "proben_hip.h"

namespace pe {
void set_error(const char* fmt, ...);
}

namespace {

// "a sorts before b" in `np.argsort(-score, kind="mergesort")` (cocoeval.py:177,257,371): descending scores, NaN LAST (NumPy orders NaN
// behind every number; NaNs are equivalent to each other, so the stable sort keeps their file order).  A bare `a > b` is not a strict weak
// ordering once a NaN is in the list - std::stable_sort then returns an arbitrary order, and a NaN score is what the reference's own
// bayesian_fusion_multiclass makes of a detection whose class probabilities sum to 1 + 1 ulp (background = 1 - sum < 0, log -> NaN;
// demo_probEn.py:32-42): 15 of 7 885 fused rows on the round-6 fused-mAP fixture moved the AP by 6 points before this rule.
inline bool score_before(double a, double b) { return a > b || (a == a && b != b); }

struct PairEval {                 // one (category, image) cell; empty() when it has neither GT nor detections
    int D = 0, G = 0;
    bool present = false;
    std::vector<double> scores;               // [D] sorted by -score (stable), D <= maxDets[-1]
    std::vector<uint8_t> matched;             // [A][T][D]  dtm != 0
    std::vector<uint8_t> dt_ignore;           // [A][T][D]
    std::vector<int32_t> gt_kept;             // [A] number of non-ignored GT
};

struct Ctx {
    const double* gt_box; const double* gt_area; const uint8_t* gt_crowd; const int64_t* gt_id;
    const double* dt_box; const double* dt_score;
    const double* iou_thrs; int T;
    const double* rec_thrs; int R;
    const int32_t* max_dets; int M;
    const double* area_rng; int A;
    int n_imgs, n_cats;
    std::vector<int64_t> gt_off, dt_off;      // [K*I + 1] bucket offsets (category-major)
    std::vector<int64_t> gt_idx, dt_idx;      // original row indices, original order inside a bucket
    std::vector<PairEval> pairs;              // [K*I]
};

template <class F>
void parallel_for(int64_t n, int threads, F f) {
    if (threads <= 1 || n <= 1) {
        for (int64_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<int64_t> next{0};
    std::vector<std::thread> pool;
    const int nt = (int)std::min<int64_t>(threads, n);
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&] {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n) return;
                f(i);
            }
        });
    for (auto& th : pool) th.join();
}

void bucket(const int32_t* img, const int32_t* cat, int64_t n, int n_imgs, int n_cats, std::vector<int64_t>& off,
            std::vector<int64_t>& idx) {
    const int64_t cells = (int64_t)n_imgs * n_cats;
    off.assign(cells + 1, 0);
    for (int64_t r = 0; r < n; ++r) off[(int64_t)cat[r] * n_imgs + img[r] + 1]++;
    for (int64_t c = 0; c < cells; ++c) off[c + 1] += off[c];
    idx.resize(n);
    std::vector<int64_t> cur(off.begin(), off.end() - 1);
    for (int64_t r = 0; r < n; ++r) idx[cur[(int64_t)cat[r] * n_imgs + img[r]]++] = r;
}

// cocoeval.py:236-313 for every area range at maxDet = maxDets[-1]
void eval_pair(Ctx& c, int64_t cell) {
    PairEval& pe = c.pairs[cell];
    const int64_t g0 = c.gt_off[cell], g1 = c.gt_off[cell + 1], d0 = c.dt_off[cell], d1 = c.dt_off[cell + 1];
    const int G = (int)(g1 - g0);
    if (G == 0 && d1 == d0) return;
    pe.present = true;
    // detections: stable sort by -score, keep maxDets[-1]   (cocoeval.py:177-179, 257-258)
    std::vector<int64_t> dts(c.dt_idx.begin() + d0, c.dt_idx.begin() + d1);
    std::stable_sort(dts.begin(), dts.end(), [&](int64_t a, int64_t b) { return score_before(c.dt_score[a], c.dt_score[b]); });
    const int cap = c.max_dets[c.M - 1];
    if ((int)dts.size() > cap) dts.resize(cap);
    const int D = (int)dts.size();
    pe.D = D; pe.G = G;
    pe.scores.resize(D);
    for (int d = 0; d < D; ++d) pe.scores[d] = c.dt_score[dts[d]];
    const int T = c.T, A = c.A;
    pe.matched.assign((size_t)A * T * D, 0);
    pe.dt_ignore.assign((size_t)A * T * D, 0);
    pe.gt_kept.assign(A, 0);
    // bbIou: inter / (a_d + a_g - inter); crowd GT: inter / a_d
    std::vector<double> iou((size_t)D * G);
    for (int d = 0; d < D; ++d) {
        const double* db = c.dt_box + dts[d] * 4;
        const double da = db[2] * db[3];
        for (int g = 0; g < G; ++g) {
            const int64_t gr = c.gt_idx[g0 + g];
            const double* gb = c.gt_box + gr * 4;
            const double ga = gb[2] * gb[3];
            const double w = std::min(db[0] + db[2], gb[0] + gb[2]) - std::max(db[0], gb[0]);
            const double h = std::min(db[1] + db[3], gb[1] + gb[3]) - std::max(db[1], gb[1]);
            const double inter = (w <= 0 || h <= 0) ? 0.0 : w * h;
            const double uni = c.gt_crowd[gr] ? da : da + ga - inter;
            iou[(size_t)d * G + g] = inter / uni;
        }
    }
    std::vector<int> gtind(G);
    std::vector<uint8_t> g_ig(G), g_crowd(G);
    std::vector<int64_t> gtm((size_t)T * G);
    std::vector<int64_t> dtm((size_t)T * D);
    for (int a = 0; a < A; ++a) {
        const double lo = c.area_rng[a * 2], hi = c.area_rng[a * 2 + 1];
        // ignore flag, then GT sorted with the non-ignored first (stable)   (cocoeval.py:247-256)
        std::vector<uint8_t> ign(G);
        for (int g = 0; g < G; ++g) {
            const int64_t gr = c.gt_idx[g0 + g];
            ign[g] = (c.gt_crowd[gr] || c.gt_area[gr] < lo || c.gt_area[gr] > hi) ? 1 : 0;
        }
        std::iota(gtind.begin(), gtind.end(), 0);
        std::stable_sort(gtind.begin(), gtind.end(), [&](int x, int y) { return ign[x] < ign[y]; });
        int kept = 0;
        for (int g = 0; g < G; ++g) {
            g_ig[g] = ign[gtind[g]];
            g_crowd[g] = c.gt_crowd[c.gt_idx[g0 + gtind[g]]];
            kept += !g_ig[g];
        }
        pe.gt_kept[a] = kept;
        std::fill(gtm.begin(), gtm.end(), 0);
        std::fill(dtm.begin(), dtm.end(), 0);
        uint8_t* mt = pe.matched.data() + (size_t)a * T * D;
        uint8_t* dig = pe.dt_ignore.data() + (size_t)a * T * D;
        if (G && D) {
            for (int t = 0; t < T; ++t) {
                for (int d = 0; d < D; ++d) {
                    double best = std::min(c.iou_thrs[t], 1 - 1e-10);
                    int m = -1;
                    for (int g = 0; g < G; ++g) {
                        if (gtm[(size_t)t * G + g] > 0 && !g_crowd[g]) continue;         // already matched, not crowd
                        if (m > -1 && !g_ig[m] && g_ig[g]) break;                        // regular match beats ignored
                        const double v = iou[(size_t)d * G + gtind[g]];
                        if (v < best) continue;
                        best = v;
                        m = g;
                    }
                    if (m == -1) continue;
                    dig[(size_t)t * D + d] = g_ig[m];
                    dtm[(size_t)t * D + d] = c.gt_id[c.gt_idx[g0 + gtind[m]]];
                    gtm[(size_t)t * G + m] = dts[d] + 1;                                 // detection id = row + 1
                }
            }
        }
        // unmatched detections outside the area range are ignored   (cocoeval.py:300-303)
        for (int d = 0; d < D; ++d) {
            const double* db = c.dt_box + dts[d] * 4;
            const double area = db[2] * db[3];
            const bool outside = area < lo || area > hi;
            for (int t = 0; t < T; ++t) {
                const bool m = dtm[(size_t)t * D + d] != 0;
                mt[(size_t)t * D + d] = m;
                if (!m && outside) dig[(size_t)t * D + d] = 1;
            }
        }
    }
}

// cocoeval.py:315-421 for one (category k, area a, maxDets m) cell
void accumulate_cell(const Ctx& c, int k, int a, int m, double* precision, double* recall) {
    const int T = c.T, R = c.R, K = c.n_cats, A = c.A, M = c.M, I = c.n_imgs;
    const int max_det = c.max_dets[m];
    struct Row { double score; const PairEval* pe; int d; };
    std::vector<Row> rows;
    int64_t npig = 0;
    bool any = false;
    for (int i = 0; i < I; ++i) {
        const PairEval& pe = c.pairs[(int64_t)k * I + i];
        if (!pe.present) continue;
        any = true;
        const int nd = std::min(pe.D, max_det);
        for (int d = 0; d < nd; ++d) rows.push_back({pe.scores[d], &pe, d});
        npig += pe.gt_kept[a];
    }
    if (!any || npig == 0) return;
    std::stable_sort(rows.begin(), rows.end(), [](const Row& x, const Row& y) { return score_before(x.score, y.score); });
    const int nd = (int)rows.size();
    const double eps = std::numeric_limits<double>::epsilon();  // np.spacing(1)
    std::vector<double> rc(nd), pr(nd);
    for (int t = 0; t < T; ++t) {
        double tp = 0, fp = 0;
        for (int j = 0; j < nd; ++j) {
            const PairEval& pe = *rows[j].pe;
            const size_t o = ((size_t)a * T + t) * pe.D + rows[j].d;
            const bool ig = pe.dt_ignore[o], mt = pe.matched[o];
            tp += (mt && !ig);
            fp += (!mt && !ig);
            rc[j] = tp / (double)npig;
            pr[j] = tp / (fp + tp + eps);
        }
        recall[(((size_t)t * K + k) * A + a) * M + m] = nd ? rc[nd - 1] : 0.0;
        for (int j = nd - 1; j > 0; --j)
            if (pr[j] > pr[j - 1]) pr[j - 1] = pr[j];
        for (int r = 0; r < R; ++r) {
            const int pi = (int)(std::lower_bound(rc.begin(), rc.end(), c.rec_thrs[r]) - rc.begin());
            precision[((((size_t)t * R + r) * K + k) * A + a) * M + m] = pi < nd ? pr[pi] : 0.0;
        }
    }
}
}  // namespace

extern "C" int pe_cocoeval_bbox(const int32_t* gt_img, const int32_t* gt_cat, const double* gt_box, const double* gt_area,
                                const uint8_t* gt_crowd, const int64_t* gt_id, int64_t n_gt, const int32_t* dt_img,
                                const int32_t* dt_cat, const double* dt_box, const double* dt_score, int64_t n_dt,
                                int32_t n_imgs, int32_t n_cats, const double* iou_thrs, int32_t T,
                                const double* rec_thrs, int32_t R, const int32_t* max_dets, int32_t M,
                                const double* area_rng, int32_t A, int32_t num_threads, double* precision,
                                double* recall) {
    if (!(precision && recall && iou_thrs && rec_thrs && max_dets && area_rng) || n_gt < 0 || n_dt < 0 || n_imgs < 0 ||
        n_cats < 0 || T < 1 || R < 1 || M < 1 || A < 1 ||
        (n_gt > 0 && !(gt_img && gt_cat && gt_box && gt_area && gt_crowd && gt_id)) ||
        (n_dt > 0 && !(dt_img && dt_cat && dt_box && dt_score))) {
        pe::set_error("pe_cocoeval_bbox: null pointer or bad sizes");
        return PE_ERR_INVALID_ARG;
    }
    for (int64_t r = 0; r < n_gt; ++r)
        if ((unsigned)gt_img[r] >= (unsigned)n_imgs || (unsigned)gt_cat[r] >= (unsigned)n_cats) {
            pe::set_error("pe_cocoeval_bbox: ground-truth row %lld has image %d / category %d out of range", (long long)r,
                          gt_img[r], gt_cat[r]);
            return PE_ERR_INVALID_ARG;
        }
    for (int64_t r = 0; r < n_dt; ++r)
        if ((unsigned)dt_img[r] >= (unsigned)n_imgs || (unsigned)dt_cat[r] >= (unsigned)n_cats) {
            pe::set_error("pe_cocoeval_bbox: detection row %lld has image %d / category %d out of range", (long long)r,
                          dt_img[r], dt_cat[r]);
            return PE_ERR_INVALID_ARG;
        }
    Ctx c;
    c.gt_box = gt_box; c.gt_area = gt_area; c.gt_crowd = gt_crowd; c.gt_id = gt_id;
    c.dt_box = dt_box; c.dt_score = dt_score;
    c.iou_thrs = iou_thrs; c.T = T; c.rec_thrs = rec_thrs; c.R = R; c.max_dets = max_dets; c.M = M;
    c.area_rng = area_rng; c.A = A; c.n_imgs = n_imgs; c.n_cats = n_cats;
    bucket(gt_img, gt_cat, n_gt, n_imgs, n_cats, c.gt_off, c.gt_idx);
    bucket(dt_img, dt_cat, n_dt, n_imgs, n_cats, c.dt_off, c.dt_idx);
    const int64_t cells = (int64_t)n_imgs * n_cats;
    c.pairs.resize(cells);
    int threads = num_threads > 0 ? num_threads : (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    const size_t np = (size_t)T * R * n_cats * A * M, nr = (size_t)T * n_cats * A * M;
    for (size_t i = 0; i < np; ++i) precision[i] = -1.0;
    for (size_t i = 0; i < nr; ++i) recall[i] = -1.0;
    parallel_for(cells, threads, [&](int64_t cell) { eval_pair(c, cell); });
    const int64_t acc = (int64_t)n_cats * A * M;
    parallel_for(acc, threads, [&](int64_t j) {
        const int m = (int)(j % M), a = (int)((j / M) % A), k = (int)(j / ((int64_t)M * A));
        accumulate_cell(c, k, a, m, precision, recall);
    });
    return PE_OK;
}
