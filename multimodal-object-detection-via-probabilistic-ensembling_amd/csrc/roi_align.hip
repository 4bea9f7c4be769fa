// ROIAlign (aligned = True / False, adaptive sampling) over NHWC feature maps for gfx950: all FPN levels
// in ONE launch, level assignment fused, no device synchronisation.
//
// Replaces ROIAlign_forward_cuda / RoIAlignForward (layers/csrc/ROIAlign/ROIAlign_cuda.cu:12-139,310-366;
// arithmetic identical to ROIAlign_cpu.cpp:22-218), the autograd wrapper layers/roi_align.py:10-49, and
// ROIPooler.forward + assign_boxes_to_levels + convert_boxes_to_pooler_format (modeling/poolers.py:13-81,180-235)
// - i.e. 4 nonzero + 4 index_put + 4 launches + 4 cudaDeviceSynchronize per forward in the reference.
//
// The reference's CUDA kernel is one thread per output element over NCHW (uncoalesced).  Here features are
// NHWC: one block per ROI, a thread owns a vector of VEC consecutive channels of one output bin, so every
// bilinear tap is a coalesced 16-byte (fp16 x8) / 16-byte (fp32 x4) load shared by a wavefront's lanes.
// fp32 features: per sample the 4 taps are combined in the reference's order (w1*v1 + w2*v2 + w3*v3 + w4*v4, then
// accumulated, then divided by the sample count) so results are bit-identical to the CPU kernel.
// fp16 features (the detector's path): separable, table-driven form below.
// Output layout: [R, ph, pw, C] (the box head's fc1 weight is permuted to match at load time).
#include <hip/hip_fp16.h>

#include <atomic>

#include "common.h"
#include "test_hooks.h"

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

struct RoiArgs {
    const void* feat[4];
    int fh[4], fw[4];
    float scale[4];
    int num_levels;      // 1: single level (generic op); 4: FPN p2..p5 with level assignment
    int N, C;
    const float* rois;   // [R, 5] (batch, x1, y1, x2, y2)  or  [N, per_image, 4] boxes when rois5 == 0
    int rois5, per_image;
    const int32_t* counts;  // [N] valid boxes per image (boxes mode), may be null
    int R;
    int ph, pw, sampling_ratio, aligned;
    int min_level, max_level, canonical_level;
    float canonical_size;
    void* out;           // [R, ph, pw, C]
    int32_t* out_level;  // optional [R]
    const int32_t* order;   // optional [R]: block -> ROI permutation of roi_order_kernel (processing order only: results do not move)
    int xcd_chunk;          // with `order`: sorted positions per XCD (= ceil(R / 8)), grid = 8 * xcd_chunk
};

template <typename T>
struct Vec;
template <>
struct Vec<_Float16> {
    static constexpr int N = 8;
    __device__ static void load(const _Float16* p, float* v) {
        const half8 h = *reinterpret_cast<const half8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    }
    __device__ static void store(_Float16* p, const float* v) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
        *reinterpret_cast<half8*>(p) = h;
    }
};
template <>
struct Vec<float> {
    static constexpr int N = 4;
    __device__ static void load(const float* p, float* v) {
        const float4v h = *reinterpret_cast<const float4v*>(p);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = h[e];
    }
    __device__ static void store(float* p, const float* v) {
        float4v h = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<float4v*>(p) = h;
    }
};

// Separable form, table driven (fp16 features): the sample grid of a bin is a tensor product, so
//   sum_iy sum_ix bilinear(y_iy, x_ix) = sum_r Wy[r] * sum_c Wx[c] * f(r, c)
// with at most (grid_h + 1) x (grid_w + 1) pixel loads per bin instead of 4 * grid_h * grid_w taps (9 vs 16 at
// grid 2, 25 vs 64 at grid 4).  One thread per (axis, bin) folds the reference's per-sample rules (validity window
// [-1, size], clamps, bilinear split: ROIAlign_cpu.cpp:43-96) into a short weight row in LDS ONCE per ROI; the
// 256 threads then only do load + 8 fused multiply-adds per pixel (the tap form spends ~100 VALU instructions per
// sample on coordinates and fp16 -> fp32 converts, which made it VALU-issue bound).  Summation order differs
// from the reference kernel, which is why fp32 features keep the tap form (bit-exact).
#ifndef ROI_MLP
#define ROI_MLP 4
#endif
constexpr int SEP_MAXB = 8;    // pooled bins per axis
constexpr int SEP_MAXN = 20;   // pixels per bin per axis (grid <= ~18); larger ROIs fall back to the tap form

struct SepTables {
    float w[2][SEP_MAXB][SEP_MAXN];
    int first[2][SEP_MAXB], n[2][SEP_MAXB];
    int fallback;
};

__device__ void sep_build_axis(SepTables& t, int axis, int bin, float start, float bin_size, int grid, int size) {
    int lo_min = 0x7fffffff, hi_max = -1;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            if (hi_max < 0) {       // no valid sample on this axis: an empty window (its weights still zeroed - a neighbour bin's window may be walked)
                t.n[axis][bin] = 0; t.first[axis][bin] = 0;
                for (int i = 0; i < SEP_MAXN; ++i) t.w[axis][bin][i] = 0.f;
                return;
            }
            const int n = hi_max - lo_min + 1;
            if (n > SEP_MAXN) { t.fallback = 1; t.n[axis][bin] = 0; return; }
            t.n[axis][bin] = n; t.first[axis][bin] = lo_min;
            for (int i = 0; i < SEP_MAXN; ++i) t.w[axis][bin][i] = 0.f;   // the whole row: the wave-uniform loop below pads with zero weights
        }
        for (int i = 0; i < grid; ++i) {
            float y = start + (float)(i + .5f) * bin_size / (float)grid;
            if (y < -1.0f || y > (float)size) continue;
            if (y <= 0) y = 0;
            int lo = (int)y, hi;
            if (lo >= size - 1) { hi = lo = size - 1; y = (float)lo; } else hi = lo + 1;
            if (pass == 0) {
                lo_min = min(lo_min, lo); hi_max = max(hi_max, hi);
            } else {
                const float l = y - lo, h = 1.f - l;
                t.w[axis][bin][lo - lo_min] += h;
                t.w[axis][bin][hi - lo_min] += l;
            }
        }
    }
}

// acc += (float)half * w in ONE instruction (v_fma_mix_f32: the fp16 -> fp32 conversion is exact, the FMA rounds once - the bits of
// v_cvt_f32_f16 + v_fma_f32, at 8 instead of 12 issue slots per 16-byte load; the compiler prefers 8 converts + 4 packed FMAs)
__device__ inline float fma_mix_lo(unsigned hp, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(w), "v"(acc));
    return d;
}
__device__ inline float fma_mix_hi(unsigned hp, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(w), "v"(acc));
    return d;
}
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ inline void fma_mix8(const uint4v& h, float w, float* acc) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        acc[2 * d] = fma_mix_lo(h[d], w, acc[2 * d]);
        acc[2 * d + 1] = fma_mix_hi(h[d], w, acc[2 * d + 1]);
    }
}
// lane l takes `yes` when bit l of the wave-uniform mask is set, else `no`: one VALU instruction with the mask in an SGPR pair
__device__ inline unsigned lane_select(unsigned long long mask, unsigned no, unsigned yes) {
    unsigned r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(no), "v"(yes), "s"(mask));
    return r;
}
std::atomic<int> g_roi_fast{1};   // test hook: 0 = the per-lane form for every launch (A/B and the bit-identity test)

// FAST (fp16, C == 256, pooled <= 8 x 8, one pooled row per pass): the wave-uniform form of the separable loop.  A wave owns two
// horizontally adjacent bins (32 lanes x 8 channels each) of one pooled row: the row window (first, n, weights) is the same for both,
// and the column loop runs over the WIDER of the two windows with zero weights padding the narrower one - so the pixel walk
// (row, column, byte offset, loop ends) is scalar arithmetic on the SALU, the lanes only differ by a constant voffset, and the
// loads are buffer loads bounded to the image's level (a padded column past the level's end reads zeros instead of faulting).
// Per pixel and wave: 1 load, 2 LDS reads, 1 multiply, 8 v_fma_mix - ~12 vector issue slots where the per-lane form needs ~29
// (that form is VALU-issue bound: 488 M VALU instructions per launch of 32 000 ROIs = 98 % of the SIMD cycles of its 0.90 ms; this
// one 199 M = 65 % of 0.57 ms, profiles/r05_roi_fast_ab.txt; 4 loads in flight per wave: 3, 6, 8 and 8 waves per SIMD are all
// within 5 %).  A bin's own pixels are accumulated in the
// same row-major order with the same weights; the padding adds w = 0 terms, which leave an accumulator that started at +0 as it is
// -> the bits of the per-lane form (tests/test_ops_gpu.py::test_roi_align_fast_form_is_bit_identical).  Round 6: a padded
// column is not LOADED either - the half-wave whose own window has ended gets an out-of-range per-lane offset (zeros from the bounds
// check), selected by a wave-uniform lane mask in one VALU instruction - so a non-finite pixel next to a bin (0 x Inf = NaN) cannot
// reach a bin that the per-lane form and the reference leave finite, and nothing rests on what lies behind the level's last pixel.
// What remains of the kind (both table forms, finite features unaffected): a bin's table is a DENSE window from its first to its last
// tap; with the adaptive sampling ratio (the detector's: POOLER_SAMPLING_RATIO 0, ceil(bin size) samples) every pixel of the window is
// sampled, with a FIXED ratio and bins wider than `ratio` pixels the pixels between two samples sit in the window at weight zero - a
// non-finite value there makes the bin NaN where the reference, which never reads it, stays finite.
template <typename T, bool SEP, bool FAST>
__global__ __launch_bounds__(256) void roi_align_kernel(RoiArgs a) {
    constexpr int V = Vec<T>::N;
    int r = blockIdx.x;
    if (a.order) {
        // Workgroups go to the 8 XCDs round-robin (MI355X_MICROARCH: blockIdx % 8); give every XCD one CONTIGUOUS stretch of the
        // sorted order (whole images in the detector: 4 of 32 per XCD), so that the ~256 ROIs an XCD works on at a time are
        // neighbours on one feature level of one image and share its L2 - instead of 1/8 of everything everywhere.
        const int pos = (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3);
        if (pos >= a.R) return;
        r = a.order[pos];
    }
    int b;
    float bx1, by1, bx2, by2;
    bool live = true;
    if (a.rois5) {
        const float* p = a.rois + (size_t)r * 5;
        b = (int)p[0]; bx1 = p[1]; by1 = p[2]; bx2 = p[3]; by2 = p[4];
    } else {
        b = r / a.per_image;
        const float* p = a.rois + (size_t)r * 4;
        bx1 = p[0]; by1 = p[1]; bx2 = p[2]; by2 = p[3];
        if (a.counts && (r - b * a.per_image) >= a.counts[b]) live = false;  // padded slot -> zeros
    }
    int lvl = 0;
    if (a.num_levels > 1) {
        // assign_boxes_to_levels (poolers.py:13-44): floor(canonical_level + log2(sqrt(area)/224 + eps))
        const float area = (bx2 - bx1) * (by2 - by1);
        const float sz = sqrtf(area);
        float lv = floorf((float)a.canonical_level + log2f(sz / a.canonical_size + 2.220446049250313e-16f));
        lv = fminf(fmaxf(lv, (float)a.min_level), (float)a.max_level);
        lvl = (int)lv - a.min_level;
    }
    if (a.out_level && threadIdx.x == 0) a.out_level[r] = live ? lvl : -1;
    const int H = a.fh[lvl], W = a.fw[lvl];
    const float scale = a.scale[lvl];
    const T* feat = reinterpret_cast<const T*>(a.feat[lvl]) + (size_t)b * H * W * a.C;
    const float offset = a.aligned ? 0.5f : 0.0f;
    const float start_w = bx1 * scale - offset, start_h = by1 * scale - offset;
    const float end_w = bx2 * scale - offset, end_h = by2 * scale - offset;
    float roi_w = end_w - start_w, roi_h = end_h - start_h;
    if (!a.aligned) { roi_w = fmaxf(roi_w, 1.f); roi_h = fmaxf(roi_h, 1.f); }
    const float bin_h = roi_h / (float)a.ph, bin_w = roi_w / (float)a.pw;
    const int grid_h = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_h / (float)a.ph);
    const int grid_w = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_w / (float)a.pw);
    const float count = (float)max(grid_h * grid_w, 1);
    const int cvec = a.C / V;
    const int items = a.ph * a.pw * cvec;
    T* out = reinterpret_cast<T*>(a.out) + (size_t)r * a.ph * a.pw * a.C;
    __shared__ SepTables tabs;
    if (SEP) {
        if (threadIdx.x == 0) tabs.fallback = (a.ph > SEP_MAXB || a.pw > SEP_MAXB) ? 1 : 0;
        __syncthreads();
        if (live && !tabs.fallback) {
            const int t = threadIdx.x;
            if (t < a.ph) sep_build_axis(tabs, 0, t, start_h + t * bin_h, bin_h, grid_h, H);
            else if (t < a.ph + a.pw) sep_build_axis(tabs, 1, t - a.ph, start_w + (t - a.ph) * bin_w, bin_w, grid_w, W);
        }
        __syncthreads();
    }
    if constexpr (FAST) {
        if (live && !tabs.fallback) {
            const int pw = (int)threadIdx.x >> 5, cv = (int)threadIdx.x & 31;
            const int pwa = __builtin_amdgcn_readfirstlane(pw), pwb = min(pwa + 1, a.pw - 1);
            const int nxm = __builtin_amdgcn_readfirstlane(max(tabs.n[1][pwa], tabs.n[1][pwb]));
            const float* wxr = tabs.w[1][pw];
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(feat), 0, H * W * a.C * 2, 0x00020000);
            const unsigned voff = (unsigned)(tabs.first[1][pw] * a.C + cv * 8) * 2u;
            const unsigned voob = 0xFFFFFFF0u;
            const int nxa = __builtin_amdgcn_readfirstlane(tabs.n[1][pwa]), nxb = __builtin_amdgcn_readfirstlane(tabs.n[1][pwb]);
            auto own = [&](int c) {      // lanes 0-31 (bin pwa) / 32-63 (bin pwb): the column is inside the lane's own window
                return (c < nxa ? 0x00000000FFFFFFFFull : 0ull) | (c < nxb ? 0xFFFFFFFF00000000ull : 0ull);
            };
            const unsigned step_c = (unsigned)a.C * 2u, step_r = (unsigned)(W - nxm + 1) * a.C * 2u;
            // x / count, correctly rounded, in 3 instead of ~10 instructions: with y = RN(1 / count), q = RN(x * y) is refined once
            // through the exact residual (Markstein).  Checked exhaustively against IEEE division over all 2^23 significands for every
            // count = g1 * g2, g <= 22 (tests/csrc/div_check.c, run by tests/test_roi_division_cpu.py; the tables hold at most 20 pixels
            // per bin, i.e. a sampling grid below 20); beyond: divide.
            const bool short_div = grid_h <= 22 && grid_w <= 22;
            const float rcp = 1.f / count;
            constexpr int MLP = ROI_MLP;
            for (int ph = 0; ph < a.ph; ++ph) {
                const int ny = __builtin_amdgcn_readfirstlane(tabs.n[0][ph]);
                const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane(tabs.first[0][ph]) * (unsigned)W * step_c;
                const float* wyr = tabs.w[0][ph];
                const int npx = ny * nxm;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
                unsigned soff = row0;
                int rr = 0, c = 0, i = 0;
                for (; i + MLP <= npx; i += MLP) {
                    uint4v h[MLP];
                    float w[MLP];
#pragma unroll
                    for (int u = 0; u < MLP; ++u) {
                        w[u] = wyr[rr] * wxr[c];
                        h[u] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_select(own(c), voob, voff), soff, 0));
                        const bool wrap = c + 1 == nxm;
                        soff += wrap ? step_r : step_c;
                        c = wrap ? 0 : c + 1;
                        rr += wrap ? 1 : 0;
                    }
#pragma unroll
                    for (int u = 0; u < MLP; ++u) fma_mix8(h[u], w[u], acc);
                }
                if (i < npx) {      // the pass's last, partial group: its loads in flight together, none issued past the end (wave-uniform branches)
                    uint4v h[MLP - 1];
                    float w[MLP - 1];
                    const int rem = npx - i;
#pragma unroll
                    for (int u = 0; u < MLP - 1; ++u)
                        if (u < rem) {
                            w[u] = wyr[rr] * wxr[c];
                            h[u] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_select(own(c), voob, voff), soff, 0));
                            const bool wrap = c + 1 == nxm;
                            soff += wrap ? step_r : step_c;
                            c = wrap ? 0 : c + 1;
                            rr += wrap ? 1 : 0;
                        }
#pragma unroll
                    for (int u = 0; u < MLP - 1; ++u)
                        if (u < rem) fma_mix8(h[u], w[u], acc);
                }
                if (short_div) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float q = acc[e] * rcp;
                        const float rq = __builtin_fmaf(__builtin_fmaf(-count, q, acc[e]), rcp, q);
                        acc[e] = __builtin_isinf(q) ? q : rq;      // an infinite sum stays infinite as under IEEE division (the residual of Inf is Inf - Inf)
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] /= count;
                }
                Vec<T>::store(out + ((size_t)ph * a.pw + pw) * a.C + cv * 8, acc);
            }
            return;
        }
    }
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int cv = it % cvec, bin = it / cvec;
        const int ph = bin / a.pw, pw = bin - ph * a.pw;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        if (live && SEP && !tabs.fallback) {
            const int ny = tabs.n[0][ph], nx = tabs.n[1][pw];
            const int npx = ny * nx;
            const T* base = feat + ((size_t)tabs.first[0][ph] * W + tabs.first[1][pw]) * a.C + cv * V;
            const float* wyr = tabs.w[0][ph];
            const float* wxr = tabs.w[1][pw];
            int rr = 0, c = 0;
            int poff = 0;                                   // element offset of pixel (rr, c): advanced by adds only (an image's
            const int step_c = a.C, step_r = (W - nx + 1) * a.C;   // level is < 2^31 elements) - no 64-bit multiplies per tap
            constexpr int MLP = ROI_MLP;   // independent 16-byte loads in flight per lane
            for (int i = 0; i < npx; i += MLP) {
                half8 h[MLP];
                float w[MLP];
#pragma unroll
                for (int u = 0; u < MLP; ++u) {
                    const bool ok = i + u < npx;
                    const float wv = wyr[rr] * wxr[c];      // (rr, c) is always a valid table slot: read, then mask
                    w[u] = ok ? wv : 0.f;
                    h[u] = *reinterpret_cast<const half8*>(base + poff);
                    if (!ok) h[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};      // a padding slot re-reads the window's last pixel: 0 x Inf would turn an Inf bin into NaN (round 6)
                    const bool adv = i + u + 1 < npx;       // stays on the last pixel past the end; selects, no branches
                    const bool wrap = adv && c + 1 == nx;
                    poff += wrap ? step_r : (adv ? step_c : 0);
                    c = wrap ? 0 : c + (adv ? 1 : 0);
                    rr += wrap ? 1 : 0;
                }
#pragma unroll
                for (int u = 0; u < MLP; ++u)
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = __builtin_fmaf((float)h[u][e], w[u], acc[e]);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] /= count;
        } else if (live) {
            for (int iy = 0; iy < grid_h; ++iy) {
                const float yy = start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)grid_h;
                for (int ix = 0; ix < grid_w; ++ix) {
                    const float xx = start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)grid_w;
                    float x = xx, y = yy;
                    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
                    if (y <= 0) y = 0;
                    if (x <= 0) x = 0;
                    int y_low = (int)y, x_low = (int)x, y_high, x_high;
                    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
                    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
                    const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                    float v1[V], v2[V], v3[V], v4[V];
                    Vec<T>::load(feat + ((size_t)y_low * W + x_low) * a.C + cv * V, v1);
                    Vec<T>::load(feat + ((size_t)y_low * W + x_high) * a.C + cv * V, v2);
                    Vec<T>::load(feat + ((size_t)y_high * W + x_low) * a.C + cv * V, v3);
                    Vec<T>::load(feat + ((size_t)y_high * W + x_high) * a.C + cv * V, v4);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
                }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] /= count;
        }
        Vec<T>::store(out + (size_t)bin * a.C + cv * V, acc);
    }
}

// ---- processing order ("(level, y, x) ROI ordering / XCD-affine mapping") ---------------------------------------
// One workgroup per image sorts its proposals by (dead slot, FPN level, Morton code of the box centre in 32-px cells): bitonic
// sort of 2048 packed 32-bit keys in LDS, the ROI's index in the low 11 bits (keys are unique -> the order is deterministic).
// Proposals arrive in RPN score order, i.e. scattered over the image and the pyramid; ROIAlign reads 100-400 KB of features per
// ROI, neighbours overlap heavily, and the L2 (4 MiB per XCD) only helps if neighbours run at the same time on the same XCD.
constexpr int ORDER_MAX = 2048;

__device__ inline unsigned morton6(unsigned x, unsigned y) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) m |= ((x >> i) & 1u) << (2 * i) | ((y >> i) & 1u) << (2 * i + 1);
    return m;
}

__global__ __launch_bounds__(1024) void roi_order_kernel(const float* boxes, const int32_t* counts, int per_image, int32_t* order,
                                                         int min_level, int max_level, int canonical_level, float canonical_size) {
    __shared__ unsigned keys[ORDER_MAX];
    const int img = blockIdx.x;
    for (int i = threadIdx.x; i < ORDER_MAX; i += blockDim.x) {
        unsigned k = 0xFFFFFFFFu;
        if (i < per_image) {
            const float* p = boxes + ((size_t)img * per_image + i) * 4;
            const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
            const bool dead = counts && i >= counts[img];
            const float sz = sqrtf((x2 - x1) * (y2 - y1));
            float lv = floorf((float)canonical_level + log2f(sz / canonical_size + 2.220446049250313e-16f));
            lv = fminf(fmaxf(lv, (float)min_level), (float)max_level);     // NaN boxes: fmaxf picks min_level - any place will do
            const int lvl = (int)lv - min_level;
            const float cx = (x1 + x2) * 0.5f, cy = (y1 + y2) * 0.5f;
            const unsigned cxq = (unsigned)fminf(fmaxf(cx * (1.f / 32.f), 0.f), 63.f);
            const unsigned cyq = (unsigned)fminf(fmaxf(cy * (1.f / 32.f), 0.f), 63.f);
            k = ((dead ? 1u : 0u) << 25) | ((unsigned)lvl << 23) | (morton6(cxq, cyq) << 11) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int len = 2; len <= ORDER_MAX; len <<= 1)
        for (int str = len >> 1; str > 0; str >>= 1) {
            for (int t = threadIdx.x; t < ORDER_MAX / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (str - 1)), hi = lo + str;
                const bool up = (lo & len) == 0;
                const unsigned a0 = keys[lo], a1 = keys[hi];
                if ((a0 > a1) == up) { keys[lo] = a1; keys[hi] = a0; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < per_image; i += blockDim.x) order[(size_t)img * per_image + i] = img * per_image + (int)(keys[i] & 2047u);
}

// ---- backward (training half) ----------------------------------------------------------------------------
// Replaces RoIAlignBackwardFeature / ROIAlign_backward_cuda (layers/csrc/ROIAlign/ROIAlign_cuda.cu:141-306,369-420; same
// arithmetic as ROIAlign_cpu.cpp:221-394) behind _ROIAlign.backward (layers/roi_align.py:26-42), and - with four levels - the
// backward of ROIPooler.forward's per-level scatter (modeling/poolers.py:180-235) in the same launch.
// Same thread mapping as the forward: one block per ROI, a thread owns V consecutive channels of one bin and hands
// grad * w / count to the four corners of every sample with hardware fp32 atomic adds (NHWC: the V atomics of a tap are
// adjacent addresses).  Like the reference's CUDA kernel the summation order over overlapping ROIs is not defined.
struct RoiBwdArgs {
    RoiArgs f;            // geometry as in the forward; f.out = grad_output [R, ph, pw, C]; f.feat unused
    float* gin[4];        // [N, H_l, W_l, C] fp32 per level, accumulated into
};

template <typename T>
__global__ __launch_bounds__(256) void roi_align_backward_kernel(RoiBwdArgs b) {
    const RoiArgs& a = b.f;
    constexpr int V = Vec<T>::N;
    const int r = blockIdx.x;
    int img;
    float bx1, by1, bx2, by2;
    if (a.rois5) {
        const float* p = a.rois + (size_t)r * 5;
        img = (int)p[0]; bx1 = p[1]; by1 = p[2]; bx2 = p[3]; by2 = p[4];
    } else {
        img = r / a.per_image;
        const float* p = a.rois + (size_t)r * 4;
        bx1 = p[0]; by1 = p[1]; bx2 = p[2]; by2 = p[3];
        if (a.counts && (r - img * a.per_image) >= a.counts[img]) return;   // padded slot: its output was zeros, no gradient
    }
    int lvl = 0;
    if (a.num_levels > 1) {
        const float area = (bx2 - bx1) * (by2 - by1);
        const float sz = sqrtf(area);
        float lv = floorf((float)a.canonical_level + log2f(sz / a.canonical_size + 2.220446049250313e-16f));
        lv = fminf(fmaxf(lv, (float)a.min_level), (float)a.max_level);
        lvl = (int)lv - a.min_level;
    }
    const int H = a.fh[lvl], W = a.fw[lvl];
    const float scale = a.scale[lvl];
    float* gin = b.gin[lvl] + (size_t)img * H * W * a.C;
    const float offset = a.aligned ? 0.5f : 0.0f;
    const float start_w = bx1 * scale - offset, start_h = by1 * scale - offset;
    const float end_w = bx2 * scale - offset, end_h = by2 * scale - offset;
    float roi_w = end_w - start_w, roi_h = end_h - start_h;
    if (!a.aligned) { roi_w = fmaxf(roi_w, 1.f); roi_h = fmaxf(roi_h, 1.f); }
    const float bin_h = roi_h / (float)a.ph, bin_w = roi_w / (float)a.pw;
    const int grid_h = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_h / (float)a.ph);
    const int grid_w = a.sampling_ratio > 0 ? a.sampling_ratio : (int)ceilf(roi_w / (float)a.pw);
    const float count = (float)(grid_h * grid_w);
    const int cvec = a.C / V;
    const int items = a.ph * a.pw * cvec;
    const T* gout = reinterpret_cast<const T*>(a.out) + (size_t)r * a.ph * a.pw * a.C;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int cv = it % cvec, bin = it / cvec;
        const int ph = bin / a.pw, pw = bin - ph * a.pw;
        float g[V];
        Vec<T>::load(gout + (size_t)bin * a.C + cv * V, g);
        for (int iy = 0; iy < grid_h; ++iy) {
            const float yy = start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ++ix) {
                const float xx = start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)grid_w;
                float x = xx, y = yy;
                if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
                if (y <= 0) y = 0;
                if (x <= 0) x = 0;
                int y_low = (int)y, x_low = (int)x, y_high, x_high;
                if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
                if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
                const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                float* p1 = gin + ((size_t)y_low * W + x_low) * a.C + cv * V;
                float* p2 = gin + ((size_t)y_low * W + x_high) * a.C + cv * V;
                float* p3 = gin + ((size_t)y_high * W + x_low) * a.C + cv * V;
                float* p4 = gin + ((size_t)y_high * W + x_high) * a.C + cv * V;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    unsafeAtomicAdd(p1 + e, g[e] * w1 / count);
                    unsafeAtomicAdd(p2 + e, g[e] * w2 / count);
                    unsafeAtomicAdd(p3 + e, g[e] * w3 / count);
                    unsafeAtomicAdd(p4 + e, g[e] * w4 / count);
                }
            }
        }
    }
}
}  // namespace

extern "C" int pe_roi_align_backward_nhwc(const void* grad_output, int32_t dtype, const int32_t* feat_hw_host,
                                          const float* scales_host, int32_t num_levels, int32_t N, int32_t C, const float* rois,
                                          int32_t rois_have_batch_index, int32_t num_rois, int32_t per_image,
                                          const int32_t* counts, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio,
                                          int32_t aligned, float* const* grad_feats_host, void* stream) {
    PE_CHECK_ARG(num_levels == 1 || num_levels == 4, "pe_roi_align_backward_nhwc: num_levels %d not in {1,4}", num_levels);
    PE_CHECK_ARG(dtype == 0 || dtype == 1, "pe_roi_align_backward_nhwc: dtype %d (0 = fp16, 1 = fp32 grad_output)", dtype);
    PE_CHECK_ARG(feat_hw_host && scales_host && grad_feats_host, "pe_roi_align_backward_nhwc: null level tables");
    PE_CHECK_ARG(C % (dtype == 0 ? 8 : 4) == 0, "pe_roi_align_backward_nhwc: C %d not a multiple of the vector width", C);
    PE_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "pe_roi_align_backward_nhwc: bad pooled size");
    if (num_rois == 0) return PE_OK;
    PE_CHECK_ARG(rois && grad_output, "pe_roi_align_backward_nhwc: null pointer");
    PE_CHECK_ARG(rois_have_batch_index || per_image > 0, "pe_roi_align_backward_nhwc: per_image required in boxes mode");
    RoiBwdArgs b{};
    RoiArgs& a = b.f;
    for (int l = 0; l < num_levels; ++l) {
        a.fh[l] = feat_hw_host[2 * l]; a.fw[l] = feat_hw_host[2 * l + 1]; a.scale[l] = scales_host[l];
        b.gin[l] = grad_feats_host[l];
        PE_CHECK_ARG(b.gin[l] != nullptr, "pe_roi_align_backward_nhwc: null gradient pointer");
    }
    a.num_levels = num_levels; a.N = N; a.C = C; a.rois = rois; a.rois5 = rois_have_batch_index;
    a.per_image = per_image; a.counts = counts; a.R = num_rois; a.ph = pooled_h; a.pw = pooled_w;
    a.sampling_ratio = sampling_ratio; a.aligned = aligned;
    a.min_level = 2; a.max_level = 5; a.canonical_level = 4; a.canonical_size = 224.f;
    a.out = const_cast<void*>(grad_output);
    if (dtype == 0)
        hipLaunchKernelGGL((roi_align_backward_kernel<_Float16>), dim3(num_rois), dim3(256), 0, (hipStream_t)stream, b);
    else
        hipLaunchKernelGGL((roi_align_backward_kernel<float>), dim3(num_rois), dim3(256), 0, (hipStream_t)stream, b);
    PE_CHECK_LAUNCH("pe_roi_align_backward_nhwc");
    return PE_OK;
}

static int roi_align_forward(const void* const* feats_host, const int32_t* feat_hw_host, const float* scales_host,
                             int32_t num_levels, int32_t N, int32_t C, int32_t dtype, const float* rois,
                             int32_t rois_have_batch_index, int32_t num_rois, int32_t per_image,
                             const int32_t* counts, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio,
                             int32_t aligned, void* output, int32_t* out_level, int32_t* order_workspace, void* stream) {
    PE_CHECK_ARG(num_levels == 1 || num_levels == 4, "pe_roi_align_nhwc: num_levels %d not in {1,4}", num_levels);
    PE_CHECK_ARG(dtype == 0 || dtype == 1, "pe_roi_align_nhwc: dtype %d (0 = fp16, 1 = fp32)", dtype);
    PE_CHECK_ARG(feats_host && feat_hw_host && scales_host, "pe_roi_align_nhwc: null level tables");
    PE_CHECK_ARG(C % (dtype == 0 ? 8 : 4) == 0, "pe_roi_align_nhwc: C %d not a multiple of the vector width", C);
    PE_CHECK_ARG(pooled_h > 0 && pooled_w > 0, "pe_roi_align_nhwc: bad pooled size");
    if (num_rois == 0) return PE_OK;  // empty in -> empty out, no launch (ROIAlign_cuda.cu:343-346)
    PE_CHECK_ARG(rois && output, "pe_roi_align_nhwc: null pointer");
    PE_CHECK_ARG(rois_have_batch_index || per_image > 0, "pe_roi_align_nhwc: per_image required in boxes mode");
    RoiArgs a{};
    for (int l = 0; l < num_levels; ++l) {
        a.feat[l] = feats_host[l]; a.fh[l] = feat_hw_host[2 * l]; a.fw[l] = feat_hw_host[2 * l + 1];
        a.scale[l] = scales_host[l];
        PE_CHECK_ARG(a.feat[l] != nullptr, "pe_roi_align_nhwc: null feature pointer");
    }
    a.num_levels = num_levels; a.N = N; a.C = C; a.rois = rois; a.rois5 = rois_have_batch_index;
    a.per_image = per_image; a.counts = counts; a.R = num_rois; a.ph = pooled_h; a.pw = pooled_w;
    a.sampling_ratio = sampling_ratio; a.aligned = aligned;
    a.min_level = 2; a.max_level = 5; a.canonical_level = 4; a.canonical_size = 224.f;
    a.out = output; a.out_level = out_level;
    int grid = num_rois;
    if (order_workspace && !rois_have_batch_index && per_image <= ORDER_MAX && num_rois == N * per_image) {
        hipLaunchKernelGGL(roi_order_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, rois, counts, per_image, order_workspace,
                           a.min_level, num_levels > 1 ? a.max_level : a.min_level, a.canonical_level, a.canonical_size);
        PE_CHECK_LAUNCH("pe_roi_align_nhwc_sorted (order)");
        a.order = order_workspace;
        a.xcd_chunk = (num_rois + 7) / 8;
        grid = 8 * a.xcd_chunk;
    }
    // (tried in r01 and dropped: a wave-per-bin variant with scalar sample math and 8-byte lanes: 1.61 vs 1.47 ms)
    if (dtype == 0) {
        // one pooled row per pass when it fits (C = 256: 7 bins x 32 lanes = 224 threads, 7 full passes instead of
        // 6 full + 1 one-eighth-full pass of 256)
        const int row_threads = (C / 8) * pooled_w;
        const int threads = row_threads <= 256 ? row_threads : 256;
        const int fast = g_roi_fast.load(std::memory_order_relaxed);
        if (C == 256 && pooled_h <= SEP_MAXB && pooled_w <= SEP_MAXB && fast) {
            hipLaunchKernelGGL((roi_align_kernel<_Float16, true, true>), dim3(grid), dim3(threads), 0, (hipStream_t)stream, a);
        } else
            hipLaunchKernelGGL((roi_align_kernel<_Float16, true, false>), dim3(grid), dim3(threads), 0, (hipStream_t)stream, a);
    }
    else
        hipLaunchKernelGGL((roi_align_kernel<float, false, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_roi_align_nhwc");
    return PE_OK;
}

extern "C" int pe_test_set_roi_fast(int on) {
    g_roi_fast.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

extern "C" int pe_roi_align_nhwc(const void* const* feats_host, const int32_t* feat_hw_host, const float* scales_host,
                                 int32_t num_levels, int32_t N, int32_t C, int32_t dtype, const float* rois,
                                 int32_t rois_have_batch_index, int32_t num_rois, int32_t per_image,
                                 const int32_t* counts, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio,
                                 int32_t aligned, void* output, int32_t* out_level, void* stream) {
    return roi_align_forward(feats_host, feat_hw_host, scales_host, num_levels, N, C, dtype, rois, rois_have_batch_index, num_rois,
                             per_image, counts, pooled_h, pooled_w, sampling_ratio, aligned, output, out_level, nullptr, stream);
}

extern "C" int pe_roi_align_nhwc_sorted(const void* const* feats_host, const int32_t* feat_hw_host, const float* scales_host,
                                        int32_t num_levels, int32_t N, int32_t C, int32_t dtype, const float* boxes,
                                        int32_t per_image, const int32_t* counts, int32_t pooled_h, int32_t pooled_w,
                                        int32_t sampling_ratio, int32_t aligned, void* output, int32_t* out_level,
                                        int32_t* order_workspace, void* stream) {
    PE_CHECK_ARG(order_workspace != nullptr, "pe_roi_align_nhwc_sorted: null order workspace ([N * per_image] int32)");
    PE_CHECK_ARG(per_image > 0, "pe_roi_align_nhwc_sorted: per_image required");
    return roi_align_forward(feats_host, feat_hw_host, scales_host, num_levels, N, C, dtype, boxes, 0, N * per_image, per_image, counts,
                             pooled_h, pooled_w, sampling_ratio, aligned, output, out_level, order_workspace, stream);
}
