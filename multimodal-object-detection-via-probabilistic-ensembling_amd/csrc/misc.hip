// Small HBM-bound layout / pooling kernels of the detector front end (gfx950).
//   pe_preprocess_pack   : resize (optional) + normalise + zero-pad + NHWC4 fp16 pack of one image
//                          (GeneralizedRCNN.preprocess_image meta_arch/rcnn.py:269-286,
//                           ImageList.from_tensors structures/image_list.py:51-102,
//                           ResizeShortestEdge/ResizeTransform data/transforms/transform.py:81-98)
//   pe_maxpool3x3s2_nhwc : the stem's max_pool2d(3, 2, 1) (backbone/resnet.py:383)
//   pe_subsample2_nhwc   : LastLevelMaxPool = max_pool2d(k=1, s=2) = x[:, ::2, ::2] (backbone/fpn.py:166-178)
#include <hip/hip_fp16.h>

#include "common.h"

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct PackArgs {
    const void* src;  // HWC uint8 / HWC float32 / CHW float32
    int src_kind;     // 0: HWC u8, 1: HWC f32, 2: CHW f32
    int src_h, src_w, src_c;
    int ch0, nch;     // channels [ch0, ch0+nch) of the source feed output channels 0..nch-1 (nch <= 4)
    int flip_rgb;     // reverse the first 3 selected channels (RGB <-> BGR, engine/defaults.py:188-190)
    int dst_h, dst_w;  // resized size (== src size when no resize)
    int pad_h, pad_w;  // padded size (multiple of 32)
    float mean[4], inv_std[4];
    _Float16* dst;  // [pad_h, pad_w, 4]
    size_t src_image_bytes;  // batched form: image z starts at src + z * src_image_bytes
};

__device__ __forceinline__ float src_at(const PackArgs& a, int y, int x, int c) {
    if (a.src_kind == 0) return (float)reinterpret_cast<const unsigned char*>(a.src)[((size_t)y * a.src_w + x) * a.src_c + c];
    if (a.src_kind == 1) return reinterpret_cast<const float*>(a.src)[((size_t)y * a.src_w + x) * a.src_c + c];
    return reinterpret_cast<const float*>(a.src)[((size_t)c * a.src_h + y) * a.src_w + x];
}

__global__ void preprocess_pack_kernel(PackArgs a) {
    a.src = reinterpret_cast<const unsigned char*>(a.src) + (size_t)blockIdx.z * a.src_image_bytes;
    a.dst += (size_t)blockIdx.z * a.pad_h * a.pad_w * 4;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= a.pad_w) return;
    half4 o = {0, 0, 0, 0};
    if (y < a.dst_h && x < a.dst_w) {
        const bool resize = a.dst_h != a.src_h || a.dst_w != a.src_w;
        float sy = 0.f, sx = 0.f;
        int y0 = y, x0 = x, y1 = y, x1 = x;
        if (resize) {
            // cv2.resize(..., INTER_LINEAR) on floating-point images (the reference's 4- / 6-channel branch,
            // transform.py:82-91), restated from OpenCV's resizeGeneric_ (third-party, absent here: parity unpinned):
            // source coordinate in double, cast to float; border taps collapse onto the edge pixel; float weights;
            // horizontal then vertical pass accumulated in double
            const double fy = ((double)y + 0.5) * ((double)a.src_h / (double)a.dst_h) - 0.5;
            const double fx = ((double)x + 0.5) * ((double)a.src_w / (double)a.dst_w) - 0.5;
            sy = (float)fy; sx = (float)fx;
            y0 = (int)floorf(sy); x0 = (int)floorf(sx);
            sy -= (float)y0; sx -= (float)x0;
            if (y0 < 0) { y0 = 0; sy = 0.f; }
            if (x0 < 0) { x0 = 0; sx = 0.f; }
            if (y0 >= a.src_h - 1) { y0 = a.src_h - 1; sy = 0.f; }
            if (x0 >= a.src_w - 1) { x0 = a.src_w - 1; sx = 0.f; }
            y1 = min(y0 + 1, a.src_h - 1); x1 = min(x0 + 1, a.src_w - 1);
        }
        for (int c = 0; c < a.nch; ++c) {
            const int sc = a.ch0 + ((a.flip_rgb && c < 3) ? 2 - c : c);
            float v;
            if (resize) {
                const double top = (double)src_at(a, y0, x0, sc) * (double)(1.f - sx) + (double)src_at(a, y0, x1, sc) * (double)sx;
                const double bot = (double)src_at(a, y1, x0, sc) * (double)(1.f - sx) + (double)src_at(a, y1, x1, sc) * (double)sx;
                v = (float)(top * (double)(1.f - sy) + bot * (double)sy);
                if (a.src_kind == 0) v = rintf(v);  // uint8 in -> uint8 out (3-channel uint8 frames take the Pillow-exact kernel)
            } else {
                v = src_at(a, y, x, sc);
            }
            o[c] = (_Float16)((v - a.mean[c]) * a.inv_std[c]);
        }
    }
    *reinterpret_cast<half4*>(a.dst + ((size_t)y * a.pad_w + x) * 4) = o;
}

// Pillow-exact bilinear resize of uint8 images (two passes, each rounded to uint8: libImaging/Resample.c
// ImagingResampleHorizontal_8bpc / Vertical_8bpc) fused with normalise + pad + NHWC4 pack.  xtab / ytab rows:
// (first tap, tap count, 22-bit weights...) from proben_amd.data.pil_bilinear_tables.
struct PilArgs {
    const unsigned char* src;   // [N, src_h, src_w, src_c] uint8
    int src_h, src_w, src_c, ch0, nch, flip_rgb;
    int dst_h, dst_w, pad_h, pad_w;
    const int32_t* xtab; int xk;   // [dst_w, 2 + xk]
    const int32_t* ytab; int yk;   // [dst_h, 2 + yk]
    float mean[4], inv_std[4];
    _Float16* dst;
};

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one output pixel, both passes from memory (the form of rounds 1-4; now the path of tiles whose source window does not fit the LDS)
template <int NCH, bool FLIP>
__device__ __forceinline__ half4 pil_pixel(const PilArgs& a, const unsigned char* src, int x, int y) {
    half4 o = {0, 0, 0, 0};
    {
        const int32_t* xt = a.xtab + (size_t)x * (2 + a.xk);
        const int32_t* yt = a.ytab + (size_t)y * (2 + a.yk);   // wave-uniform: scalar loads
        const int xmin = xt[0], nx = xt[1], ymin = yt[0], ny = yt[1];
        int accv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) accv[c] = 1 << 21;
        for (int j = 0; j < ny; ++j) {
            const unsigned char* row = src + ((size_t)(ymin + j) * a.src_w + xmin) * a.src_c + a.ch0;
            int acch[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) acch[c] = 1 << 21;
            for (int i = 0; i < nx; ++i) {
                const int k = xt[2 + i];
#pragma unroll
                for (int c = 0; c < NCH; ++c) acch[c] += (int)row[i * a.src_c + ((FLIP && c < 3) ? 2 - c : c)] * k;
            }
            const int k = yt[2 + j];
#pragma unroll
            for (int c = 0; c < NCH; ++c) accv[c] += clip8(acch[c] >> 22) * k;   // the horizontal pass result is a uint8 image
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) o[c] = (_Float16)(((float)clip8(accv[c] >> 22) - a.mean[c]) * a.inv_std[c]);
    }
    return o;
}

// A workgroup makes a 256-column x 16-row tile of the output.  Pillow's horizontal pass produces a uint8 image of the SOURCE's
// height; the tile needs rows r_lo .. r_hi of it (13 of them for 16 output rows at the detectors' 1.5625x), each computed ONCE per
// column into LDS (thread = column: its tap table row stays in registers), and the vertical pass reads them back - the per-pixel
// form recomputed the horizontal pass for every vertical tap: 27 single-byte loads + the table reads per output pixel, ~40 vector
// memory instructions, which is what bound it (0.20 ms for 32 frames; TA-issue, not bytes).  Same integer arithmetic, same bits.
constexpr int PIL_TX = 256, PIL_TY = 16, PIL_MAXR = 16, PIL_MAXK = 8;
template <int NCH, bool FLIP>
__global__ __launch_bounds__(PIL_TX) void preprocess_pil_kernel(PilArgs a) {
    __shared__ unsigned hbuf[PIL_MAXR][PIL_TX];
    const unsigned char* src = a.src + (size_t)blockIdx.z * a.src_h * a.src_w * a.src_c;
    _Float16* dst = a.dst + (size_t)blockIdx.z * a.pad_h * a.pad_w * 4;
    const int tid = threadIdx.x;
    const int x = blockIdx.x * PIL_TX + tid;
    const int y0 = blockIdx.y * PIL_TY;
    const int rows = min(PIL_TY, a.pad_h - y0);
    const bool col_live = x < a.dst_w;
    int r_lo = 0, nr = 0;
    if (y0 < a.dst_h && blockIdx.x * PIL_TX < a.dst_w) {      // the tile has resized pixels
        const int yl = min(y0 + PIL_TY, a.dst_h);
        int lo = 0x7FFFFFFF, hi = 0;     // the union of the tile rows' windows (the tables carry trimmed windows: not monotone row to row)
        for (int y = y0; y < yl; ++y) {
            const int32_t* t = a.ytab + (size_t)y * (2 + a.yk);   // wave-uniform: scalar loads
            lo = min(lo, t[0]);
            hi = max(hi, t[0] + t[1]);
        }
        r_lo = lo;
        nr = hi - lo;
    }
    const bool tiled = nr > 0 && nr <= PIL_MAXR && a.xk <= PIL_MAXK;     // workgroup-uniform
    if (tiled) {
        if (col_live) {
            const int32_t* xt = a.xtab + (size_t)x * (2 + a.xk);
            const int xmin = xt[0], nx = xt[1];
            int kx[PIL_MAXK];
#pragma unroll
            for (int i = 0; i < PIL_MAXK; ++i) kx[i] = i < nx ? xt[2 + i] : 0;
            for (int r = 0; r < nr; ++r) {
                const unsigned char* row = src + ((size_t)(r_lo + r) * a.src_w + xmin) * a.src_c + a.ch0;
                int acch[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) acch[c] = 1 << 21;
#pragma unroll
                for (int i = 0; i < PIL_MAXK; ++i)
                    if (i < nx) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) acch[c] += (int)row[i * a.src_c + ((FLIP && c < 3) ? 2 - c : c)] * kx[i];
                    }
                // the horizontal pass result is a uint8 image.  Byte stores: packing the three with shifts and ORs makes hipcc 7.2 emit
                // v_ashr_pk_u8_i32 for two of them and OR the third into its UPPER half, which that instruction does not clear
                // (garbage in channel 2 on gfx950; found by tests/test_ops_gpu.py::test_preprocess_pil_exact_vs_oracle)
                unsigned char* hb = reinterpret_cast<unsigned char*>(&hbuf[r][tid]);
#pragma unroll
                for (int c = 0; c < NCH; ++c) hb[c] = (unsigned char)clip8(acch[c] >> 22);
            }
        }
        __syncthreads();
    }
    if (x >= a.pad_w) return;
    for (int yy = 0; yy < rows; ++yy) {
        const int y = y0 + yy;
        half4 o = {0, 0, 0, 0};
        if (y < a.dst_h && col_live) {
            if (tiled) {
                const int32_t* yt = a.ytab + (size_t)y * (2 + a.yk);   // wave-uniform: scalar loads
                const int ymin = yt[0], ny = yt[1];
                int accv[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) accv[c] = 1 << 21;
                for (int j = 0; j < ny; ++j) {
                    const unsigned h = hbuf[ymin - r_lo + j][tid];
                    const int k = yt[2 + j];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) accv[c] += (int)((h >> (8 * c)) & 255u) * k;
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) o[c] = (_Float16)(((float)clip8(accv[c] >> 22) - a.mean[c]) * a.inv_std[c]);
            } else {
                o = pil_pixel<NCH, FLIP>(a, src, x, y);
            }
        }
        *reinterpret_cast<half4*>(dst + ((size_t)y * a.pad_w + x) * 4) = o;
    }
}

__global__ void maxpool3x3s2_kernel(const _Float16* in, _Float16* out, int N, int H, int W, int C, int Ho, int Wo) {
    const int cv = C / 8;
    const size_t total = (size_t)N * Ho * Wo * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv);
        size_t t = i / cv;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const half8 v = *reinterpret_cast<const half8*>(in + (((size_t)n * H + ih) * W + iw) * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = __builtin_elementwise_maximum(m[e], (float)v[e]);
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)m[e];
        *reinterpret_cast<half8*>(out + i * 8) = o;
    }
}

__global__ void subsample2_kernel(const _Float16* in, _Float16* out, int N, int H, int W, int C, int Ho, int Wo) {
    const int cv = C / 8;
    const size_t total = (size_t)N * Ho * Wo * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv);
        size_t t = i / cv;
        const int ow = (int)(t % Wo); t /= Wo;
        const int oh = (int)(t % Ho);
        const int n = (int)(t / Ho);
        *reinterpret_cast<half8*>(out + i * 8) =
            *reinterpret_cast<const half8*>(in + (((size_t)n * H + oh * 2) * W + ow * 2) * C + c8 * 8);
    }
}
}  // namespace

static int preprocess_launch(const void* src, int32_t num_images, int32_t src_kind, int32_t src_h, int32_t src_w,
                             int32_t src_c, int32_t ch0, int32_t nch, int32_t flip_rgb, int32_t dst_h, int32_t dst_w,
                             int32_t pad_h, int32_t pad_w, const float* mean_host, const float* std_host, void* dst,
                             void* stream) {
    PE_CHECK_ARG(src && dst && mean_host && std_host, "pe_preprocess_pack: null pointer");
    PE_CHECK_ARG(src_kind >= 0 && src_kind <= 2, "pe_preprocess_pack: src_kind %d", src_kind);
    PE_CHECK_ARG(nch >= 1 && nch <= 4 && ch0 >= 0 && ch0 + nch <= src_c, "pe_preprocess_pack: channel window [%d,%d) of %d",
                 ch0, ch0 + nch, src_c);
    PE_CHECK_ARG(dst_h <= pad_h && dst_w <= pad_w && dst_h > 0 && dst_w > 0, "pe_preprocess_pack: bad sizes");
    PackArgs a{};
    a.src = src; a.src_kind = src_kind; a.src_h = src_h; a.src_w = src_w; a.src_c = src_c; a.ch0 = ch0; a.nch = nch;
    a.flip_rgb = flip_rgb; a.dst_h = dst_h; a.dst_w = dst_w; a.pad_h = pad_h; a.pad_w = pad_w; a.dst = (_Float16*)dst;
    a.src_image_bytes = (size_t)src_h * src_w * src_c * (src_kind == 0 ? 1 : 4);
    PE_CHECK_ARG(num_images >= 1 && num_images <= 65535, "pe_preprocess_pack: num_images %d", num_images);
    for (int c = 0; c < 4; ++c) {
        a.mean[c] = c < nch ? mean_host[c] : 0.f;
        a.inv_std[c] = c < nch ? 1.f / std_host[c] : 0.f;
    }
    hipLaunchKernelGGL(preprocess_pack_kernel, dim3(pe::ceil_div(pad_w, 256), pad_h, num_images), dim3(256), 0,
                       (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_preprocess_pack");
    return PE_OK;
}

extern "C" int pe_preprocess_pack(const void* src, int32_t src_kind, int32_t src_h, int32_t src_w, int32_t src_c,
                                  int32_t ch0, int32_t nch, int32_t flip_rgb, int32_t dst_h, int32_t dst_w,
                                  int32_t pad_h, int32_t pad_w, const float* mean_host, const float* std_host,
                                  void* dst, void* stream) {
    return preprocess_launch(src, 1, src_kind, src_h, src_w, src_c, ch0, nch, flip_rgb, dst_h, dst_w, pad_h, pad_w,
                             mean_host, std_host, dst, stream);
}

extern "C" int pe_preprocess_pack_batch(const void* src, int32_t num_images, int32_t src_kind, int32_t src_h,
                                        int32_t src_w, int32_t src_c, int32_t ch0, int32_t nch, int32_t flip_rgb,
                                        int32_t dst_h, int32_t dst_w, int32_t pad_h, int32_t pad_w,
                                        const float* mean_host, const float* std_host, void* dst, void* stream) {
    return preprocess_launch(src, num_images, src_kind, src_h, src_w, src_c, ch0, nch, flip_rgb, dst_h, dst_w, pad_h,
                             pad_w, mean_host, std_host, dst, stream);
}

extern "C" int pe_preprocess_pack_pil_u8(const void* src, int32_t num_images, int32_t src_h, int32_t src_w, int32_t src_c,
                                         int32_t ch0, int32_t nch, int32_t flip_rgb, int32_t dst_h, int32_t dst_w,
                                         int32_t pad_h, int32_t pad_w, const float* mean_host, const float* std_host,
                                         const int32_t* xtab, int32_t xk, const int32_t* ytab, int32_t yk, void* dst,
                                         void* stream) {
    PE_CHECK_ARG(src && dst && mean_host && std_host && xtab && ytab, "pe_preprocess_pack_pil_u8: null pointer");
    PE_CHECK_ARG(nch >= 1 && nch <= 4 && ch0 >= 0 && ch0 + nch <= src_c, "pe_preprocess_pack_pil_u8: channel window [%d,%d) of %d",
                 ch0, ch0 + nch, src_c);
    PE_CHECK_ARG(dst_h <= pad_h && dst_w <= pad_w && dst_h > 0 && dst_w > 0 && xk >= 1 && yk >= 1, "pe_preprocess_pack_pil_u8: bad sizes");
    PE_CHECK_ARG(num_images >= 1 && num_images <= 65535, "pe_preprocess_pack_pil_u8: num_images %d", num_images);
    PilArgs a{};
    a.src = (const unsigned char*)src; a.src_h = src_h; a.src_w = src_w; a.src_c = src_c; a.ch0 = ch0; a.nch = nch;
    a.flip_rgb = flip_rgb; a.dst_h = dst_h; a.dst_w = dst_w; a.pad_h = pad_h; a.pad_w = pad_w;
    a.xtab = xtab; a.xk = xk; a.ytab = ytab; a.yk = yk; a.dst = (_Float16*)dst;
    for (int c = 0; c < 4; ++c) {
        a.mean[c] = c < nch ? mean_host[c] : 0.f;
        a.inv_std[c] = c < nch ? 1.f / std_host[c] : 0.f;
    }
    const dim3 grid(pe::ceil_div(pad_w, PIL_TX), pe::ceil_div(pad_h, PIL_TY), num_images), block(PIL_TX);
    hipStream_t st = (hipStream_t)stream;
#define PE_PIL_LAUNCH(N, F) hipLaunchKernelGGL((preprocess_pil_kernel<N, F>), grid, block, 0, st, a)
    switch (nch * 2 + (flip_rgb ? 1 : 0)) {
        case 2: PE_PIL_LAUNCH(1, false); break;
        case 3: PE_PIL_LAUNCH(1, true); break;
        case 4: PE_PIL_LAUNCH(2, false); break;
        case 5: PE_PIL_LAUNCH(2, true); break;
        case 6: PE_PIL_LAUNCH(3, false); break;
        case 7: PE_PIL_LAUNCH(3, true); break;
        case 8: PE_PIL_LAUNCH(4, false); break;
        default: PE_PIL_LAUNCH(4, true); break;
    }
#undef PE_PIL_LAUNCH
    PE_CHECK_LAUNCH("pe_preprocess_pack_pil_u8");
    return PE_OK;
}

extern "C" int pe_maxpool3x3s2_nhwc(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    PE_CHECK_ARG(in && out && C % 8 == 0, "pe_maxpool3x3s2_nhwc: bad args (C %% 8 == 0 required)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int grid = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in,
                       (_Float16*)out, N, H, W, C, Ho, Wo);
    PE_CHECK_LAUNCH("pe_maxpool3x3s2_nhwc");
    return PE_OK;
}

extern "C" int pe_subsample2_nhwc(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    PE_CHECK_ARG(in && out && C % 8 == 0, "pe_subsample2_nhwc: bad args (C %% 8 == 0 required)");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    const int grid = (int)std::min<size_t>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in,
                       (_Float16*)out, N, H, W, C, Ho, Wo);
    PE_CHECK_LAUNCH("pe_subsample2_nhwc");
    return PE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone forms of two pieces the detector kernels carry fused (rpn.hip decodes only survivors against analytic
// anchors; boxhead.hip decodes per class): the reference exposes them as Python API, so they exist as ops too.
// Compiled with -ffp-contract=off: the expressions round like the reference's separate torch ops.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ void apply_deltas_kernel(const float* deltas, const float* boxes, int N, int k, float wx, float wy, float ww,
                                    float wh, float clamp, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * k) return;
    const int n = idx / k;
    const float* b = boxes + (size_t)n * 4;
    const float* d = deltas + (size_t)idx * 4;
    const float wd = b[2] - b[0], ht = b[3] - b[1];
    const float cx = b[0] + 0.5f * wd, cy = b[1] + 0.5f * ht;
    const float dx = d[0] / wx, dy = d[1] / wy;
    const float dw = fminf(d[2] / ww, clamp), dh = fminf(d[3] / wh, clamp);
    const float pcx = dx * wd + cx, pcy = dy * ht + cy;
    const float pw = expf(dw) * wd, ph = expf(dh) * ht;
    float* o = out + (size_t)idx * 4;
    o[0] = pcx - 0.5f * pw;
    o[1] = pcy - 0.5f * ph;
    o[2] = pcx + 0.5f * pw;
    o[3] = pcy + 0.5f * ph;
}

__global__ void grid_anchors_kernel(const float* cell, int A, int H, int W, float stride, float offset, float* out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W * A) return;
    const int a = idx % A, x = (idx / A) % W, y = idx / (A * W);
    const float sx = offset * stride + (float)x * stride, sy = offset * stride + (float)y * stride;   // arange(offset*stride, ..., step=stride)
    float* o = out + (size_t)idx * 4;
    o[0] = sx + cell[a * 4 + 0];
    o[1] = sy + cell[a * 4 + 1];
    o[2] = sx + cell[a * 4 + 2];
    o[3] = sy + cell[a * 4 + 3];
}
}  // namespace

extern "C" int pe_box2box_apply_deltas(const float* deltas, const float* boxes, int32_t N, int32_t k, const float* weights_host,
                                       float scale_clamp, float* out, void* stream) {
    PE_CHECK_ARG(N >= 0 && k >= 1 && weights_host, "pe_box2box_apply_deltas: bad args");
    if (N == 0) return PE_OK;
    PE_CHECK_ARG(deltas && boxes && out, "pe_box2box_apply_deltas: null pointer");
    hipLaunchKernelGGL(apply_deltas_kernel, dim3(pe::ceil_div((long long)N * k, 256)), dim3(256), 0, (hipStream_t)stream, deltas, boxes, N, k,
                       weights_host[0], weights_host[1], weights_host[2], weights_host[3], scale_clamp, out);
    PE_CHECK_LAUNCH("pe_box2box_apply_deltas");
    return PE_OK;
}

extern "C" int pe_grid_anchors(const float* cell_anchors, int32_t num_cell_anchors, int32_t H, int32_t W, int32_t stride,
                               float offset, float* out, void* stream) {
    PE_CHECK_ARG(cell_anchors && out && num_cell_anchors >= 1 && H >= 0 && W >= 0 && stride > 0, "pe_grid_anchors: bad args");
    if (H * W == 0) return PE_OK;
    hipLaunchKernelGGL(grid_anchors_kernel, dim3(pe::ceil_div((long long)H * W * num_cell_anchors, 256)), dim3(256), 0,
                       (hipStream_t)stream, cell_anchors, num_cell_anchors, H, W, (float)stride, offset, out);
    PE_CHECK_LAUNCH("pe_grid_anchors");
    return PE_OK;
}
