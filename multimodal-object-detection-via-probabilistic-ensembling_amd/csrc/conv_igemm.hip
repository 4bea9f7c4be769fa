// Implicit-GEMM (im2col-free) convolution / GEMM for gfx950 with MFMA 32x32x16 f16.
//
// Replaces what the reference runs as three cuDNN/ATen kernels per layer - conv2d, FrozenBatchNorm2d
// (layers/batch_norm.py:45-65), relu_ (+ the residual add of backbone/resnet.py:205-221 and the
// nearest-2x top-down add of backbone/fpn.py:129-137) - by ONE kernel: BN is folded into the fp16
// weights + an fp32 bias at load time, and bias / residual / top-down add / ReLU run in the epilogue.
// The same kernel is the FC GEMM of the box head (roi_heads/box_head.py:73-81) with H = W = 1.
//
// Layout: activations NHWC fp16, weights [Cout][KH][KW][Cin] fp16 (K contiguous), fp32 accumulate.
//   GEMM view: D[M = N*Ho*Wo, Cout] = A[M, K = KH*KW*Cin] * B^T[Cout, K];  A is gathered on the fly.
// Tiling: block = 256 threads = 4 waves (2x2), block tile BM x BN x 64, wave tile (BM/2) x (BN/2) as
//   TMxTN MFMA 32x32x16 tiles (fp32 accumulators in VGPR/AGPR).  Global -> registers -> LDS staging,
//   double-buffered LDS, ONE barrier per K-step, next K-step's global loads in flight under the MFMAs.
//   LDS rows are padded 128 B -> 144 B so the ds_read_b128 fragment reads are bank-conflict-free.
// Epilogue: accumulators -> LDS (fp32) -> each thread owns 8 consecutive channels of a pixel: bias,
//   residual (16-B loads), ReLU, one 16-B fp16 store (or fp32 stores for the small heads).
// XCD-aware: block ids are remapped so that the n-tiles of one m-tile (same A rows) and neighbouring
//   m-tiles (3x3 halo) sit on one XCD's L2.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int BK = 64;              // K-step (halfs)
constexpr int LDS_ROW = BK + 8;     // padded row: 144 bytes
constexpr int MODE_1X1 = 0, MODE_3X3 = 1, MODE_STEM = 2;

struct ConvArgs {
    const _Float16* in;
    const _Float16* wgt;
    const float* bias;
    const _Float16* res;
    void* out;
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int stride;
    int M, K;
    int relu, res_mode;  // res_mode: 0 none, 1 same-shape add, 2 nearest-2x upsampled add (res is [N,resH,resW,Cout])
    int resH, resW;
    int out_f32, cout_store, out_stride;
    int tiles_m, tiles_n;
};

template <int BM, int BN, int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2;    // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;  // MFMA tiles per wave
    constexpr int A_LOADS = BM * 8 / 256;      // 16-byte chunks per thread per K-step
    constexpr int B_LOADS = BN * 8 / 256;
    constexpr int A_TILE = BM * LDS_ROW, B_TILE = BN * LDS_ROW;  // halfs
    constexpr int EP_ROW = BN + 4;                               // floats
    static_assert((size_t)BM * EP_ROW * 4 <= (size_t)2 * (A_TILE + B_TILE) * 2, "epilogue tile must fit");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* lds = reinterpret_cast<_Float16*>(smem);

    // ---- XCD-aware block remap (bijective for any grid size) ----
    const int nwg = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int seg = tid & 7;        // 16-byte chunk within the 128-byte K-step row
    const int row0 = tid >> 3;      // first of this thread's rows (stride 32)

    // ---- per-thread A-row descriptors (fixed across the K loop) ----
    const _Float16* a_base[A_LOADS];
    int a_oh[A_LOADS], a_ow[A_LOADS];
    bool a_ok[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        const int m = m0 + row0 + 32 * i;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        const int ow = mm % a.Wo, t = mm / a.Wo;
        const int oh = t % a.Ho, n = t / a.Ho;
        a_oh[i] = oh * a.stride;
        a_ow[i] = ow * a.stride;
        a_base[i] = a.in + (size_t)n * a.H * a.W * a.Cin;
    }
    const _Float16* b_base[B_LOADS];
    bool b_ok[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
        const int n = n0 + row0 + 32 * i;
        b_ok[i] = n < a.Cout;
        b_base[i] = a.wgt + (size_t)(b_ok[i] ? n : 0) * a.K + seg * 8;
    }

    half8 areg[A_LOADS], breg[B_LOADS];
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        if (MODE == MODE_1X1) {
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const _Float16* p = a_base[i] + ((size_t)a_oh[i] * a.W + a_ow[i]) * a.Cin + k0 + seg * 8;
                areg[i] = a_ok[i] ? *reinterpret_cast<const half8*>(p) : zero8;
            }
        } else if (MODE == MODE_3X3) {
            const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;  // block-uniform: Cin % 64 == 0
            const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int ih = a_oh[i] + kh - 1, iw = a_ow[i] + kw - 1;
                const bool ok = a_ok[i] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                const _Float16* p = a_base[i] + ((size_t)ih * a.W + iw) * a.Cin + c0 + seg * 8;
                areg[i] = ok ? *reinterpret_cast<const half8*>(p) : zero8;
            }
        } else {  // MODE_STEM: 7x7 stride 2 pad 3 over NHWC4; K = 8 (kh, last is zero-weight) x 8 pixels x 4 channels
            const int kh = kt * 2 + (seg >> 2), px = (seg & 3) * 2;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i) {
                const int ih = a_oh[i] + kh - 3, iw = a_ow[i] + px - 3;
                const bool okr = a_ok[i] && kh < 7 && (unsigned)ih < (unsigned)a.H;
                const _Float16* p = a_base[i] + ((size_t)ih * a.W + iw) * 4;
                const half4 z4 = {0, 0, 0, 0};
                const half4 lo = (okr && (unsigned)iw < (unsigned)a.W) ? *reinterpret_cast<const half4*>(p) : z4;
                const half4 hi = (okr && (unsigned)(iw + 1) < (unsigned)a.W) ? *reinterpret_cast<const half4*>(p + 4) : z4;
                areg[i] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            breg[i] = b_ok[i] ? *reinterpret_cast<const half8*>(b_base[i] + k0) : zero8;
    };
    auto store_tiles = [&](int buf) {
        _Float16* la = lds + buf * (A_TILE + B_TILE);
        _Float16* lb = la + A_TILE;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            *reinterpret_cast<half8*>(la + (row0 + 32 * i) * LDS_ROW + seg * 8) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
            *reinterpret_cast<half8*>(lb + (row0 + 32 * i) * LDS_ROW + seg * 8) = breg[i];
    };

    float16v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = a.K / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);  // in flight under the MFMAs below
        const _Float16* la = lds + buf * (A_TILE + B_TILE) + (wm * WM + frow) * LDS_ROW + fk;
        const _Float16* lb = lds + buf * (A_TILE + B_TILE) + A_TILE + (wn * WN + frow) * LDS_ROW + fk;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            half8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const half8*>(la + i * 32 * LDS_ROW + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const half8*>(lb + j * 32 * LDS_ROW + ks * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS (fp32) -> vectorised bias / residual / ReLU / store ----
    // Each thread owns NV fixed (row, 8-channel) vectors; their residual operands are fetched FIRST so the
    // HBM latency overlaps the accumulator -> LDS transposition instead of serialising per vector.
    constexpr int VEC_PER_ROW = BN / 8;
    constexpr int NV = BM * VEC_PER_ROW / 256;
    half8 rres[NV];
    if (a.res_mode) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * 256;
            const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
            const int m = m0 + r, c = n0 + c8;
            rres[i] = zero8;
            if (m < a.M && c < a.cout_store) {
                size_t ro;
                if (a.res_mode == 1) {
                    ro = (size_t)m * a.Cout + c;
                } else {
                    const int ow = m % a.Wo, t = m / a.Wo;
                    const int oh = t % a.Ho, n = t / a.Ho;
                    ro = (((size_t)n * a.resH + (oh >> 1)) * a.resW + (ow >> 1)) * a.Cout + c;
                }
                rres[i] = *reinterpret_cast<const half8*>(a.res + ro);
            }
        }
    }
    float* ep = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int c = wn * WN + j * 32 + (lane & 31);
                ep[r * EP_ROW + c] = acc[i][j][e];
            }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
        const int m = m0 + r, c = n0 + c8;
        if (m >= a.M || c >= a.cout_store) continue;
        const float4v x0 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8);
        const float4v x1 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8 + 4);
        float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (a.bias) {
            if (c + 8 <= a.Cout) {
                const float4v b0 = *reinterpret_cast<const float4v*>(a.bias + c);
                const float4v b1 = *reinterpret_cast<const float4v*>(a.bias + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] += b0[e]; x[e + 4] += b1[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += (c + e < a.Cout) ? a.bias[c + e] : 0.f;
            }
        }
        if (a.res_mode) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += (float)rres[i][e];
        }
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = pe::relu_nan(x[e]);
        }
        if (a.out_f32) {
            float* o = reinterpret_cast<float*>(a.out) + (size_t)m * a.out_stride + c;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c + e < a.cout_store) o[e] = x[e];
        } else {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)x[e];
            *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(a.out) + (size_t)m * a.out_stride + c) = h;
        }
    }
}

template <int BM, int BN, int MODE>
int launch(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    a.tiles_m = pe::ceil_div(a.M, BM);
    a.tiles_n = pe::ceil_div(a.Cout, BN);
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDS_ROW * 2;
    PE_ENSURE_LDS((conv_igemm_kernel<BM, BN, MODE>), lds, "pe_conv2d_nhwc_f16");
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, MODE>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, st, a);
    PE_CHECK_LAUNCH("pe_conv2d_nhwc_f16");
    return PE_OK;
}

}  // namespace

namespace pe {
int conv2_dispatch(const void* in, const void* wgt, const float* bias, const void* res, void* out, int N, int H, int W,
                   int Cin, int Cout, int Ho, int Wo, int K, int M, int mode3x3, int stride, int relu, int res_mode,
                   int resH, int resW, int out_f32, int cout_store, int out_stride, hipStream_t st);
}  // namespace pe

extern "C" int pe_conv2d_nhwc_f16(const void* input, const void* weight, const float* bias, const void* residual,
                                  void* output, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                  int32_t kernel, int32_t stride, int32_t relu, int32_t residual_mode,
                                  int32_t res_h, int32_t res_w, int32_t out_f32, int32_t cout_store,
                                  int32_t out_stride, void* stream) {
    PE_CHECK_ARG(input && weight && output, "pe_conv2d_nhwc_f16: null pointer");
    PE_CHECK_ARG(kernel == 1 || kernel == 3 || kernel == 7, "pe_conv2d_nhwc_f16: kernel %d not in {1,3,7}", kernel);
    PE_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "pe_conv2d_nhwc_f16: bad dims");
    PE_CHECK_ARG(residual_mode == 0 || residual, "pe_conv2d_nhwc_f16: residual_mode set but residual is null");
    ConvArgs a{};
    a.in = (const _Float16*)input; a.wgt = (const _Float16*)weight; a.bias = bias; a.res = (const _Float16*)residual;
    a.out = output; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.stride = stride;
    a.relu = relu; a.res_mode = residual_mode; a.resH = res_h; a.resW = res_w;
    a.out_f32 = out_f32; a.cout_store = cout_store > 0 ? cout_store : Cout; a.out_stride = out_stride > 0 ? out_stride : Cout;
    int mode;
    if (kernel == 1) {
        PE_CHECK_ARG(stride == 1 || stride == 2, "pe_conv2d_nhwc_f16: 1x1 stride %d", stride);
        PE_CHECK_ARG(Cin % 64 == 0, "pe_conv2d_nhwc_f16: 1x1 needs Cin %% 64 == 0 (got %d)", Cin);
        a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1; a.K = Cin; mode = MODE_1X1;
    } else if (kernel == 3) {
        PE_CHECK_ARG(stride == 1, "pe_conv2d_nhwc_f16: 3x3 supports stride 1 (pad 1) only");
        PE_CHECK_ARG(Cin % 64 == 0, "pe_conv2d_nhwc_f16: 3x3 needs Cin %% 64 == 0 (got %d)", Cin);
        a.Ho = H; a.Wo = W; a.K = 9 * Cin; mode = MODE_3X3;
    } else {
        PE_CHECK_ARG(stride == 2 && Cin == 4, "pe_conv2d_nhwc_f16: 7x7 is the stem: stride 2, NHWC4 input");
        a.Ho = (H + 6 - 7) / 2 + 1; a.Wo = (W + 6 - 7) / 2 + 1; a.K = 256; mode = MODE_STEM;
    }
    PE_CHECK_ARG(out_f32 || (a.cout_store % 8 == 0 && a.out_stride % 8 == 0),
                 "pe_conv2d_nhwc_f16: fp16 output needs channel counts that are multiples of 8");
    PE_CHECK_ARG(residual_mode == 0 || (Cout % 8 == 0 && !out_f32), "pe_conv2d_nhwc_f16: residual needs fp16 out, Cout %% 8 == 0");
    const long long M = (long long)N * a.Ho * a.Wo;
    PE_CHECK_ARG(M < (1ll << 31), "pe_conv2d_nhwc_f16: M too large");
    a.M = (int)M;
    hipStream_t st = (hipStream_t)stream;
    if (mode != MODE_STEM)
        return pe::conv2_dispatch(input, weight, bias, residual, output, N, H, W, Cin, Cout, a.Ho, a.Wo, a.K, a.M,
                                  mode == MODE_3X3, stride, relu, residual_mode, res_h, res_w, out_f32, a.cout_store,
                                  a.out_stride, st);
    // the unfused 7x7 stem (inputs whose H or W is not a multiple of 4; otherwise stem.hip's one-pass kernel runs): the only
    // launch left on this file's register-staged kernel
    return launch<128, 64, MODE_STEM>(a, st);
}
