// Shared definitions of the LDS-DMA convolution kernels (conv_igemm2.hip: kernels and dispatch).  Internal linkage: every translation
// unit gets its own copy of the helpers and of the 16-byte zero page.
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"

namespace pe {
struct Conv2Args {
    const _Float16* in;
    const _Float16* wgt;
    const float* bias;
    const _Float16* res;
    void* out;
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int stride;
    int M, K;
    int relu, res_mode;
    int resH, resW;
    int out_f32, cout_store, out_stride;
    int tiles_m, tiles_n;
};
}  // namespace pe
using pe::Conv2Args;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int BK = 64;       // halfs per K-step = one 128-byte LDS row
constexpr int ROW_B = 128;   // bytes per LDS row
constexpr int MODE_1X1 = 0, MODE_3X3 = 1;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0, 0, 0, 0};


typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// The 8 bias values a thread needs in EVERY pass of the epilogue below (its channel group (tid % (BN / 8)) * 8 never changes), loaded
// ONCE, before the K-loop: a vector load issued inside the epilogue queues behind the kernel's own DMA stream (round 5, the lesson of
// csrc/conv1x1_ring.hip) and was exposed once per 64-row pass.
struct Bias8 { float v[8]; };
template <int BN>
__device__ __forceinline__ Bias8 preload_bias8(const Conv2Args& a, int n0, int tid) {
    Bias8 b;
    const int c = n0 + (tid % (BN / 8)) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) b.v[e] = 0.f;
    if (a.bias) {
        if (c + 8 <= a.Cout) {
            const float4v b0 = *reinterpret_cast<const float4v*>(a.bias + c);
            const float4v b1 = *reinterpret_cast<const float4v*>(a.bias + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { b.v[e] = b0[e]; b.v[e + 4] = b1[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) b.v[e] = (c + e < a.Cout) ? a.bias[c + e] : 0.f;
        }
    }
    return b;
}

// The residual vectors of one 64-row epilogue pass (the thread's NV 16-byte pieces; zeros where there is nothing to add).  Pass 0's are
// requested BEFORE the K-loop and pass p + 1's at the top of pass p, so that the HBM latency of the residual - which is part of the
// layer's compulsory traffic and, issued inside the pass, was exposed once per pass - runs under the K-loop / the previous pass.
template <int BN, int THREADS>
struct ResVecs { half8 v[64 * (BN / 8) / THREADS]; };
template <int BN, int THREADS>
__device__ __forceinline__ ResVecs<BN, THREADS> load_res(const Conv2Args& a, int m0, int n0, int tid, int pass) {
    constexpr int VEC_PER_ROW = BN / 8, NV = 64 * VEC_PER_ROW / THREADS;
    ResVecs<BN, THREADS> out;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        out.v[i] = zero8;
        if (a.res_mode) {
            const int v = tid + i * THREADS;
            const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
            const int m = m0 + pass * 64 + r, c = n0 + c8;
            if (m < a.M && c < a.cout_store) {
                size_t ro;
                if (a.res_mode == 1) {
                    ro = (size_t)m * a.Cout + c;
                } else {
                    const int ow = m % a.Wo, t = m / a.Wo;
                    const int oh = t % a.Ho, n = t / a.Ho;
                    ro = (((size_t)n * a.resH + (oh >> 1)) * a.resW + (ow >> 1)) * a.Cout + c;
                }
                out.v[i] = *reinterpret_cast<const half8*>(a.res + ro);
            }
        }
    }
    return out;
}

// ---- shared epilogue: passes of 64 rows through LDS (fp32), vectorised bias / residual / ReLU / store ----
// Wave (wm, wn) owns rows [64*wm, 64*wm+64) x columns [WN*wn, WN*wn+WN) of the block tile.
template <int BM, int BN, int THREADS = 256>
__device__ __forceinline__ void epilogue(const Conv2Args& a, float16v (&acc)[2][BN / 64], unsigned char* smem, int m0,
                                         int n0, int tid, int lane, int wm, int wn, const Bias8& pre, const ResVecs<BN, THREADS>* res0) {
    constexpr int WN = BN / 2;
    constexpr int TM = 2, TN = BN / 64;
    constexpr int EP_ROWS = 64;
    constexpr int PASSES = BM / 64;
    constexpr int EP_ROW = BN + 4;
    constexpr int VEC_PER_ROW = BN / 8;
    constexpr int NV = EP_ROWS * VEC_PER_ROW / THREADS;
    static_assert(NV >= 1, "epilogue needs at least one vector per thread");
    float* ep = reinterpret_cast<float*>(smem);
    ResVecs<BN, THREADS> rbuf[2];
    rbuf[0] = res0 ? *res0 : load_res<BN, THREADS>(a, m0, n0, tid, 0);      // (a kernel at its register limit leaves pass 0 to the epilogue)
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass + 1 < PASSES) rbuf[(pass + 1) & 1] = load_res<BN, THREADS>(a, m0, n0, tid, pass + 1);
        const half8* rres = rbuf[pass & 1].v;
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        const int c = wn * WN + j * 32 + (lane & 31);
                        ep[r * EP_ROW + c] = acc[i][j][e];
                    }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
            const int m = m0 + pass * EP_ROWS + r, c = n0 + c8;
            if (m >= a.M || c >= a.cout_store) continue;
            const float4v x0 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8);
            const float4v x1 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8 + 4);
            float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            if (a.bias) {      // (x + 0.f would turn -0.f into +0.f: a launch without a bias adds nothing)
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += pre.v[e];
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
        __syncthreads();
    }
}

// ---- epilogue of the 256 x 256 kernels (8 waves as 2 x 4, acc[4][2] of 32 x 32 tiles per wave): 4 passes of
// 64 rows through LDS (fp32), vectorised bias / residual / ReLU / fp16 store ----
__device__ __forceinline__ void epilogue256(const Conv2Args& a, float16v (&acc)[4][2], unsigned char* smem, int m0, int n0,
                                            int tid, int lane, int wm, int wn) {
    constexpr int BN = 256, THREADS = 512, WN = 64, TN = 2;
    // ---- epilogue: 4 passes of 64 rows; pass p is owned by the waves with wm == p >> 1 (their row tiles 2*(p&1), +1) ----
    constexpr int EP_ROW = BN + 4;
    constexpr int VEC_PER_ROW = BN / 8;
    constexpr int NV = 64 * VEC_PER_ROW / THREADS;  // 4
    float* ep = reinterpret_cast<float*>(smem);
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        half8 rres[NV];
        if (a.res_mode) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int v = tid + i * THREADS;
                const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
                const int m = m0 + pass * 64 + r, c = n0 + c8;
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
        if (wm == (pass >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = ii * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        const int c = wn * WN + j * 32 + (lane & 31);
                        ep[r * EP_ROW + c] = acc[2 * (pass & 1) + ii][j][e];
                    }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + i * THREADS;
            const int r = v / VEC_PER_ROW, c8 = (v - r * VEC_PER_ROW) * 8;
            const int m = m0 + pass * 64 + r, c = n0 + c8;
            if (m >= a.M || c >= a.cout_store) continue;
            const float4v x0 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8);
            const float4v x1 = *reinterpret_cast<const float4v*>(ep + r * EP_ROW + c8 + 4);
            float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            if (a.bias) {
                const float4v b0 = *reinterpret_cast<const float4v*>(a.bias + c);
                const float4v b1 = *reinterpret_cast<const float4v*>(a.bias + c + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] += b0[e]; x[e + 4] += b1[e]; }
            }
            if (a.res_mode) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += (float)rres[i][e];
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = pe::relu_nan(x[e]);
            }
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)x[e];
            *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(a.out) + (size_t)m * a.out_stride + c) = h;
        }
        __syncthreads();
    }
}

}  // namespace

