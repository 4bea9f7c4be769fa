// Stand-alone probe of the weights-direct convolution kernels (csrc/conv_wd.h) + the MFMA yardstick.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I scripts -I multimodal-object-detection-via-probabilistic-ensembling_amd/csrc \
//         scripts/conv_wd_probe.hip -o scripts/conv_wd_probe && scripts/conv_wd_probe
#include "conv_wd.h"
#include "lab/conv_wd_1x1.h"

#include <cmath>
#include <cstdlib>
#include <random>
#include <vector>

namespace pe {
void set_error(const char*, ...) {}
}

using wd::half8;
using wd::float16v;

// ---- MFMA yardstick: 4 waves per CU (one per SIMD), 8 independent accumulators, operands from memory ----
__global__ __launch_bounds__(256, 1) void mfma_peak(const _Float16* src, float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    half8 a = *reinterpret_cast<const half8*>(src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 16);
    half8 b = *reinterpret_cast<const half8*>(src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 16 + 8);
    float16v acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
    if (s == 12345.678f) sink[0] = s;
}

// ---- yardstick 2: what the MFMA pipe sustains with REAL operand traffic and nothing else ----
// 2 blocks of 4 waves per CU (2 waves per SIMD, like the conv kernel).  Per iteration 8 MFMAs on 8 accumulators.
//   CYCLE: the A / B operands rotate through 8 register sets each (operand toggling) instead of staying constant
//   LDSR : 4 ds_read_b128 per iteration (the conv kernel's 0.5 KiB per MFMA), software-pipelined one iteration ahead
//   USEL : the MFMA B operands ARE the LDS data (else the reads are only kept alive)
template <int CYCLE, int LDSR, int USEL>
__global__ __launch_bounds__(256, 2) void mfma_y2(const _Float16* wsrc, const _Float16* xsrc, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 144];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 256 * 8; i += 256)
        *reinterpret_cast<half8*>(lds + (i >> 3) * 144 + (i & 7) * 16) = *reinterpret_cast<const half8*>(xsrc + ((size_t)(blockIdx.x & 255) * 2048 + i) * 8);
    __syncthreads();
    half8 A[8], B[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        A[j] = *reinterpret_cast<const half8*>(wsrc + ((size_t)(blockIdx.x & 255) * 2048 + j * 256 + tid) * 8);
        B[j] = *reinterpret_cast<const half8*>(xsrc + ((size_t)(blockIdx.x & 255) * 2048 + j * 256 + tid) * 8);
    }
    float16v acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned char* base = lds + (wave * 32 + (lane & 31)) * 144 + (lane >> 5) * 16;
    half8 r[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[0][j] = *reinterpret_cast<const half8*>(base + j * 32);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (LDSR) {
                const int e = ((it + h) & 15) * 144;   // 16 different entry offsets -> changing data
#pragma unroll
                for (int j = 0; j < 4; ++j) r[h ^ 1][j] = *reinterpret_cast<const half8*>(base + e + (j + (h ? 0 : 4)) * 16 * 0 + j * 32);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const half8 a = A[CYCLE ? i : 0];
                const half8 b = USEL ? r[h][i & 3] : B[CYCLE ? (i * 3) & 7 : 0];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
            if (LDSR && !USEL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(r[h ^ 1][j]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][lane & 15];
    if (s == 12345.678f) sink[0] = s;
}

template <int CYCLE, int LDSR, int USEL>
void run_y2(const char* name, const _Float16* w, const _Float16* x, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512 * 4, iters = 8000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_y2<CYCLE, LDSR, USEL>), dim3(blocks), dim3(256), 0, 0, w, x, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * 32 * 32 * 16 * 8.0 * iters * 4 * blocks / (ms * 1e-3) / 1e12;
    printf("yardstick2 %-58s: %8.3f ms %8.0f TFLOP/s\n", name, ms, tf);
    fflush(stdout);
}

static float time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventElapsedTime(&ms, e0, e1); return ms; }

template <int WM, int WN, int TPX, int DEPTH, int ABL = 0>
void run_variant(const char* name, pe::ConvWdArgs a, const std::vector<_Float16>& hin, const std::vector<_Float16>& hw,
                 const std::vector<float>& hb, std::vector<_Float16>& hout, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(a.out, 0xff, (size_t)a.M * a.out_stride * 2);
    int st = wd::launch_conv3x3_wd<WM, WN, TPX, DEPTH, ABL>(a, 0);
    hipError_t err = hipDeviceSynchronize();
    if (st != 0 || err != hipSuccess) { printf("%-22s unsupported/failed (%d, %s)\n", name, st, hipGetErrorString(err)); return; }
    hipMemcpy(hout.data(), a.out, hout.size() * 2, hipMemcpyDeviceToHost);
    // sampled check against a double-precision host reference
    std::mt19937 rng(7);
    double max_err = 0, max_ref = 0; int bad = 0;
    const int K = 9 * a.Cin;
    for (int s = 0; s < 3000; ++s) {
        int m, c;
        if (s < 600) {   // image borders / tile seams
            const int n = rng() % a.N, hh = (s & 1) ? (rng() % 2 ? 0 : a.H - 1) : rng() % a.H, ww = (s & 2) ? (rng() % 2 ? 0 : a.W - 1) : rng() % a.W;
            m = (n * a.H + hh) * a.W + ww;
        } else m = rng() % a.M;
        c = rng() % a.Cout;
        const int ow = m % a.W, oh = (m / a.W) % a.H, n = m / (a.W * a.H);
        double ref = hb[c];
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh + kh - 1, iw = ow + kw - 1;
                if (ih < 0 || ih >= a.H || iw < 0 || iw >= a.W) continue;
                const _Float16* x = &hin[((size_t)(n * a.H + ih) * a.W + iw) * a.Cin];
                const _Float16* w = &hw[(size_t)c * K + (kh * 3 + kw) * a.Cin];
                for (int ci = 0; ci < a.Cin; ++ci) ref += (double)(float)x[ci] * (double)(float)w[ci];
            }
        if (a.relu && ref < 0) ref = 0;
        const double got = (float)hout[(size_t)m * a.out_stride + c];
        const double e = fabs(got - ref);
        if (e > max_err) max_err = e;
        if (fabs(ref) > max_ref) max_ref = fabs(ref);
        if (e > 2e-2 + 4e-3 * fabs(ref)) ++bad;
    }
    for (int i = 0; i < 3; ++i) wd::launch_conv3x3_wd<WM, WN, TPX, DEPTH, ABL>(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) wd::launch_conv3x3_wd<WM, WN, TPX, DEPTH, ABL>(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    const double ms = time_ms(e0, e1) / reps;
    const double tf = 2.0 * a.M * a.Cout * K / (ms * 1e-3) / 1e12;
    printf("%-22s %8.4f ms %8.1f TFLOP/s   check: %d bad / 3000, max err %.4f (max |ref| %.2f)\n", name, ms, tf, bad, max_err, max_ref);
    fflush(stdout);
}

// ---------------- 1x1 probe ----------------
template <int WM, int WN, int TPX, int DEPTH>
void run_1x1(const char* name, pe::ConvWdArgs a, const std::vector<_Float16>& hin, const std::vector<_Float16>& hw, const std::vector<float>& hb,
             const std::vector<_Float16>& hres, std::vector<_Float16>& hout, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(a.out, 0xff, (size_t)a.M * a.out_stride * 2);
    int st = wd::launch_conv1x1_wd<WM, WN, TPX, DEPTH>(a, 0);
    hipError_t err = hipDeviceSynchronize();
    if (st != 0 || err != hipSuccess) { printf("%-22s failed (%d, %s)\n", name, st, hipGetErrorString(err)); return; }
    hipMemcpy(hout.data(), a.out, hout.size() * 2, hipMemcpyDeviceToHost);
    std::mt19937 rng(9);
    double max_err = 0; int bad = 0;
    for (int s = 0; s < 3000; ++s) {
        const int m = s < 200 ? (s < 100 ? s : a.M - 1 - (s - 100)) : (int)(rng() % a.M);
        const int c = rng() % a.Cout;
        const int ow = m % a.Wo, oh = (m / a.Wo) % a.Ho, n = m / (a.Wo * a.Ho);
        const _Float16* x = &hin[((size_t)(n * a.H + oh * a.stride) * a.W + ow * a.stride) * a.Cin];
        const _Float16* w = &hw[(size_t)c * a.Cin];
        double ref = hb[c];
        for (int ci = 0; ci < a.Cin; ++ci) ref += (double)(float)x[ci] * (double)(float)w[ci];
        if (a.res_mode == 1) ref += (double)(float)hres[(size_t)m * a.Cout + c];
        if (a.res_mode == 2) ref += (double)(float)hres[(((size_t)n * a.resH + (oh >> 1)) * a.resW + (ow >> 1)) * a.Cout + c];
        if (a.relu && ref < 0) ref = 0;
        const double e = fabs((double)(float)hout[(size_t)m * a.out_stride + c] - ref);
        if (e > max_err) max_err = e;
        if (e > 2e-2 + 4e-3 * fabs(ref)) ++bad;
    }
    for (int i = 0; i < 3; ++i) wd::launch_conv1x1_wd<WM, WN, TPX, DEPTH>(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) wd::launch_conv1x1_wd<WM, WN, TPX, DEPTH>(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    const double ms = time_ms(e0, e1) / reps;
    const double bytes = 2.0 * ((double)a.M * a.Cin + (double)a.Cout * a.Cin + (double)a.M * a.Cout + (a.res_mode == 1 ? (double)a.M * a.Cout : (a.res_mode == 2 ? (double)a.N * a.resH * a.resW * a.Cout : 0.0)));
    printf("%-22s %8.4f ms %8.1f TFLOP/s %7.0f GB/s algorithmic   check: %d bad / 3000, max err %.4f\n", name, ms,
           2.0 * a.M * a.Cout * a.Cin / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9, bad, max_err);
    fflush(stdout);
}

void probe_1x1() {
    struct S { int N, H, W, Cin, Cout, stride, res_mode, relu; const char* what; };
    const S shapes[] = {{3, 10, 12, 128, 256, 1, 1, 1, "tiny (tail + res)"},
                        {32, 50, 64, 256, 1024, 1, 1, 1, "res4 conv3 + residual"}, {32, 50, 64, 256, 1024, 1, 0, 1, "res4 conv3 WITHOUT residual"}, {32, 50, 64, 1024, 256, 1, 0, 1, "res4 conv1"},
                        {32, 200, 256, 64, 256, 1, 1, 1, "res2 conv3 + residual"}, {32, 100, 128, 128, 512, 1, 1, 1, "res3 conv3 + residual"},
                        {32, 100, 128, 512, 256, 1, 2, 0, "fpn lateral3 + top-down"}, {32, 100, 128, 512, 1024, 2, 0, 0, "res4 shortcut (stride 2)"},
                        {32, 25, 32, 2048, 512, 1, 0, 1, "res5 conv1"}, {32000, 1, 1, 1024, 1024, 1, 0, 1, "fc2"}, {32000, 1, 1, 12544, 1024, 1, 0, 1, "fc1"},
                        {32, 25, 32, 512, 2048, 1, 1, 1, "res5 conv3 + residual"}};
    for (const S& s : shapes) {
        const int Ho = (s.H - 1) / s.stride + 1, Wo = (s.W - 1) / s.stride + 1, M = s.N * Ho * Wo;
        const int rH = (Ho + 1) / 2, rW = (Wo + 1) / 2;
        std::vector<_Float16> hin((size_t)s.N * s.H * s.W * s.Cin), hw((size_t)s.Cout * s.Cin), hout((size_t)M * s.Cout);
        std::vector<_Float16> hres(s.res_mode == 1 ? (size_t)M * s.Cout : (s.res_mode == 2 ? (size_t)s.N * rH * rW * s.Cout : 1));
        std::vector<float> hb(s.Cout);
        std::mt19937 rng(21);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto& v : hin) { const float x = nd(rng); v = (_Float16)(x > 0 ? x : 0.f); }
        for (auto& v : hw) v = (_Float16)(nd(rng) / sqrtf((float)s.Cin));
        for (auto& v : hres) v = (_Float16)nd(rng);
        for (auto& v : hb) v = nd(rng) * 0.1f;
        _Float16 *din, *dw, *dwp, *dout, *dres; float* db;
        hipMalloc(&din, hin.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dwp, hw.size() * 2);
        hipMalloc(&dout, hout.size() * 2); hipMalloc(&dres, hres.size() * 2); hipMalloc(&db, hb.size() * 4);
        hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dres, hres.data(), hres.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        const long long total = (long long)(s.Cout / 32) * (s.Cin / 16) * 64;
        hipLaunchKernelGGL(wd::pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dw, dwp, s.Cout, s.Cin, s.Cin, 4, 0);
        pe::ConvWdArgs a{};
        a.in = din; a.wpk = dwp; a.bias = db; a.res = s.res_mode ? dres : nullptr; a.out = dout; a.N = s.N; a.H = s.H; a.W = s.W;
        a.Cin = s.Cin; a.Cout = s.Cout; a.M = M; a.relu = s.relu; a.out_stride = s.Cout; a.stride = s.stride; a.Ho = Ho; a.Wo = Wo;
        a.res_mode = s.res_mode; a.resH = rH; a.resW = rW;
        printf("--- 1x1 %s: N%d %dx%d %d->%d s%d res%d\n", s.what, s.N, s.H, s.W, s.Cin, s.Cout, s.stride, s.res_mode);
        run_1x1<1, 4, 4, 4>("wd1x1<1,4,tpx4>", a, hin, hw, hb, hres, hout, 20);
        hipFree(din); hipFree(dw); hipFree(dwp); hipFree(dout); hipFree(dres); hipFree(db);
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == -1) { probe_1x1(); return 0; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // ---------------- yardstick ----------------
    {
        const int blocks = 256 * 4, iters = 20000;
        std::vector<_Float16> h((size_t)blocks * 256 * 16);
        _Float16* src; float* sink;
        hipMalloc(&src, h.size() * 2); hipMalloc(&sink, 4);
        std::mt19937 rng(1);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (int fill = 0; fill < 3; ++fill) {
            for (auto& v : h) v = (_Float16)(fill == 0 ? 0.f : (fill == 1 ? nd(rng) : fabsf(nd(rng)) * (rng() % 2)));
            hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, src, sink, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            const double ms = time_ms(e0, e1);
            const double tf = 2.0 * 32 * 32 * 16 * 8.0 * iters * 4 * blocks / (ms * 1e-3) / 1e12;
            printf("mfma_f32_32x32x16_f16 yardstick, operands %-28s: %8.3f ms %8.0f TFLOP/s (= %.2f GHz x 256 CU x 4096 flop/clk)\n",
                   fill == 0 ? "all zero" : (fill == 1 ? "N(0,1)" : "half zero / |N(0,1)| (post-ReLU)"), ms, tf, tf * 1e12 / (256.0 * 4096) / 1e9);
        }
        hipFree(src); hipFree(sink);
    }
    if (argc > 4) {   // yardstick 2 (operand toggling / LDS traffic), weights N(0, 1/48), activations post-ReLU
        std::vector<_Float16> hw2((size_t)256 * 2048 * 8), hx2((size_t)256 * 2048 * 8);
        std::mt19937 rng(11);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto& v : hw2) v = (_Float16)(nd(rng) / 48.f);
        for (auto& v : hx2) { const float x = nd(rng); v = (_Float16)(x > 0 ? x : 0.f); }
        _Float16 *dw2, *dx2; float* sink2;
        hipMalloc(&dw2, hw2.size() * 2); hipMalloc(&dx2, hx2.size() * 2); hipMalloc(&sink2, 4);
        hipMemcpy(dw2, hw2.data(), hw2.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dx2, hx2.data(), hx2.size() * 2, hipMemcpyHostToDevice);
        run_y2<0, 0, 0>("constant operands, no LDS traffic", dw2, dx2, sink2);
        run_y2<1, 0, 0>("operands rotate over 8 register sets, no LDS traffic", dw2, dx2, sink2);
        run_y2<0, 1, 0>("constant operands + 4 ds_read_b128 / 8 MFMA (unused)", dw2, dx2, sink2);
        run_y2<1, 1, 0>("rotating operands + 4 ds_read_b128 / 8 MFMA (unused)", dw2, dx2, sink2);
        run_y2<1, 1, 1>("rotating A, B operands read from LDS (4 / 8 MFMA)", dw2, dx2, sink2);
        hipFree(dw2); hipFree(dx2); hipFree(sink2);
        if (atoi(argv[4]) == 2) return 0;
    }
    // ---------------- 3x3 convolution ----------------
    struct Shape { int N, H, W, Cin, Cout; };
    const Shape shapes[] = {{2, 16, 64, 64, 256}, {32, 50, 64, 256, 256}, {32, 200, 256, 256, 256}, {32, 100, 128, 256, 256}, {32, 25, 32, 256, 256}};
    const int first = argc > 1 ? atoi(argv[1]) : 0, last = argc > 2 ? atoi(argv[2]) : 4, mode = argc > 3 ? atoi(argv[3]) : 0;
    for (int si = first; si <= last; ++si) {
        const Shape s = shapes[si];
        const int M = s.N * s.H * s.W, K = 9 * s.Cin;
        std::vector<_Float16> hin((size_t)M * s.Cin), hw((size_t)s.Cout * K), hout((size_t)M * s.Cout);
        std::vector<float> hb(s.Cout);
        std::mt19937 rng(3 + si);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto& v : hin) { const float x = nd(rng); v = (_Float16)(x > 0 ? x : 0.f); }   // post-ReLU activations
        const float wsc = 1.f / sqrtf((float)K);
        for (auto& v : hw) v = (_Float16)(nd(rng) * wsc);
        for (auto& v : hb) v = nd(rng) * 0.1f;
        _Float16 *din, *dw, *dwp, *dout; float* db;
        hipMalloc(&din, hin.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dwp, hw.size() * 2);
        hipMalloc(&dout, hout.size() * 2); hipMalloc(&db, hb.size() * 4);
        hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        pe::ConvWdArgs a{};
        a.in = din; a.bias = db; a.res = nullptr; a.out = dout; a.N = s.N; a.H = s.H; a.W = s.W; a.Cin = s.Cin; a.Cout = s.Cout;
        a.M = M; a.relu = 1; a.out_stride = s.Cout;
        printf("--- 3x3 N%d %dx%d %d->%d (%.1f GFLOP)\n", s.N, s.H, s.W, s.Cin, s.Cout, 2.0 * M * s.Cout * K / 1e9);
        const int reps = si == 2 ? 5 : 20;
        for (int WN : {4, 2}) {
            const long long total = (long long)(s.Cout / 32) * (K / 16) * 64;
            hipLaunchKernelGGL(wd::pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dw, dwp, s.Cout, K, s.Cin, WN, 1);
            a.wpk = dwp;
            if (WN == 4 && mode == 1) {   // PMC runs: the main variant only
                run_variant<1, 4, 4, 3>("wd<1,4,tpx4,d3>", a, hin, hw, hb, hout, reps);
            } else if (WN == 4) {
                run_variant<1, 4, 4, 4>("wd<1,4,tpx4,d4>", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 4, 3>("wd<1,4,tpx4,d3>", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 4, 3, 1>("  abl: no w loads", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 4, 3, 2>("  abl: no ds_reads", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 4, 3, 4>("  abl: no slab/barrier", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 4, 3, 3>("  abl: no w, no ds", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 4, 3, 7>("  abl: MFMA only", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 4, 2>("wd<1,4,tpx4,d2>", a, hin, hw, hb, hout, reps);
                run_variant<2, 4, 4, 3>("wd<2,4,tpx4,d3>", a, hin, hw, hb, hout, reps);
                run_variant<1, 4, 2, 4>("wd<1,4,tpx2,d4>", a, hin, hw, hb, hout, reps);


            }
        }
        hipFree(din); hipFree(dw); hipFree(dwp); hipFree(dout); hipFree(db);
    }
    return 0;
}
