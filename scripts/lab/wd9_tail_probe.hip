// Development harness for the fused bottleneck tail on the one-wave-per-SIMD structure (csrc/conv_wd9_tail.h): shipped two-wave
// kernel (conv_wd.h, HEAD = 2) vs the new one on res4 shapes - sampled fp64 check, full comparison against the shipped kernel's
// output (same arithmetic up to summation order), timing, ablations and an s_memtime timeline.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-spill-vgpr-to-agpr=0 -I include \
//         -I multimodal-object-detection-via-probabilistic-ensembling_amd/csrc scripts/lab/wd9_tail_probe.hip -o scripts/wd9_tail_probe
#include "conv_wd9_tail.h"

#include <cmath>
#include <cstdlib>
#include <random>
#include <vector>

namespace pe {
void set_error(const char*, ...) {}
int ensure_dynamic_lds(const void* k, size_t bytes, const char*) {
    return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
}  // namespace pe

int main(int argc, char** argv) {
    struct Shape { int N, H, W, Cin, TC; };
    const Shape shapes[] = {{2, 7, 64, 64, 256}, {3, 50, 64, 256, 1024}, {32, 50, 64, 256, 1024}, {1, 50, 64, 256, 1024}};
    const int first = argc > 1 ? atoi(argv[1]) : 0, last = argc > 2 ? atoi(argv[2]) : 3;
    unsigned long long* dbg; hipMalloc(&dbg, 256 * 128 * 8);
    for (int si = first; si <= last; ++si) {
        const Shape s = shapes[si];
        const int M = s.N * s.H * s.W, K = 9 * s.Cin;
        std::vector<_Float16> hin((size_t)M * s.Cin), hw((size_t)256 * K), hw3((size_t)s.TC * 256), hres((size_t)M * s.TC), hout((size_t)M * s.TC), href((size_t)M * s.TC);
        std::vector<float> hb(256), hb3(s.TC);
        std::mt19937 rng(11 + si);
        std::normal_distribution<float> nd(0.f, 1.f);
        for (auto& v : hin) { const float x = nd(rng); v = (_Float16)(x > 0 ? x : 0.f); }
        for (auto& v : hres) { const float x = nd(rng); v = (_Float16)(x > 0 ? x : 0.f); }
        const float wsc = 1.f / sqrtf((float)K);
        for (auto& v : hw) v = (_Float16)(nd(rng) * wsc);
        for (auto& v : hw3) v = (_Float16)(nd(rng) / 16.f);
        for (auto& v : hb) v = nd(rng) * 0.1f;
        for (auto& v : hb3) v = nd(rng) * 0.1f;
        _Float16 *din, *dw, *dwp, *dw3, *dw3p, *dres, *dout; float *db, *db3;
        hipMalloc(&din, hin.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dwp, hw.size() * 2); hipMalloc(&dw3, hw3.size() * 2); hipMalloc(&dw3p, hw3.size() * 2);
        hipMalloc(&dres, hres.size() * 2); hipMalloc(&dout, hout.size() * 2); hipMalloc(&db, 256 * 4); hipMalloc(&db3, s.TC * 4);
        hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dw3, hw3.data(), hw3.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dres, hres.data(), hres.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 256 * 4, hipMemcpyHostToDevice);
        hipMemcpy(db3, hb3.data(), s.TC * 4, hipMemcpyHostToDevice);
        const long long total = (long long)(256 / 32) * (K / 16) * 64;
        hipLaunchKernelGGL(wd::pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dw, dwp, 256, K, s.Cin, 4, 1);
        hipLaunchKernelGGL(wd::pack_tail_kernel, dim3((s.TC * 256 / 8 + 255) / 256), dim3(256), 0, 0, dw3, dw3p, s.TC);
        pe::ConvWdArgs a{};
        a.in = din; a.wpk = dwp; a.bias = db; a.N = s.N; a.H = s.H; a.W = s.W; a.Cin = s.Cin; a.Cout = 256; a.M = M; a.relu = 1; a.out_stride = 256;
        a.tail_w = dw3p; a.tail_b = db3; a.tail_res = dres; a.tail_out = dout; a.tail_cout = s.TC;
        const double gflop = 2.0 * M * 256 * K / 1e9 + 2.0 * M * s.TC * 256 / 1e9;
        printf("--- tail N%d %dx%d %d->256->%d (%.1f GFLOP, %.1f MB algorithmic)\n", s.N, s.H, s.W, s.Cin, s.TC, gflop,
               (M * (s.Cin + 2.0 * s.TC) * 2 + 256.0 * K * 2 + s.TC * 512.0) / 1e6);
        const int reps = 20;
        auto time_it = [&](const char* name, auto launch, bool check, bool is_ref) {
            hipMemset(dout, 0xff, hout.size() * 2);
            const int st = launch();
            hipError_t err = hipDeviceSynchronize();
            if (st != 0 || err != hipSuccess) { printf("%-36s unsupported/failed (%d, %s)\n", name, st, hipGetErrorString(err)); return; }
            int bad = 0; double max_err = 0, max_dref = 0; long long ndiff = 0;
            if (check) {
                hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost);
                std::mt19937 r2(5);
                std::vector<double> t(256);
                for (int smp = 0; smp < 160; ++smp) {
                    int m;
                    if (smp < 60) { const int n = (smp & 4) ? s.N - 1 : r2() % s.N, hh = (smp & 1) ? (r2() % 2 ? 0 : s.H - 1) : r2() % s.H, ww = (smp & 2) ? (r2() % 2 ? 0 : s.W - 1) : r2() % s.W; m = (n * s.H + hh) * s.W + ww; }
                    else m = r2() % M;
                    const int ow = m % s.W, oh = (m / s.W) % s.H, n = m / (s.W * s.H);
                    for (int c = 0; c < 256; ++c) {
                        double acc = hb[c];
                        for (int kh = 0; kh < 3; ++kh)
                            for (int kw = 0; kw < 3; ++kw) {
                                const int ih = oh + kh - 1, iw = ow + kw - 1;
                                if (ih < 0 || ih >= s.H || iw < 0 || iw >= s.W) continue;
                                const _Float16* x = &hin[((size_t)(n * s.H + ih) * s.W + iw) * s.Cin];
                                const _Float16* w = &hw[(size_t)c * K + (kh * 3 + kw) * s.Cin];
                                for (int ci = 0; ci < s.Cin; ++ci) acc += (double)(float)x[ci] * (double)(float)w[ci];
                            }
                        t[c] = (double)(float)(_Float16)(float)(acc > 0 ? acc : 0);
                    }
                    for (int k = 0; k < 24; ++k) {
                        const int o = r2() % s.TC;
                        double ref = hb3[o] + (double)(float)hres[(size_t)m * s.TC + o];
                        for (int c = 0; c < 256; ++c) ref += t[c] * (double)(float)hw3[(size_t)o * 256 + c];
                        if (ref < 0) ref = 0;
                        const double e = fabs((double)(float)hout[(size_t)m * s.TC + o] - ref);
                        max_err = e > max_err ? e : max_err;
                        if (!(e <= 3e-2 + 6e-3 * fabs(ref))) ++bad;
                    }
                }
                if (is_ref) href = hout;
                else for (size_t i = 0; i < hout.size(); ++i) {
                    const double d = fabs((double)(float)hout[i] - (double)(float)href[i]);
                    max_dref = d > max_dref ? d : max_dref;
                    ndiff += d > 0.02 + 0.004 * fabs((double)(float)href[i]);
                }
            }
            for (int i = 0; i < 2; ++i) launch();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            if (check) printf("%-36s %8.4f ms %8.1f TFLOP/s   fp64 check: %d bad / 3840, max err %.4f | vs shipped: max diff %.4f, beyond tolerance %lld\n", name, ms, gflop / ms, bad, max_err, max_dref, ndiff);
            else printf("%-36s %8.4f ms %8.1f TFLOP/s\n", name, ms, gflop / ms);
            fflush(stdout);
        };
        auto timeline = [&](const char* name, auto launch_dbg) {
            hipMemset(dbg, 0, 256 * 128 * 8);
            if (launch_dbg(dbg) != 0) return;
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); launch_dbg(dbg); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(256 * 128);
            hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
            double tot = 0, totmax = 0, pa = 0, pb = 0; int nw = 0, nt = 0;
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < 256; ++b) {
                if (!h[b * 128]) continue;
                int lastj = -1;
                for (int j = 0; j < 40 && h[b * 128 + 1 + j * 3]; ++j) {
                    pa += (double)(h[b * 128 + 2 + j * 3] - h[b * 128 + 1 + j * 3]); pb += (double)(h[b * 128 + 3 + j * 3] - h[b * 128 + 2 + j * 3]); ++nt; lastj = j;
                }
                if (lastj < 0) continue;
                const double w = (double)(h[b * 128 + 3 + lastj * 3] - h[b * 128]);
                tot += w; totmax = w > totmax ? w : totmax; ++nw;
            }
            if (!nw) return;
            printf("    [%s] wg cycles avg %.0f max %.0f = %.3f GHz x %.4f ms | per tile: phase A %.0f, phase B %.0f | tiles/wg %.2f\n", name, tot / nw, totmax,
                   totmax / (ms * 1e6), ms, pa / nt, pb / nt, (double)nt / nw);
            for (int b : {0, 3}) {
                printf("      wg %d:", b);
                for (int j = 0; j < 4 && h[b * 128 + 1 + j * 3]; ++j) printf(" A %llu B %llu |", h[b * 128 + 2 + j * 3] - h[b * 128 + 1 + j * 3], h[b * 128 + 3 + j * 3] - h[b * 128 + 2 + j * 3]);
                printf("\n");
            }
            (void)t0; (void)t1;
            fflush(stdout);
        };
        time_it("shipped tail wd<1,4,tpx4,d4,HEAD 2>", [&] { return wd::launch_conv3x3_wd<1, 4, 4, 4, 0, 2>(a, 0); }, true, true);
#define TAILV(label, DA, DBv, SK)                                                                                \
        time_it(label, [&] { return wd9t::launch<DA, DBv, 0, 0>(a, 0, 256, nullptr, SK); }, true, false);        \
        timeline(label, [&](unsigned long long* d) { return wd9t::launch<DA, DBv, 0, 1>(a, 0, 256, d, SK); });
        TAILV("wd9 tail dA4 dB4", 4, 4, 0)
        TAILV("wd9 tail dA4 dB4 skew 2k", 4, 4, 2000)
        TAILV("wd9 tail dA4 dB4 skew 4k", 4, 4, 4000)
        TAILV("wd9 tail dA4 dB4 skew 6k", 4, 4, 6000)
        TAILV("wd9 tail dA4 dB4 skew 8k", 4, 4, 8000)
        TAILV("wd9 tail dA4 dB4 skew 12k", 4, 4, 12000)
        TAILV("wd9 tail dA4 dB4 (again)", 4, 4, 0)
        hipFree(din); hipFree(dw); hipFree(dwp); hipFree(dw3); hipFree(dw3p); hipFree(dres); hipFree(dout); hipFree(db); hipFree(db3);
    }
    return 0;
}
