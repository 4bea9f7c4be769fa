// Can a CU overlap MFMA work with LDS fragment reads and LDS-DMA (global_load_lds) issued by OTHER waves?
// One 512-thread block per CU.  Waves 0-3 (one per SIMD) run a chain-free MFMA loop; waves 4-7 (the second wave of
// every SIMD) run, depending on `mode`, nothing / ds_read_b128 loop / global_load_lds loop / both.
//   hipcc --offload-arch=gfx950 -O3 scripts/overlap_probe.hip -o scripts/overlap_probe && scripts/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__global__ __launch_bounds__(512, 2) void probe(const unsigned char* src, float* sink, int iters, int mode_mfma, int mode_other) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float acc_out = 0.f;
    if (wave < 4) {
        if (mode_mfma) {
            float16v acc[8];
            for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            half8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(lane * 0.001f + e); b[e] = (_Float16)(0.5f - e * 0.01f); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
            for (int i = 0; i < 8; ++i) acc_out += acc[i][0];
        }
    } else {
        const int w = wave - 4;
        if (mode_other & 1) {   // fragment-read loop: 12 ds_read_b128 per iteration (the p8 kernel's heaviest phase)
            half8 s = {0, 0, 0, 0, 0, 0, 0, 0};
            const unsigned lds_addr = (unsigned)(size_t)(lptr_t)(smem + w * 8192 + (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4));
            for (int it = 0; it < iters; ++it) {
                half8 v0, v1, v2, v3;
#pragma unroll
                for (int k = 0; k < 3; ++k) {   // 12 reads, conflict-free swizzled rows, results discarded
                    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:4128"
                                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(lds_addr));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                s = v0;
            }
            acc_out += (float)s[0];
        }
        if (mode_other & 4) {   // register-staged loop: 2 x (global_load_dwordx4 -> VGPR -> ds_write_b128) per iteration
            const unsigned char* g = src + (size_t)blockIdx.x * 65536 + lane * 16;
            unsigned char* l = smem + 65536 + w * 2048 + lane * 16;
            half8 r0 = *reinterpret_cast<const half8*>(g), r1 = *reinterpret_cast<const half8*>(g + 1024);
            for (int it = 1; it <= iters; ++it) {
                const half8 n0 = *reinterpret_cast<const half8*>(g + ((it * 2) & 63) * 1024);
                const half8 n1 = *reinterpret_cast<const half8*>(g + ((it * 2 + 1) & 63) * 1024);
                *reinterpret_cast<half8*>(l) = r0;
                *reinterpret_cast<half8*>(l + 1024) = r1;
                r0 = n0; r1 = n1;
            }
            acc_out += (float)r0[0] + (float)r1[0];
        }
        if (mode_other & 2) {   // LDS-DMA loop: 2 x 1 KiB per iteration from an L2-resident 64 KiB window per CU
            const unsigned char* g = src + (size_t)blockIdx.x * 65536 + lane * 16;
            for (int it = 0; it < iters; ++it) {
                __builtin_amdgcn_global_load_lds((gptr_t)(g + ((it * 2) & 63) * 1024), (lptr_t)(smem + 65536 + w * 2048), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(g + ((it * 2 + 1) & 63) * 1024), (lptr_t)(smem + 65536 + w * 2048 + 1024), 16, 0, 0);
                if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (acc_out == 12345.678f) sink[0] = acc_out;
}

int main() {
    const int blocks = 256, iters = 20000;
    unsigned char* src; float* sink;
    hipMalloc(&src, (size_t)blocks * 65536); hipMemset(src, 0, (size_t)blocks * 65536);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-34s %10s %12s %14s %14s\n", "config", "ms", "MFMA TF/s", "LDS rd B/clk/CU", "DMA GB/s/CU");
    const char* names[] = {"other waves idle", "other waves: ds_read_b128", "other waves: global_load_lds", "other waves: both",
                           "other waves: load + ds_write", "other waves: ld+ds_write+ds_read"};
    for (int mm = 1; mm >= 0; --mm)
        for (int mo = 0; mo < 6; ++mo) {
            if (!mm && !mo) continue;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 96 * 1024, 0, src, sink, iters, mm, mo);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tf = mm ? 2.0 * 32 * 32 * 16 * 8.0 * iters * 4 * blocks / (ms * 1e-3) / 1e12 : 0;
            const double rd = (mo & 1) ? 12.0 * 1024 * iters * 4 / (ms * 1e-3) / 2.1e9 : 0;   // bytes per clock at 2.1 GHz
            const double dma = (mo & 6) ? 2.0 * 1024 * iters * 4 / (ms * 1e-3) / 1e9 : 0;
            char label[64]; snprintf(label, sizeof label, "%s%s", mm ? "MFMA + " : "no MFMA, ", names[mo]);
            printf("%-34s %10.3f %12.0f %14.1f %14.1f\n", label, ms, tf, rd, dma);
        }
    return 0;
}
