// Micro-benchmark (measurement aid, not part of the library): throughput of global_load_lds_dwordx4 as a function
// of the bytes a CU keeps in flight.  Each wave issues `n` LDS-DMA instructions (1 KiB each), waits vmcnt(0),
// and repeats; blocks x waves x n KiB are in flight per CU between waits.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_dma_probe.hip -o scripts/lds_dma_probe && scripts/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int N>
__global__ void probe(const unsigned char* src, size_t span_bytes, int rounds, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int waves = blockDim.x >> 6;
    // every wave streams its own region; consecutive rounds advance through the span (wraps)
    size_t off = ((size_t)blockIdx.x * waves + wave) * (size_t)N * 1024;
    const size_t stride = (size_t)gridDim.x * waves * N * 1024;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const unsigned char* p = src + (span_bytes > 1 ? (off + (size_t)i * 1024) % span_bytes : off + (size_t)i * 1024) + lane * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + (wave * N + i) * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (span_bytes > 1) off += stride;  // span_bytes == 1: every wave re-reads its own N KiB (L2 / L1 hits)
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = smem[blockIdx.x & 1023];
}

template <int N>
double run(const unsigned char* src, size_t span, int blocks, int threads, int rounds, unsigned* sink) {
    const size_t lds = (size_t)(threads / 64) * N * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(threads), lds, 0, src, span, 4, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(threads), lds, 0, src, span, rounds, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * (threads / 64) * N * 1024.0 * rounds / (ms * 1e-3) / 1e12;  // TB/s
}

int main() {
    const size_t big = (size_t)2 << 30, small = (size_t)16 << 20;  // HBM-streaming span / L2+MALL-resident span
    unsigned char* src; unsigned* sink;
    hipMalloc(&src, big); hipMemset(src, 1, big); hipMalloc(&sink, 1 << 20);
    printf("%-10s %-9s %-7s %-9s %-14s %-10s %-10s\n", "span", "blocks/CU", "waves", "KiB/wave", "KiB inflight/CU", "TB/s", "GB/s/CU");
    for (size_t span : {(size_t)1, small, big})
        for (int bpc : {1, 2, 4})
            for (int waves : {4, 8})
                for (int n : {2, 4, 8, 16}) {
                    const size_t lds = (size_t)waves * n * 1024;
                    if (lds * bpc > 160 * 1024 || waves * bpc > 32) continue;
                    const int blocks = 256 * bpc, rounds = 2000;
                    double t = n == 2 ? run<2>(src, span, blocks, waves * 64, rounds, sink)
                             : n == 4 ? run<4>(src, span, blocks, waves * 64, rounds, sink)
                             : n == 8 ? run<8>(src, span, blocks, waves * 64, rounds, sink)
                                      : run<16>(src, span, blocks, waves * 64, rounds, sink);
                    printf("%-10s %-9d %-7d %-9d %-14zu %-10.2f %-10.1f\n", span == 1 ? "private" : (span == small ? "16MiB" : "2GiB"), bpc, waves, n,
                           lds * bpc / 1024, t, t * 1e3 / 256);
                }
    return 0;
}
