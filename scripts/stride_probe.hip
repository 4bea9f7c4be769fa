// Is HBM efficiency sensitive to HOW a [M][C] fp16 activation is swept?  The 1x1 kernels read a K-chunk of 64 channels
// (128 B) of 128 consecutive pixel rows per step, i.e. 128-byte pieces at a row stride of 2 * C bytes, and come back for
// the next 128 B of the same rows a few microseconds later.  This probe sweeps a 210 MB buffer the same way with plain
// global loads (PIECE bytes per row per pass, rows of ROW bytes) and reports GB/s.
//   hipcc --offload-arch=gfx950 -O3 scripts/stride_probe.hip -o scripts/stride_probe && scripts/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float float4v __attribute__((ext_vector_type(4)));

// block = 256 threads handles `rows_per_block` consecutive rows; pass p reads bytes [p*PIECE, (p+1)*PIECE) of every row
template <int PIECE>
__global__ __launch_bounds__(256) void sweep(const char* src, float* sink, int row_bytes, int rows_per_block, int inflight) {
    constexpr int LPR = PIECE / 16;                 // lanes per row
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * rows_per_block * row_bytes;
    const int rows_per_pass = 256 / LPR;            // rows covered by one block-wide load
    float4v acc = {0, 0, 0, 0};
    const int passes = row_bytes / PIECE;
    for (int p = 0; p < passes; ++p) {
        for (int r0 = 0; r0 < rows_per_block; r0 += rows_per_pass * inflight) {
#pragma unroll 8
            for (int j = 0; j < inflight; ++j) {
                const int r = r0 + j * rows_per_pass + tid / LPR;
                if (r < rows_per_block) {
                    const float4v v = *reinterpret_cast<const float4v*>(src + base + (size_t)r * row_bytes + p * PIECE + (tid % LPR) * 16);
                    acc += v;
                }
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// read `row_bytes` per row (128-byte pieces, pass by pass like the 1x1 kernels), write `out_bytes` per row once at the end
__global__ __launch_bounds__(256) void sweep_rw(const char* src, char* dst, int row_bytes, int out_bytes, int rows_per_block) {
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * rows_per_block * row_bytes;
    float4v acc = {0, 0, 0, 0};
    for (int p = 0; p < row_bytes / 128; ++p)
#pragma unroll 4
        for (int r0 = 0; r0 < rows_per_block; r0 += 32) {
            const int r = r0 + tid / 8;
            acc += *reinterpret_cast<const float4v*>(src + base + (size_t)r * row_bytes + p * 128 + (tid % 8) * 16);
        }
    const size_t obase = (size_t)blockIdx.x * rows_per_block * out_bytes;
    const int lpr = out_bytes / 16;
    for (int i = tid; i < rows_per_block * lpr; i += 256)
        *reinterpret_cast<float4v*>(dst + obase + (size_t)i * 16) = acc;
}

// the register epilogue's pattern: lane l owns 64 contiguous bytes of row (l & 31) (+64 B for lanes 32..63), i.e. every
// 16-byte request of an instruction sits in a different 64-byte segment.  MODE 0: the four 16-byte pieces of a lane's 64 bytes
// back to back; MODE 1: one piece from each of four 32-row groups, coming back for the next piece later (the tail kernel's
// "channel quarter" order); MODE 2: coalesced reference (8 lanes per 128-byte piece of a row).
template <int MODE>
__global__ __launch_bounds__(256) void sweep_lane64(const char* src, float* sink, int row_bytes) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block = 128 rows; wave w owns byte columns [w * row_bytes / 4, +row_bytes / 4) of all 128 rows (like the tail's wn split)
    const size_t base = (size_t)blockIdx.x * 128 * row_bytes + (size_t)wave * (row_bytes / 4);
    float4v acc = {0, 0, 0, 0};
    for (int c = 0; c < row_bytes / 4 / 128; ++c) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc += *reinterpret_cast<const float4v*>(src + base + (size_t)(i * 32 + (lane & 31)) * row_bytes + c * 128 + (lane >> 5) * 64 + j * 16);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc += *reinterpret_cast<const float4v*>(src + base + (size_t)(i * 32 + (lane & 31)) * row_bytes + c * 128 + (lane >> 5) * 64 + j * 16);
        } else if (MODE == 3) {   // quarter order, but a quarter = one 32-byte sector per row (lane pair l, l + 32 adjacent)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc += *reinterpret_cast<const float4v*>(src + base + (size_t)(i * 32 + (lane & 31)) * row_bytes + c * 128 + j * 32 + (lane >> 5) * 16);
        } else if (MODE == 4) {   // quarter order behind a one-dword-per-line touch of the chunk's 128 lines
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[0] += *reinterpret_cast<const float*>(src + base + (size_t)(i * 64 + lane) * row_bytes + c * 128);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc += *reinterpret_cast<const float4v*>(src + base + (size_t)(i * 32 + (lane & 31)) * row_bytes + c * 128 + (lane >> 5) * 64 + j * 16);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc += *reinterpret_cast<const float4v*>(src + base + (size_t)(i * 8 + (lane >> 3)) * row_bytes + c * 128 + (lane & 7) * 16);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
    const size_t bytes = 1680ull << 20;   // well beyond the 256 MB MALL
    char* src; float* sink;
    hipMalloc(&src, bytes); hipMemset(src, 0, bytes); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rows_per_block = 128;
    for (int row_bytes : {2048, 512, 4096}) {
        const int blocks = (int)(bytes / ((size_t)rows_per_block * row_bytes));
        for (int piece : {128, 256, 512, 2048}) {
            if (piece > row_bytes) continue;
            for (int inflight : {2, 8}) {
                float best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    hipEventRecord(e0);
                    if (piece == 128) hipLaunchKernelGGL(sweep<128>, dim3(blocks), dim3(256), 0, 0, src, sink, row_bytes, rows_per_block, inflight);
                    if (piece == 256) hipLaunchKernelGGL(sweep<256>, dim3(blocks), dim3(256), 0, 0, src, sink, row_bytes, rows_per_block, inflight);
                    if (piece == 512) hipLaunchKernelGGL(sweep<512>, dim3(blocks), dim3(256), 0, 0, src, sink, row_bytes, rows_per_block, inflight);
                    if (piece == 2048) hipLaunchKernelGGL(sweep<2048>, dim3(blocks), dim3(256), 0, 0, src, sink, row_bytes, rows_per_block, inflight);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("row %4d B, piece %4d B per row per pass, %d loads in flight per lane: %7.3f ms  %6.0f GB/s\n", row_bytes, piece, inflight, best,
                       bytes / (best * 1e-3) / 1e9);
            }
        }
    }
    for (int mode = 0; mode < 5; ++mode) {
        const int rb = 2048;
        const int blocks = (int)(bytes / (128ull * rb));
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(sweep_lane64<0>, dim3(blocks), dim3(256), 0, 0, src, sink, rb);
            if (mode == 1) hipLaunchKernelGGL(sweep_lane64<1>, dim3(blocks), dim3(256), 0, 0, src, sink, rb);
            if (mode == 2) hipLaunchKernelGGL(sweep_lane64<2>, dim3(blocks), dim3(256), 0, 0, src, sink, rb);
            if (mode == 3) hipLaunchKernelGGL(sweep_lane64<3>, dim3(blocks), dim3(256), 0, 0, src, sink, rb);
            if (mode == 4) hipLaunchKernelGGL(sweep_lane64<4>, dim3(blocks), dim3(256), 0, 0, src, sink, rb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("lane-owns-64-bytes pattern, mode %d (0 = pieces back to back, 1 = quarter order, 2 = coalesced, 3 = quarter order by 32-byte sectors, 4 = quarter order behind a line touch): %7.3f ms  %6.0f GB/s\n", mode, best,
               bytes / (best * 1e-3) / 1e9);
    }
    char* dst; hipMalloc(&dst, bytes);
    for (auto io : {std::pair<int,int>{2048, 512}, {512, 2048}, {1024, 1024}, {2048, 2048}}) {
        const int rb = io.first, ob = io.second;
        const size_t rows = (840ull << 20) / (rb > ob ? rb : ob);
        const int blocks = (int)(rows / rows_per_block);
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(sweep_rw, dim3(blocks), dim3(256), 0, 0, src, dst, rb, ob, rows_per_block);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double moved = (double)blocks * rows_per_block * (rb + ob);
        printf("read %4d B + write %4d B per row: %7.3f ms  %6.0f GB/s (read+write)\n", rb, ob, best, moved / (best * 1e-3) / 1e9);
    }
    return 0;
}
