// Fused ResNet stem for gfx950: conv 7x7 / stride 2 / pad 3 (BN folded) + ReLU + max_pool2d(3, 2, 1).
//   BasicStem.forward (modeling/backbone/resnet.py:375-384): x = conv1(x); x = relu_(x); x = max_pool2d(x, 3, 2, 1).
//
// Why fused: unfused, the stem writes N*H/2*W/2*64 fp16 (839 MB for 32 images of 800x1024) and the pool reads
// it back; fused, a persistent block turns a 23 x 72 pixel input patch (NHWC4 fp16, 13 KB) into a 4 x 16 x 64
// pooled tile entirely through LDS: HBM sees the 4-channel image once and the pooled map once.
//
// One block (4 waves) per pooled tile of PH x PW = 5 x 16 (round 6; 4 x 16 before):
//   conv region  : rows 2*ph0-1 .. 2*ph0+9 (11), cols 2*pw0-1 .. 2*pw0+31 (33)  -> 363 conv pixels = 12 MFMA pixel tiles of 32,
//                  THREE per wave (4 x 16 gave 297 pixels = 10 tiles: 3 / 3 / 2 / 2 - the block waited for its two slow waves with half
//                  the matrix pipes idle, and computed 1.16 conv pixels per useful one where this tile computes 1.13)
//   input patch  : rows 4*ph0-5 .. +26 (27), cols 4*pw0-6 .. +71 (72); out-of-image pixels are zero (conv padding)
//   GEMM         : D[cout 64][pixel 32] per MFMA pair; K = 7 rows x (8 taps x 4 channels) = 14 steps of 16.
//                  The 8th tap is a zero weight in FRONT (tap t reads input column 2*c - 4 + t), which makes
//                  every fragment a 16-byte aligned pair of NHWC4 pixels: ds_read_b128, no shuffles.
//   weights      : A operand, 14 x 2 fragments held in registers for the block's whole life (persistent grid).
//   epilogue     : bias + ReLU -> fp16 -> LDS [363][72]; conv pixels outside the conv map become 0, which equals
//                  the pool's -inf padding because every window holds at least one real, non-negative value.
//   pool         : 3x3/2 max over the LDS tile, 16-byte stores of 8 channels.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int PH = 5, PW = 16;                       // pooled tile
constexpr int CR = 2 * PH + 1, CC = 2 * PW + 1;      // conv region 11 x 33
constexpr int NPIX = CR * CC;                        // 363
constexpr int PTILES = (NPIX + 31) / 32;             // 12 MFMA pixel tiles: three per wave
constexpr int IR = 2 * CR + 5, IC = 2 * CC + 6;      // input patch 27 x 72 pixels of 8 bytes
constexpr int CHUNKS = IR * IC / 2;                  // 16-byte chunks (pixel pairs): 972
constexpr int OPITCH = 72;                           // halves per conv pixel in LDS (64 + pad, 16-byte multiple)
constexpr int THREADS = 256;
constexpr int LOADS = (CHUNKS + THREADS - 1) / THREADS;  // 4
constexpr int KSTEPS = 14;
constexpr int WROW = 7 * 8 * 4;                      // 224 halves of weights per output channel
constexpr int PATCH_HALFS = IR * IC * 4, COUT_HALFS = NPIX * OPITCH;
constexpr int LDS_BYTES = (PATCH_HALFS + COUT_HALFS) * 2;   // 15 552 + 52 272 = 67 824: two blocks per CU
constexpr int POOL_ITEMS = PH * PW * 8;              // 8-channel groups of the pooled tile
static_assert(PTILES % 4 == 0, "the pixel tiles divide evenly over the four waves");
static_assert(PATCH_HALFS % 8 == 0, "the conv tile starts 16-byte aligned behind the patch");

struct StemArgs {
    const _Float16* x;     // [N, H, W, 4]
    const _Float16* w;     // [64, 7, 8, 4], tap 0 zero
    const float* bias;     // [64]
    _Float16* out;         // [N, Hp, Wp, 64]
    int N, H, W, Hc, Wc, Hp, Wp;
    int tiles_w, tiles_h, total_tiles;
};

struct Tile { int n, ph0, pw0; };

__device__ __forceinline__ Tile tile_of(const StemArgs& a, int t) {
    Tile r;
    r.pw0 = (t % a.tiles_w) * PW;
    t /= a.tiles_w;
    r.ph0 = (t % a.tiles_h) * PH;
    r.n = t / a.tiles_h;
    return r;
}

__device__ __forceinline__ void load_patch(const StemArgs& a, const Tile& t, half8 (&regs)[LOADS]) {
    const int row0 = 4 * t.ph0 - 5, col0 = 4 * t.pw0 - 6;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int q = threadIdx.x + i * THREADS;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q < CHUNKS) {
            const int pr = q / (IC / 2), pc = (q % (IC / 2)) * 2;
            const int gy = row0 + pr, gx = col0 + pc;  // gx even, W even: a pixel pair is in or out as a whole
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const half8*>(a.x + (((size_t)t.n * a.H + gy) * a.W + gx) * 4);
        }
        regs[i] = v;
    }
}

__global__ __launch_bounds__(THREADS, 2) void stem7x7_pool_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16* patch = reinterpret_cast<_Float16*>(smem);
    _Float16* cout_lds = patch + PATCH_HALFS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;

    // weights: A operand fragments, row = output channel
    half8 wf[KSTEPS][2];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            wf[ks][mb] = *reinterpret_cast<const half8*>(a.w + (size_t)(mb * 32 + l31) * WROW + ks * 16 + khalf * 8);
    float bv[2][4][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[mb][j][i] = a.bias[mb * 32 + 8 * j + 4 * khalf + i];

    int t = blockIdx.x;
    if (t >= a.total_tiles) return;
    half8 regs[LOADS];
    Tile cur = tile_of(a, t);
    load_patch(a, cur, regs);
    for (; t < a.total_tiles; t += gridDim.x) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int q = threadIdx.x + i * THREADS;
            if (q < CHUNKS) *reinterpret_cast<half8*>(patch + q * 8) = regs[i];
        }
        __syncthreads();
        const int tn = t + gridDim.x;
        Tile nxt = cur;
        if (tn < a.total_tiles) {  // prefetch the next patch while this one is consumed
            nxt = tile_of(a, tn);
            load_patch(a, nxt, regs);
        }
        // ---- conv: each wave takes pixel tiles wave, wave+4, ...
        for (int pt = wave; pt < PTILES; pt += 4) {
            const int p = pt * 32 + l31;
            const int pc = min(p, NPIX - 1);
            const int r = pc / CC, c = pc % CC;
            const _Float16* base = patch + ((2 * r) * IC + 2 * c + khalf * 2) * 4;
            float16v acc0 = {0}, acc1 = {0};
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const half8 f = *reinterpret_cast<const half8*>(base + (kh * IC + h * 4) * 4);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kh * 2 + h][0], f, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kh * 2 + h][1], f, acc1, 0, 0, 0);
                }
            }
            if (p < NPIX) {
                const int gr = 2 * cur.ph0 - 1 + r, gc = 2 * cur.pw0 - 1 + c;
                const bool valid = (unsigned)gr < (unsigned)a.Hc && (unsigned)gc < (unsigned)a.Wc;
                _Float16* dst = cout_lds + p * OPITCH + 4 * khalf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half4 o0, o1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o0[i] = (_Float16)(valid ? pe::relu_nan(acc0[j * 4 + i] + bv[0][j][i]) : 0.f);
                        o1[i] = (_Float16)(valid ? pe::relu_nan(acc1[j * 4 + i] + bv[1][j][i]) : 0.f);
                    }
                    *reinterpret_cast<half4*>(dst + 8 * j) = o0;
                    *reinterpret_cast<half4*>(dst + 32 + 8 * j) = o1;
                }
            }
        }
        __syncthreads();
        // ---- pool 3x3 / 2 over the conv tile, 8 channels per item
#pragma unroll
        for (int it = 0; it < (POOL_ITEMS + THREADS - 1) / THREADS; ++it) {
            const int item = threadIdx.x + it * THREADS;
            if (item >= POOL_ITEMS) break;
            const int g = item & 7, pp = item >> 3;
            const int pr = pp / PW, pcol = pp % PW;
            const _Float16* src = cout_lds + ((2 * pr) * CC + 2 * pcol) * OPITCH + g * 8;
            half8 m = *reinterpret_cast<const half8*>(src);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    m = __builtin_elementwise_maximum(m, *reinterpret_cast<const half8*>(src + (dy * CC + dx) * OPITCH));
                }
            const int oy = cur.ph0 + pr, ox = cur.pw0 + pcol;
            if (oy < a.Hp && ox < a.Wp)
                *reinterpret_cast<half8*>(a.out + (((size_t)cur.n * a.Hp + oy) * a.Wp + ox) * 64 + g * 8) = m;
        }
        cur = nxt;
        // the next iteration's patch store is ordered behind every wave's conv phase by the barrier above;
        // its conv phase (which rewrites cout_lds) is ordered behind this pool phase by its first barrier.
    }
}
}  // namespace

extern "C" int pe_stem_conv7x7_maxpool_f16(const void* x, const void* w_packed, const float* bias, void* out, int32_t N,
                                           int32_t H, int32_t W, void* stream) {
    PE_CHECK_ARG(x && w_packed && bias && out, "pe_stem_conv7x7_maxpool_f16: null pointer");
    PE_CHECK_ARG(N >= 1 && H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0,
                 "pe_stem_conv7x7_maxpool_f16: N %d H %d W %d (H, W must be multiples of 4)", N, H, W);
    StemArgs a{};
    a.x = (const _Float16*)x; a.w = (const _Float16*)w_packed; a.bias = bias; a.out = (_Float16*)out;
    a.N = N; a.H = H; a.W = W; a.Hc = H / 2; a.Wc = W / 2; a.Hp = H / 4; a.Wp = W / 4;
    a.tiles_w = pe::ceil_div(a.Wp, PW);
    a.tiles_h = pe::ceil_div(a.Hp, PH);
    const long long total = (long long)N * a.tiles_h * a.tiles_w;
    PE_CHECK_ARG(total < (1ll << 31), "pe_stem_conv7x7_maxpool_f16: too many tiles");
    a.total_tiles = (int)total;
    const int grid = (int)std::min<long long>(total, 256 * 2);
    PE_ENSURE_LDS(stem7x7_pool_kernel, (size_t)LDS_BYTES, "pe_stem_conv7x7_maxpool_f16");
    hipLaunchKernelGGL(stem7x7_pool_kernel, dim3(grid), dim3(THREADS), (size_t)LDS_BYTES, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_stem_conv7x7_maxpool_f16");
    return PE_OK;
}
