// Fused SGD-with-momentum step over FLAT fp32 buffers for gfx950 (training half, SURVEY 8(f)-4, ranked last).
//
// Replaces the per-parameter loop of torch.optim.SGD that solver/build.py:93-133 (`build_optimizer`) configures - momentum
// SOLVER.MOMENTUM (0.9), per-group lr / weight decay (BIAS_LR_FACTOR, WEIGHT_DECAY_BIAS, WEIGHT_DECAY_NORM), dampening 0, no
// Nesterov - behind DefaultTrainer (engine/defaults.py:250-262):
//     d = grad * grad_scale + weight_decay * p        (grad_scale folds DDP's 1 / world_size and the inverse fp16 loss scale)
//     buf = first_step ? d : momentum * buf + d
//     p  -= lr * buf
// and, in the same pass over HBM, refreshes the fp16 shadow copy the MFMA GEMMs read (one kernel instead of SGD's four
// elementwise launches + a cast per parameter: 12 B read + 10 B written per element, HBM-bound).
// Parameters, gradients and momentum live in ONE flat buffer each (proben_amd/training.py::FlatParams), so a parameter group is a
// contiguous range and a step is one launch per group.  Built with -ffp-contract=off: the operation order above is the rounding.
#include <hip/hip_fp16.h>

#include "common.h"

namespace {
typedef float float4v __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

struct SgdArgs {
    float* p;
    const float* g;
    float* m;
    _Float16* h;      // optional fp16 shadow of p
    long long n;
    float lr, mu, wd, gscale;
    int first;
};

__device__ __forceinline__ float sgd1(float p, float g, float& m, const SgdArgs& a) {
    const float d = g * a.gscale + a.wd * p;
    m = a.first ? d : a.mu * m + d;
    return p - a.lr * m;
}

__global__ __launch_bounds__(256) void sgd_momentum_kernel(SgdArgs a) {
    const long long nvec = a.n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float4v p = reinterpret_cast<const float4v*>(a.p)[i];
        const float4v g = reinterpret_cast<const float4v*>(a.g)[i];
        float4v m = a.first ? float4v{0.f, 0.f, 0.f, 0.f} : reinterpret_cast<const float4v*>(a.m)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float me = m[e];
            p[e] = sgd1(p[e], g[e], me, a);
            m[e] = me;
        }
        reinterpret_cast<float4v*>(a.p)[i] = p;
        reinterpret_cast<float4v*>(a.m)[i] = m;
        if (a.h) {
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (_Float16)p[e];
            reinterpret_cast<half4*>(a.h)[i] = hv;
        }
    }
    // tail (n % 4 elements): one thread each
    const long long t0 = nvec << 2;
    const long long i = t0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) {
        float me = a.first ? 0.f : a.m[i];
        const float pn = sgd1(a.p[i], a.g[i], me, a);
        a.p[i] = pn;
        a.m[i] = me;
        if (a.h) a.h[i] = (_Float16)pn;
    }
}
}  // namespace

extern "C" int pe_sgd_momentum_f32(float* params, const float* grads, float* momentum_buf, void* fp16_shadow, int64_t n, float lr,
                                   float momentum, float weight_decay, float grad_scale, int32_t first_step, void* stream) {
    PE_CHECK_ARG(n >= 0, "pe_sgd_momentum_f32: negative element count");
    if (n == 0) return PE_OK;
    PE_CHECK_ARG(params && grads && momentum_buf, "pe_sgd_momentum_f32: null pointer");
    PE_CHECK_ARG(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) % 16 == 0 && (uintptr_t)fp16_shadow % 8 == 0,
                 "pe_sgd_momentum_f32: buffers must be 16-byte aligned (fp16 shadow: 8): pass group ranges that start at multiples of 4 elements");
    SgdArgs a{params, grads, momentum_buf, (_Float16*)fp16_shadow, (long long)n, lr, momentum, weight_decay, grad_scale, first_step ? 1 : 0};
    const long long nvec = n >> 2;
    long long blocks = (std::max<long long>(nvec, 4) + 255) / 256;
    blocks = std::min<long long>(blocks, 256 * 16);      // grid-stride: 16 workgroups per CU are plenty for an HBM-bound stream
    hipLaunchKernelGGL(sgd_momentum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    PE_CHECK_LAUNCH("pe_sgd_momentum_f32");
    return PE_OK;
}
