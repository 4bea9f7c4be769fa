// C-ABI of the weights-direct convolution kernels (csrc/conv_wd.h): packing, support query, launch.
#include "conv_wd.h"

namespace pe {
int wd9_conv3x3(ConvWdArgs a, hipStream_t st);   // csrc/conv_wd9.hip: the large launches of the same layers, same bits
int wd9_rpn_head(ConvWdArgs a, hipStream_t st);   // csrc/conv_wd9.hip: the fused RPN head on the one-wave structure, same bits
}

namespace {

// channel split of a block: WN waves x 64 output channels
inline int wd_wn(int Cout) { return Cout % 256 == 0 ? 4 : 0; }

}  // namespace

extern "C" int pe_conv_wd_supported(int32_t kernel, int32_t stride, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (kernel != 3 || stride != 1) return 0;
    if (Cin % 64 || Cin <= 0 || wd_wn(Cout) == 0 || H <= 0) return 0;
    int seg, nseg;
    return wd::wd3x3_geometry(W, 128, &seg, &nseg) ? 1 : 0;
}

extern "C" int pe_conv_wd_pack_weights(const void* weight, void* packed, int32_t Cout, int32_t Cin, int32_t kernel,
                                       void* stream) {
    PE_CHECK_ARG(weight && packed, "pe_conv_wd_pack_weights: null pointer");
    PE_CHECK_ARG(kernel == 3, "pe_conv_wd_pack_weights: kernel %d not supported (3x3 only)", kernel);
    PE_CHECK_ARG(Cin > 0 && Cin % 64 == 0 && wd_wn(Cout) != 0,
                 "pe_conv_wd_pack_weights: needs Cin %% 64 == 0 and Cout %% 256 == 0 (got %d, %d)", Cin, Cout);
    const int K = 9 * Cin;
    const long long total = (long long)(Cout / 32) * (K / 16) * 64;
    hipLaunchKernelGGL(wd::pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)weight, (_Float16*)packed, Cout, K, Cin, wd_wn(Cout), 1);
    PE_CHECK_LAUNCH("pe_conv_wd_pack_weights");
    return PE_OK;
}

extern "C" int pe_conv3x3_wd_f16(const void* input, const void* packed_weight, const float* bias, void* output,
                                 int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t relu,
                                 int32_t out_stride, void* stream) {
    PE_CHECK_ARG(input && packed_weight && bias && output, "pe_conv3x3_wd_f16: null pointer (bias is required)");
    PE_CHECK_ARG(N > 0 && H > 0 && W > 0, "pe_conv3x3_wd_f16: bad dims");
    if (!pe_conv_wd_supported(3, 1, H, W, Cin, Cout)) {
        pe::set_error("pe_conv3x3_wd_f16: geometry not supported (W %d, Cin %d, Cout %d): use pe_conv2d_nhwc_f16", W, Cin, Cout);
        return PE_ERR_UNSUPPORTED;
    }
    const long long M = (long long)N * H * W;
    PE_CHECK_ARG(M * Cin * 2 < (1ll << 31), "pe_conv3x3_wd_f16: input larger than 2 GiB (32-bit buffer offsets)");
    const int os = out_stride > 0 ? out_stride : Cout;
    PE_CHECK_ARG(os % 8 == 0, "pe_conv3x3_wd_f16: out_stride must be a multiple of 8");
    pe::ConvWdArgs a{};
    a.in = (const _Float16*)input; a.wpk = (const _Float16*)packed_weight; a.bias = bias; a.res = nullptr;
    a.out = (_Float16*)output; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.M = (int)M; a.relu = relu;
    a.out_stride = os;
    PE_CHECK_ARG(M * os * 2 < (1ll << 32), "pe_conv3x3_wd_f16: output window larger than 4 GiB (32-bit buffer offsets)");
    int st = pe::wd9_conv3x3(a, (hipStream_t)stream);
    if (st == PE_ERR_UNSUPPORTED) st = wd::launch_conv3x3_wd<1, 4, 4, 4>(a, (hipStream_t)stream);
    if (st != PE_OK) {
        pe::set_error("pe_conv3x3_wd_f16: unsupported geometry");
        return st;
    }
    PE_CHECK_LAUNCH("pe_conv3x3_wd_f16");
    return PE_OK;
}

extern "C" int pe_conv_wd_pack_head(const void* head_weight, void* packed, int32_t rows, int32_t C, void* stream) {
    PE_CHECK_ARG(head_weight && packed && rows >= 1 && rows <= 16 && C == 256,
                 "pe_conv_wd_pack_head: needs rows <= 16 and 256 input channels (got %d, %d)", rows, C);
    hipLaunchKernelGGL(wd::pack_head_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, (const _Float16*)head_weight,
                       (_Float16*)packed, rows, C);
    PE_CHECK_LAUNCH("pe_conv_wd_pack_head");
    return PE_OK;
}

extern "C" int pe_conv3x3_wd_rpn_head_f16(const void* input, const void* packed_weight, const float* bias,
                                          const void* packed_head, const float* head_bias16, float* head_out, int32_t N,
                                          int32_t H, int32_t W, int32_t Cin, void* stream) {
    PE_CHECK_ARG(input && packed_weight && bias && packed_head && head_bias16 && head_out, "pe_conv3x3_wd_rpn_head_f16: null pointer");
    PE_CHECK_ARG(N > 0 && H > 0 && W > 0, "pe_conv3x3_wd_rpn_head_f16: bad dims");
    if (!pe_conv_wd_supported(3, 1, H, W, Cin, 256)) {
        pe::set_error("pe_conv3x3_wd_rpn_head_f16: geometry not supported (W %d, Cin %d): run the two convolutions", W, Cin);
        return PE_ERR_UNSUPPORTED;
    }
    const long long M = (long long)N * H * W;
    PE_CHECK_ARG(M * Cin * 2 < (1ll << 31), "pe_conv3x3_wd_rpn_head_f16: input larger than 2 GiB (32-bit buffer offsets)");
    pe::ConvWdArgs a{};
    a.in = (const _Float16*)input; a.wpk = (const _Float16*)packed_weight; a.bias = bias; a.out = nullptr;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = 256; a.M = (int)M; a.relu = 1; a.out_stride = 256;
    a.head_w = (const _Float16*)packed_head; a.head_b = head_bias16; a.head_out = head_out;
    int st = pe::wd9_rpn_head(a, (hipStream_t)stream);
    if (st == PE_ERR_UNSUPPORTED) st = wd::launch_conv3x3_wd<1, 4, 4, 4, 0, 1>(a, (hipStream_t)stream);
    if (st != PE_OK) {
        pe::set_error("pe_conv3x3_wd_rpn_head_f16: unsupported geometry");
        return st;
    }
    PE_CHECK_LAUNCH("pe_conv3x3_wd_rpn_head_f16");
    return PE_OK;
}

extern "C" int pe_conv_wd_pack_tail(const void* weight, void* packed, int32_t tail_cout, int32_t C, void* stream) {
    PE_CHECK_ARG(weight && packed && C == 256 && tail_cout > 0 && tail_cout % 256 == 0,
                 "pe_conv_wd_pack_tail: needs 256 input channels and tail_cout %% 256 == 0 (got %d, %d)", C, tail_cout);
    const int total = tail_cout * 256 / 8;
    hipLaunchKernelGGL(wd::pack_tail_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const _Float16*)weight,
                       (_Float16*)packed, tail_cout);
    PE_CHECK_LAUNCH("pe_conv_wd_pack_tail");
    return PE_OK;
}

extern "C" int pe_bottleneck_tail_wd_f16(const void* input, const void* packed_weight3x3, const float* bias3x3,
                                         const void* packed_tail, const float* tail_bias, const void* residual, void* output,
                                         int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t tail_cout, void* stream) {
    PE_CHECK_ARG(input && packed_weight3x3 && bias3x3 && packed_tail && tail_bias && output, "pe_bottleneck_tail_wd_f16: null pointer");
    PE_CHECK_ARG(N > 0 && H > 0 && W > 0 && tail_cout > 0 && tail_cout % 256 == 0, "pe_bottleneck_tail_wd_f16: bad dims");
    if (!pe_conv_wd_supported(3, 1, H, W, Cin, 256)) {
        pe::set_error("pe_bottleneck_tail_wd_f16: geometry not supported (W %d, Cin %d): run the two convolutions", W, Cin);
        return PE_ERR_UNSUPPORTED;
    }
    const long long M = (long long)N * H * W;
    PE_CHECK_ARG(M * tail_cout * 2 < (1ll << 31) && M * Cin * 2 < (1ll << 31), "pe_bottleneck_tail_wd_f16: tensors larger than 2 GiB (32-bit buffer offsets)");
    pe::ConvWdArgs a{};
    a.in = (const _Float16*)input; a.wpk = (const _Float16*)packed_weight3x3; a.bias = bias3x3; a.out = nullptr;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = 256; a.M = (int)M; a.relu = 1; a.out_stride = 256;
    a.tail_w = (const _Float16*)packed_tail; a.tail_b = tail_bias; a.tail_res = (const _Float16*)residual;
    a.tail_out = (_Float16*)output; a.tail_cout = tail_cout;
    int st = wd::launch_conv3x3_wd<1, 4, 4, 4, 0, 2>(a, (hipStream_t)stream);
    if (st != PE_OK) {
        pe::set_error("pe_bottleneck_tail_wd_f16: unsupported geometry");
        return st;
    }
    PE_CHECK_LAUNCH("pe_bottleneck_tail_wd_f16");
    return PE_OK;
}

#ifdef PE_LAB
// LAB library only (python -m proben_amd.build --lab): the fused tail with the work of a fused NEXT conv1 added on garbage (conv_wd.h, ABL & 8;
// results WRONG) - scripts/lab/r06_tail_next_pricing.py.  `next_out`: [M, 256] fp16 scratch for the extra lines.
extern "C" int pe_lab_bottleneck_tail_next_pricing(const void* input, const void* packed_weight3x3, const float* bias3x3, const void* packed_tail,
                                                   const float* tail_bias, const void* residual, void* output, void* next_out, int32_t N,
                                                   int32_t H, int32_t W, int32_t Cin, int32_t tail_cout, void* stream) {
    const long long M = (long long)N * H * W;
    pe::ConvWdArgs a{};
    a.in = (const _Float16*)input; a.wpk = (const _Float16*)packed_weight3x3; a.bias = bias3x3; a.out = (_Float16*)next_out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = 256; a.M = (int)M; a.relu = 1; a.out_stride = 256;
    a.tail_w = (const _Float16*)packed_tail; a.tail_b = tail_bias; a.tail_res = (const _Float16*)residual;
    a.tail_out = (_Float16*)output; a.tail_cout = tail_cout;
    const int st = wd::launch_conv3x3_wd<1, 4, 4, 4, 8, 2>(a, (hipStream_t)stream);
    if (st != PE_OK) return st;
    PE_CHECK_LAUNCH("pe_lab_bottleneck_tail_next_pricing");
    return PE_OK;
}
#endif
