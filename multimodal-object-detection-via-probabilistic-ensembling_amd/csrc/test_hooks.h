// Exported by libproben_hip.so for tests/ and scripts/ ONLY - not part of the drop-in C-ABI (include/proben_hip.h), not bound by the
// product's Python surface except through proben_amd._lib.test_hooks().
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* Kernel-selection policy of pe_conv2d_nhwc_f16 (process-global, relaxed atomics; affects launches issued afterwards).
 *   tile_bits (default 329 = 1|8|64|256):  1: 256-row block tiles (8 waves) for 3x3 launches with >= 512 such tiles   2: the same for 1x1
 *                                 4: two-stage pipeline in the generic 1x1 kernel   8: 256x256 two-stage kernel for long-K GEMMs
 *                                16: 256x256 kernel for every eligible launch
 *                                32: persistent loader / consumer 1x1 kernel (csrc/conv1x1_ring.hip) for K >= 1024, Cout == 256 (res4 conv1)
 *                                64: ... for every eligible residual-free 1x1 launch (bias, fp16 out, Cout % 256 == 0; K >= 512 at stride 1, K >= 256 at stride 2)
 *                               256: ... and for the stride-1 layers WITH a residual (both modes) from K = 128
 *   reuse3x3 (default 1): 1 = kw-reuse 3x3 kernel, 0 = generic per-tap 3x3 kernel */
int pe_test_set_conv_policy(int tile_bits, int reuse3x3);
/* Which generation of the weights-direct kernels takes a launch.  Bit mask (default 1 | 8; a negative mode restores the default;
 * bit 2 was round 4's fused tail on this structure, now scripts/lab/conv_wd9_tail.h):
 *   1 = pure 3x3: conv_wd9.h for launches of >= 128 tiles of 256 pixels (same bits as conv_wd.h)   2 = ... whenever the geometry allows
 *   8 = fused RPN head: conv_wd9.h's head epilogue under the size rule of bit 0 / bit 1 (same bits)  0 = conv_wd.h only */
int pe_test_set_wd9_mode(int mode);
/* 1 when a 3x3 launch of this shape is taken by the conv_wd9.h kernel under the current mode (bench.py labels its kernel table with it) */
int pe_test_wd9_takes(int N, int H, int W, int Cin, int Cout);
/* the same for the fused RPN head (pe_conv3x3_wd_rpn_head_f16) */
int pe_test_wd9_head_takes(int N, int H, int W);
/* workgroups of the persistent 3x3 kernels (multiples of 8 in 8 .. 256; 0 = leave unchanged; default 256 = one per CU): what leaving CUs
 * to the other detector's stream is worth (scripts/archive/r04_ab2.sh); the second argument is ignored (it sized the round-4 tail kernel) */
int pe_test_set_wd9_wgs(int pure, int tail);
/* workgroups of the persistent 1x1 ring kernel (default 256 = one per CU) */
int pe_test_set_ring_wgs(int wgs);
#ifdef PE_LAB
/* LAB library only (`python -m proben_amd.build --lab`): ablation bits of the ring kernel (csrc/conv1x1_ring.hip RingArgs::abl; non-zero =
 * wrong results, timing only).  The product library has neither the symbol nor the branches. */
int pe_test_set_ring_ablation(int bits);
#endif
/* batched NMS: 1 (default) = input whose live rows are already in (class, score descending, row) order skips the sorting network
 * (the RPN's hand-over), 0 = the network always runs (same result; A/B and the identity test).  Synchronises the device. */
int pe_test_set_nms_presorted(int on);
/* ROIAlign (fp16, C == 256): 1 (default) = the wave-uniform form, 0 = the per-lane form for every launch (same bits; A/B and the identity test) */
int pe_test_set_roi_fast(int on);
#ifdef __cplusplus
}
#endif
