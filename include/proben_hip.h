/*
 * proben_hip.h - C-ABI of libproben_hip.so: the MI355X (gfx950) implementation of the
 * RGB+thermal detection-and-fusion inference path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every entry point
 *     only ENQUEUES work on that stream: no allocation, no hidden synchronisation;
 *   - the caller allocates all outputs and scratch;
 *   - return value: 0 = ok, negative = error (PE_ERR_*); pe_last_error() returns a
 *     thread-local, human-readable message for the last failing call on this thread.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * upstream repository Jamie725/Multimodal-Object-Detection-via-Probabilistic-Ensembling).
 */
#ifndef PROBEN_HIP_H
#define PROBEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PE_OK 0
#define PE_ERR_INVALID_ARG (-1)
#define PE_ERR_UNSUPPORTED (-2)
#define PE_ERR_HIP (-3)

/* score_mode / box_mode of pe_proben_fuse_batch (demo/FLIR/demo_probEn.py:145-167,
 * CLI flags --score_fusion / --box_fusion in detectron2/utils/opt.py:14-17) */
#define PE_SCORE_PROBEN 0
#define PE_SCORE_AVG 1
#define PE_SCORE_MAX 2
#define PE_SCORE_PROBEN_BINARY 3 /* demo_probEn.py:24-30, the K = 1 (KAIST) form */
#define PE_BOX_VAVG 0
#define PE_BOX_SAVG 1
#define PE_BOX_AVG 2
#define PE_BOX_ARGMAX 3

const char* pe_last_error(void);
int pe_version(void);

/* ---------------------------------------------------------------------------------------------
 * ProbEn late fusion, batched over images.
 * Replaces nms_bayesian + bayesian_fusion_multiclass + weighted_box_fusion + avg_bbox_fusion
 * (demo/FLIR/demo_probEn.py:20-42,73-77,92-187), called per image from `fusion`
 * (demo_probEn.py:189-196) inside apply_late_fusion_and_evaluate (demo_probEn.py:198-298).
 *
 * Rows of image b are [offsets[b], offsets[b+1]) of the flat arrays, already concatenated in
 * detector order (prepare_data, demo_probEn.py:79-90).  float64 in (the reference's NumPy math),
 * float64 boxes / float32 scores and classes out (the reference's torch.Tensor exit).
 * Output rows of image b are written at [offsets[b], offsets[b] + out_counts[b]) in pivot order.
 * out_counts[b] = -1 if the image has more than max_rows_per_image rows.
 * One 1024-thread workgroup per image; max_rows_per_image (R) sizes its LDS slab: (8 * (11 + L) + 17) bytes per row, L = num_classes + 1
 * (probEn), 2 (binary) or 0 - 160 KiB hold 1 195 rows at num_classes 3; a bound that does not fit returns PE_ERR_UNSUPPORTED.  With
 * 16 * ceil(R / 64) more bytes per row (R <= ~560 at num_classes 3) the pair tests go into two bit matrices first; the results are the same.
 * ------------------------------------------------------------------------------------------- */
int pe_proben_fuse_batch(const double* boxes,     /* [Ntot,4] xyxy */
                         const double* scores,    /* [Ntot] */
                         const double* probs,     /* [Ntot,K] */
                         const double* variances, /* [Ntot] */
                         const int32_t* classes,  /* [Ntot] */
                         const int32_t* offsets,  /* [B+1] (or [B] when row_counts is given) */
                         const int32_t* row_counts, /* optional [B]: rows of image b = row_counts[b] */
                         const int32_t* passthrough, /* optional [B]: != 0 -> copy the rows unchanged (only ONE
                                                        detector fired: demo_probEn.py:240-253) */
                         int32_t num_images, int32_t num_classes, int32_t max_rows_per_image,
                         int32_t score_mode, int32_t box_mode, double iou_thresh,
                         double frame_w, double frame_h, /* class-band shift: 640, 512 */
                         double* out_boxes,              /* [Ntot,4] */
                         float* out_scores,              /* [Ntot] */
                         float* out_classes,             /* [Ntot] */
                         int32_t* out_keep,              /* [Ntot] row index (image-local) of each pivot */
                         int32_t* out_counts,            /* [B] */
                         void* stream);

/* Detector outputs -> ProbEn input rows, on the device (replaces the JSON hop between
 * demo/FLIR/demo_FLIR_save_predictions.py:133-176 and demo_probEn.py:205-234 + prepare_data :79-90).
 * det_*[d]: DEVICE pointers of detector d's padded outputs: boxes f32 [B,D,4], scores f32 [B,D],
 * classes i32 [B,D], probs f32 [B,D,K], vars f32 [B,D], counts i32 [B]  (host arrays of pointers).
 * Rows with class > max_class are dropped (the reference keeps `classes <= 2`, :148-155).
 * Image b's rows are written at b*row_stride in detector order; out_counts[b] = rows written;
 * out_single_source[b] = 1 when exactly one detector contributed rows (-> passthrough). */
int pe_proben_pack_detections(const float* const* det_boxes_host, const float* const* det_scores_host,
                              const int32_t* const* det_classes_host, const float* const* det_probs_host,
                              const float* const* det_vars_host, const int32_t* const* det_counts_host,
                              int32_t num_detectors, int32_t num_images, int32_t det_stride,
                              int32_t num_classes, int32_t max_class, int32_t row_stride, double* out_boxes,
                              double* out_scores, double* out_probs, double* out_vars, int32_t* out_classes,
                              int32_t* out_offsets, int32_t* out_counts, int32_t* out_single_source,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused convolution / GEMM: NHWC fp16 activations, [Cout][KH][KW][Cin] fp16 weights, fp32 accumulate
 * on MFMA, epilogue = + bias[Cout] (fp32) + residual + ReLU, fp16 (or fp32) NHWC output.
 * Replaces, per layer, detectron2.layers.Conv2d.forward (layers/wrappers.py:62-98) + FrozenBatchNorm2d
 * (layers/batch_norm.py:45-65; folded into weight/bias by the caller) + relu_ + the residual add of
 * BottleneckBlock.forward (modeling/backbone/resnet.py:205-221) + the nearest-2x top-down add of
 * FPN.forward (modeling/backbone/fpn.py:129-137); with H = W = 1 it is the nn.Linear of
 * FastRCNNConvFCHead / FastRCNNOutputLayers (roi_heads/box_head.py:73-81, fast_rcnn.py:531-545).
 *   kernel 1: 1x1, stride 1|2, no padding, Cin % 64 == 0
 *   kernel 3: 3x3, stride 1, padding 1, Cin % 64 == 0
 *   kernel 7: the stem, 7x7 stride 2 padding 3 over an NHWC4 input; weight packed [Cout][8][8][4]
 *             (kh 0..6 real + 1 zero row, kw 0..6 real + 1 zero column, 4 channels)
 *   residual_mode 0: none; 1: residual has the output's shape; 2: residual is [N,res_h,res_w,Cout]
 *             and is read at (oh/2, ow/2) (nearest-2x upsample).
 *   out_f32 != 0: fp32 output, only channels [0, cout_store) are written, row stride out_stride.
 * Kernels behind it (all with the same fp32 summation order per output - K ascending, zero-initialised accumulators, then + bias,
 * + residual, ReLU - so which one takes a launch never shows in a result, and the choice looks at channel counts / stride only, never at
 * the batch): csrc/conv1x1_ring.hip (persistent loader / consumer kernel: 1x1 with a bias, fp16 output, Cout % 256 == 0, tensors < 2 GiB;
 * residual-free from Cin 512 at stride 1 / 256 at stride 2, with a residual from Cin 128), csrc/conv_igemm2.hip (every other 1x1 and 3x3),
 * csrc/conv_igemm.hip (the unfused stem).
 * ------------------------------------------------------------------------------------------- */
int pe_conv2d_nhwc_f16(const void* input, const void* weight, const float* bias, const void* residual,
                       void* output, int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                       int32_t kernel, int32_t stride, int32_t relu, int32_t residual_mode,
                       int32_t res_h, int32_t res_w, int32_t out_f32, int32_t cout_store,
                       int32_t out_stride, void* stream);
/* (kernel-selection knobs for A/B measurements are not part of this ABI: csrc/test_hooks.h) */

/* ---------------------------------------------------------------------------------------------
 * "Weights-direct" 3x3 convolution (csrc/conv_wd.h): same reference rows as pe_conv2d_nhwc_f16 with kernel 3
 * (layers/wrappers.py:62-98 + folded FrozenBatchNorm2d layers/batch_norm.py:45-65 + relu_; the 3x3 convolutions of
 * backbone/resnet.py:205-221, backbone/fpn.py:127-137 and proposal_generator/rpn.py:74-85).
 * The weights are packed ONCE into MFMA-fragment order (pe_conv_wd_pack_weights) and streamed L2 -> VGPR; only the
 * pixels go through LDS.  Supported: stride 1 / pad 1, Cin % 64 == 0, Cout % 256 == 0, W % 32 == 0 with W | 128 or
 * 128 | W, input < 2 GiB; everything else -> pe_conv2d_nhwc_f16.  Epilogue: + bias (required) + ReLU, fp16 NHWC
 * output with row stride out_stride (0 = Cout).
 * ------------------------------------------------------------------------------------------- */
int pe_conv_wd_supported(int32_t kernel, int32_t stride, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
/* Two kernel generations stand behind pe_conv3x3_wd_f16, pe_conv3x3_wd_rpn_head_f16 and pe_bottleneck_tail_wd_f16: two waves per SIMD,
 * one tile per workgroup (csrc/conv_wd.h), and one wave per SIMD with 256 accumulators in the AGPRs and PERSISTENT workgroups
 * (csrc/conv_wd9.h: the pure 3x3 and the fused RPN head at image widths 64 / 128 / 256 from 128 tiles of 256 pixels on - the same
 * bits as the two-wave kernels, so the size rule may look at the batch).  pe_bottleneck_tail_wd_f16 always runs on csrc/conv_wd.h.
 * A persistent kernel occupies every CU it runs on for its whole duration; the ones that ship (csrc/conv_wd9.h, csrc/conv1x1_ring.hip)
 * measured best at one workgroup per CU under one, two and three concurrent detector streams (profiles/r04_pipeline_ab_2.txt,
 * r05_pipeline_ab_ring.txt), so there is nothing for a caller to tune (round 4-5's `pe_conv_wd_set_concurrent_streams` hint, a
 * validated no-op since its kernel left the library, was removed from the ABI in round 6). */
/* weight: [Cout][3][3][Cin] fp16 (the layout pe_conv2d_nhwc_f16 takes); packed: Cout*9*Cin halfs */
int pe_conv_wd_pack_weights(const void* weight, void* packed, int32_t Cout, int32_t Cin, int32_t kernel, void* stream);
int pe_conv3x3_wd_f16(const void* input, const void* packed_weight, const float* bias, void* output, int32_t N,
                      int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t relu, int32_t out_stride, void* stream);
/* StandardRPNHead.forward for one level in ONE launch (proposal_generator/rpn.py:74-85): t = relu(conv3x3(x)) with 256
 * output channels never leaves the chip; head_out[m][0..15] = head_bias16 + head_weight[rows <= 16][256] * t[m]  (fp32 rows of
 * 16: 3 objectness logits, 12 anchor deltas, 1 pad - the layout pe_rpn_select_topk reads).  head_weight is packed once with
 * pe_conv_wd_pack_head ([rows][256] fp16 in, 16 KiB out); geometry rule of pe_conv_wd_supported with Cout = 256. */
/* The second half of BottleneckBlock.forward in ONE launch (modeling/backbone/resnet.py:207-221): conv2 3x3 (Cin -> 256,
 * folded BN) + ReLU -> conv3 1x1 (256 -> tail_cout, folded BN) + shortcut + ReLU.  The 256-channel intermediate stays on the
 * chip (LDS) and the stand-alone, latency-bound 1x1 launch disappears.  tail weight [tail_cout][256] fp16 is packed once with
 * pe_conv_wd_pack_tail (same size out); residual (optional) and output are [N,H,W,tail_cout] fp16; tail_cout % 256 == 0;
 * geometry rule of pe_conv_wd_supported with Cout = 256. */
int pe_conv_wd_pack_tail(const void* weight, void* packed, int32_t tail_cout, int32_t C, void* stream);
int pe_bottleneck_tail_wd_f16(const void* input, const void* packed_weight3x3, const float* bias3x3, const void* packed_tail,
                              const float* tail_bias, const void* residual, void* output, int32_t N, int32_t H, int32_t W,
                              int32_t Cin, int32_t tail_cout, void* stream);
/* A 64-channel-wide stride-1 BottleneckBlock from its 3x3 on, plus the next block's first convolution, in ONE launch (res2:
 * modeling/backbone/resnet.py:107-221 with bottleneck_channels = 64, out_channels = 256; build_resnet_backbone :558-572):
 *     t2 = relu(conv2_3x3(t1) + bias2);  out = relu(conv3(t2) + bias3 + shortcut);  [t1_next = relu(conv1_next(out) + bias1n)]
 * t1 [N,H,W,64] fp16 is relu(conv1(x)) of this block.  shortcut_src: the block input x [N,H,W,256] (identity shortcut), or -
 * has_shortcut_conv - the input s [N,H,W,64] of the first block's shortcut convolution (64 -> 256, folded BN, bias_sc).
 * out [N,H,W,256] fp16; t1_next [N,H,W,64] fp16 when has_next.  Any H, W; tensors < 2 GiB.  The intermediate t2, the shortcut
 * convolution's output and the re-read of `out` by the next conv1 never touch HBM.
 * pe_bneck64_pack: w2 [64][3][3][64], w3 [256][64], wsc [256][64] or NULL, w1n [64][256] or NULL (fp16, the layouts
 * pe_conv2d_nhwc_f16 takes) -> one fragment-ordered stream of pe_bneck64_packed_bytes(wsc != NULL, w1n != NULL) bytes. */
size_t pe_bneck64_packed_bytes(int32_t has_shortcut_conv, int32_t has_next);
int pe_bneck64_pack(const void* w2, const void* w3, const void* wsc, const void* w1n, void* packed, void* stream);
int pe_bneck64_f16(const void* t1, const void* shortcut_src, const void* packed, const float* bias2, const float* bias3,
                   const float* bias_sc, const float* bias1n, void* out, void* t1_next, int32_t N, int32_t H, int32_t W,
                   int32_t has_shortcut_conv, int32_t has_next, void* stream);
int pe_conv_wd_pack_head(const void* head_weight, void* packed, int32_t rows, int32_t C, void* stream);
int pe_conv3x3_wd_rpn_head_f16(const void* input, const void* packed_weight, const float* bias, const void* packed_head,
                               const float* head_bias16, float* head_out, int32_t N, int32_t H, int32_t W, int32_t Cin,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Front-end layout kernels.
 * pe_preprocess_pack: one image -> normalised, zero-padded NHWC4 fp16 (optionally bilinear-resized first).
 *   Replaces GeneralizedRCNN.preprocess_image (modeling/meta_arch/rcnn.py:269-286), ImageList.from_tensors
 *   (structures/image_list.py:51-102) and, when dst size != src size, ResizeTransform.apply_image
 *   (data/transforms/transform.py:81-98; half-pixel bilinear, uint8 sources rounded back to integers: this is the
 *   stand-in for the cv2.resize branch of the 4- / 6-channel inputs, parity unpinned; 3-channel uint8 images use
 *   pe_preprocess_pack_pil_u8 below, which is Pillow-exact).
 *   src_kind 0: HWC uint8, 1: HWC float32, 2: CHW float32.  Source channels [ch0, ch0+nch) -> output
 *   channels 0..nch-1 (flip_rgb reverses the first three); mean/std are HOST arrays of length nch.
 * pe_maxpool3x3s2_nhwc: F.max_pool2d(x, 3, 2, 1) of BasicStem.forward (modeling/backbone/resnet.py:383).
 * pe_subsample2_nhwc:   LastLevelMaxPool (modeling/backbone/fpn.py:166-178) = x[:, ::2, ::2].
 * ------------------------------------------------------------------------------------------- */
int pe_preprocess_pack(const void* src, int32_t src_kind, int32_t src_h, int32_t src_w, int32_t src_c,
                       int32_t ch0, int32_t nch, int32_t flip_rgb, int32_t dst_h, int32_t dst_w,
                       int32_t pad_h, int32_t pad_w, const float* mean_host, const float* std_host,
                       void* dst, void* stream);
/* Same for num_images equally sized images stored back to back (src [N,...], dst [N,pad_h,pad_w,4]): one launch. */
int pe_preprocess_pack_batch(const void* src, int32_t num_images, int32_t src_kind, int32_t src_h, int32_t src_w,
                             int32_t src_c, int32_t ch0, int32_t nch, int32_t flip_rgb, int32_t dst_h, int32_t dst_w,
                             int32_t pad_h, int32_t pad_w, const float* mean_host, const float* std_host, void* dst,
                             void* stream);
/* The same with Pillow's EXACT bilinear resampler for uint8 sources [N,src_h,src_w,src_c] - the reference resizes
 * 3-channel images with Image.fromarray(img.astype(uint8)).resize((w, h), BILINEAR) (data/transforms/transform.py:92-97):
 * horizontal pass rounded to uint8, then vertical pass, 22-bit fixed-point weights.  xtab [dst_w, 2 + xk] /
 * ytab [dst_h, 2 + yk] DEVICE int32 tables (first tap, tap count, weights) of libImaging/Resample.c
 * precompute_coeffs + normalize_coeffs_8bpc (host helper: proben_amd.data.pil_bilinear_tables). */
int pe_preprocess_pack_pil_u8(const void* src, int32_t num_images, int32_t src_h, int32_t src_w, int32_t src_c,
                              int32_t ch0, int32_t nch, int32_t flip_rgb, int32_t dst_h, int32_t dst_w, int32_t pad_h,
                              int32_t pad_w, const float* mean_host, const float* std_host, const int32_t* xtab,
                              int32_t xk, const int32_t* ytab, int32_t yk, void* dst, void* stream);
int pe_maxpool3x3s2_nhwc(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int pe_subsample2_nhwc(const void* in, void* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* Fused BasicStem.forward (modeling/backbone/resnet.py:375-384): conv 7x7 / 2 / pad 3 with the frozen BN folded
 * + ReLU + max_pool2d(3, 2, 1) in one pass: x [N,H,W,4] fp16 (H, W multiples of 4) -> out [N,H/4,W/4,64] fp16.
 * w_packed [64,7,8,4] fp16 where tap t multiplies input column 2*c - 4 + t (tap 0 is zero); bias fp32 [64]. */
int pe_stem_conv7x7_maxpool_f16(const void* x, const void* w_packed, const float* bias, void* out, int32_t N,
                                int32_t H, int32_t W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched class-aware greedy NMS (float32).  Replaces detectron2.layers.batched_nms
 * (layers/nms.py:20-37) -> torchvision.ops.boxes.batched_nms / nms (torchvision 0.13.0), call sites
 * proposal_generator/rpn_outputs.py:147, roi_heads/fast_rcnn.py:130, demo/FLIR/demo_probEn.py:64.
 *   boxes [B,n_max,4], scores [B,n_max], idxs [B,n_max] (class / level id, may be NULL; any int32 - ids in [0, 2^18) get
 *   the per-class fast path: tiles of the suppression matrix between different classes are never computed),
 *   counts [B] rows used per image (NULL = n_max), valid [B,n_max] optional row mask.
 *   mode 0: coordinate trick (boxes + idx*(max+1)), mode 1: suppress only within equal idx ("vanilla").
 *   Mode 0 is evaluated per class while every live coordinate of the image is >= 0 (then the bands cannot meet); an image
 *   with a negative coordinate is compared all-pairs on the shifted boxes, so a box reaching below -1 suppresses - and is
 *   suppressed by - the neighbouring class exactly as torchvision's trick does.
 *   out_keep [B,max_out] input-row indices in score-descending order (ties: lower index first),
 *   out_counts [B].  scratch: pe_nms_scratch_bytes(B, n_max) bytes of device memory.
 * ------------------------------------------------------------------------------------------- */
#define PE_NMS_TRICK 0
#define PE_NMS_CLASS 1
size_t pe_nms_scratch_bytes(int32_t B, int32_t n_max);
int pe_nms_batched(const float* boxes, const float* scores, const int32_t* idxs, const int32_t* counts,
                   const uint8_t* valid, int32_t B, int32_t n_max, float iou_thresh, int32_t mode,
                   int32_t max_out, int32_t* out_keep, int32_t* out_counts, void* scratch,
                   size_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RPN proposal selection: per (image, level) top-k of the objectness logits, decode of the survivors
 * against analytically generated anchors, finite / clip-to-unpadded-size / non-empty flags.
 * Replaces DefaultAnchorGenerator (modeling/anchor_generator.py:43-56,130-199), RPNOutputs.predict_proposals
 * / predict_objectness_logits (proposal_generator/rpn_outputs.py:409-452), Box2BoxTransform.apply_deltas
 * (box_regression.py:73-110) and find_top_rpn_proposals' selection half (rpn_outputs.py:100-145).
 *   level_heads_host[l]: DEVICE pointer to the fused RPN head output of level l, fp32 [N*H*W, head_stride]
 *     (columns 0..2 objectness for anchors a=0..2, columns 3+4a..6+4a the deltas of anchor a); head_stride >= 16,
 *     a multiple of 4, pointers 16-byte aligned (the logits of a cell are fetched as one 16-byte load, once, into LDS);
 *   level_hw_host [L,2], level_stride_host [L], cell_anchors_host [L,3,4] are HOST arrays;
 *   image_hw [N,2] device int32 (h,w) of the unpadded resized images;
 *   outputs per image: cand_per_image = sum_l min(pre_nms_topk, H*W*3) rows, level-major, each level
 *   sorted by logit descending (ties: anchor index ascending).
 * pe_gather_boxes: rows keep[n, :counts[n]] of boxes/scores -> dense [N,max_out,...], zero padded.
 * ------------------------------------------------------------------------------------------- */
int pe_rpn_select_topk(const float* const* level_heads_host, const int32_t* level_hw_host,
                       const int32_t* level_stride_host, const float* cell_anchors_host,
                       int32_t num_levels, int32_t N, int32_t head_stride, int32_t pre_nms_topk,
                       const int32_t* image_hw, float scale_clamp, float* cand_boxes, float* cand_scores,
                       int32_t* cand_level, uint8_t* cand_valid, int32_t cand_per_image, void* scratch,
                       size_t scratch_bytes, void* stream);
/* Optional scratch for pe_rpn_select_topk: with >= this many bytes, levels larger than 16384 anchors are selected in
 * two exact stages (per-slice top-k on many CUs, then a merge) instead of one workgroup per (image, level). */
size_t pe_rpn_scratch_bytes(const int32_t* level_hw_host, int32_t num_levels, int32_t N);
int pe_gather_boxes(const float* boxes, const float* scores, const int32_t* keep, const int32_t* counts,
                    int32_t N, int32_t n_in, int32_t max_out, float* out_boxes, float* out_scores,
                    void* stream);

/* Stand-alone forms of two steps the detector kernels carry fused, because the reference exposes them as Python API:
 * pe_box2box_apply_deltas: Box2BoxTransform.apply_deltas (modeling/box_regression.py:73-110): deltas [N, 4k] and boxes
 *   [N,4] -> [N, 4k]; weights_host = (wx, wy, ww, wh); dw / dh are clamped to scale_clamp before exp.
 * pe_grid_anchors: DefaultAnchorGenerator.grid_anchors for ONE level (modeling/anchor_generator.py:43-56,120-146):
 *   cell_anchors [A,4] + shifts (x * stride, y * stride) (+ offset * stride) -> [H*W*A, 4] in (y, x, anchor) order. */
int pe_box2box_apply_deltas(const float* deltas, const float* boxes, int32_t N, int32_t k, const float* weights_host,
                            float scale_clamp, float* out, void* stream);
int pe_grid_anchors(const float* cell_anchors, int32_t num_cell_anchors, int32_t H, int32_t W, int32_t stride,
                    float offset, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused SGD-with-momentum step over flat fp32 buffers (training half).  Replaces the per-parameter loop of the
 * torch.optim.SGD that solver/build.py:93-133 builds (momentum, per-group lr / weight decay, dampening 0, no Nesterov) behind
 * DefaultTrainer (engine/defaults.py:250-262):  d = grad * grad_scale + weight_decay * p;  buf = first_step ? d : momentum * buf + d;
 * p -= lr * buf;  fp16_shadow (optional, [n] halfs) = (half)p in the same pass.  grad_scale folds DDP's 1 / world_size and the
 * inverse loss scale.  All pointers 16-byte aligned (shadow 8): one call per parameter group of a flat buffer.
 * ------------------------------------------------------------------------------------------- */
int pe_sgd_momentum_f32(float* params, const float* grads, float* momentum_buf, void* fp16_shadow, int64_t n, float lr,
                        float momentum, float weight_decay, float grad_scale, int32_t first_step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ROIAlign forward over NHWC feature maps.  Replaces roi_align_forward of detectron2._C
 * (layers/csrc/ROIAlign/ROIAlign.h:54-84, ROIAlign_cuda.cu:12-139,310-366; same arithmetic as
 * ROIAlign_cpu.cpp:22-218) and, with num_levels == 4, ROIPooler.forward + assign_boxes_to_levels
 * (modeling/poolers.py:13-44,180-235) in one launch.
 *   feats_host[l]: DEVICE pointers, feature l is [N, H_l, W_l, C]; dtype 0 = fp16, 1 = fp32 (in and out);
 *   rois: [R,5] (batch,x1,y1,x2,y2) when rois_have_batch_index, else boxes [N, per_image, 4] with
 *         counts[n] live rows per image (dead rows produce zeros);
 *   output [R, pooled_h, pooled_w, C]; out_level (optional) [R] assigned level or -1.
 * ------------------------------------------------------------------------------------------- */
int pe_roi_align_nhwc(const void* const* feats_host, const int32_t* feat_hw_host, const float* scales_host,
                      int32_t num_levels, int32_t N, int32_t C, int32_t dtype, const float* rois,
                      int32_t rois_have_batch_index, int32_t num_rois, int32_t per_image,
                      const int32_t* counts, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio,
                      int32_t aligned, void* output, int32_t* out_level, void* stream);

/* The same, boxes form only ([N, per_image, 4] + counts), with the PROCESSING order chosen by the library: one workgroup per
 * image first sorts its proposals by (FPN level, Morton code of the box centre) into order_workspace ([N * per_image] int32,
 * per_image <= 2048, else the order is left as given), and the ROIAlign workgroups walk that order, one contiguous stretch per
 * XCD.  Outputs land where pe_roi_align_nhwc puts them, bit for bit; only the feature traffic changes (proposals arrive in RPN
 * score order, scattered over the pyramid).  This is what ROIPooler's per-level nonzero/index_put grouping
 * (modeling/poolers.py:219-233) does for locality in the reference, without its four device synchronisations. */
int pe_roi_align_nhwc_sorted(const void* const* feats_host, const int32_t* feat_hw_host, const float* scales_host,
                             int32_t num_levels, int32_t N, int32_t C, int32_t dtype, const float* boxes, int32_t per_image,
                             const int32_t* counts, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio,
                             int32_t aligned, void* output, int32_t* out_level, int32_t* order_workspace, void* stream);

/* ROIAlign backward (training half).  Replaces roi_align_backward of detectron2._C
 * (layers/csrc/ROIAlign/ROIAlign.h:86-115, ROIAlign_cuda.cu:141-306,369-420; same arithmetic as ROIAlign_cpu.cpp:221-394)
 * behind _ROIAlign.backward (layers/roi_align.py:26-42) and, with num_levels == 4, the backward of ROIPooler's per-level
 * scatter.  Arguments as pe_roi_align_nhwc; grad_output [R, pooled_h, pooled_w, C] (dtype 0 = fp16, 1 = fp32);
 * grad_feats_host[l]: DEVICE pointers to fp32 [N, H_l, W_l, C] gradients that are ACCUMULATED into (the caller zeroes
 * them); rows beyond counts[n] contribute nothing.  fp32 atomic adds: the summation order over overlapping ROIs is not
 * defined (as in the reference's CUDA kernel). */
int pe_roi_align_backward_nhwc(const void* grad_output, int32_t dtype, const int32_t* feat_hw_host, const float* scales_host,
                               int32_t num_levels, int32_t N, int32_t C, const float* rois, int32_t rois_have_batch_index,
                               int32_t num_rois, int32_t per_image, const int32_t* counts, int32_t pooled_h,
                               int32_t pooled_w, int32_t sampling_ratio, int32_t aligned, float* const* grad_feats_host,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Box-head post-processing.  Replaces FastRCNNOutputs.inference / fast_rcnn_inference_single_image
 * (modeling/roi_heads/fast_rcnn.py:43-147,345-360,417-452) and detector_postprocess
 * (modeling/postprocessing.py:8-38).  head: fp32 [N*per_image, head_stride] with columns
 * [0,K] logits, [K+1,5K] deltas (class-major), 5K+1 log-variance.
 *   pe_boxhead_candidates: softmax, decode (weights reg_weights_host[4]), finite mask, clip to image_hw,
 *     score > thresh -> candidates in (row, class) order: boxes, scores, class, (filtered row, original row).
 *     At most cand_max candidates per image are kept (proposal-row order); cand_total (optional, may be NULL) receives
 *     the uncapped number so a caller can detect - and refuse - an overflow instead of losing detections silently.
 *   (caller runs pe_nms_batched over the candidates, class-aware, max_out = max_det)
 *   pe_boxhead_finalize: gathers the kept candidates' fields (incl. the reference's Q3/Q4 index quirks;
 *     fix_vars != 0 pairs each detection with its own proposal's variance), rescales to out_hw, clips,
 *     drops empty boxes.
 * ------------------------------------------------------------------------------------------- */
int pe_boxhead_candidates(const float* head, int32_t head_stride, int32_t N, int32_t per_image,
                          int32_t num_classes, const int32_t* prop_counts, const float* proposals,
                          const int32_t* image_hw, const float* reg_weights_host, float scale_clamp,
                          float score_thresh, int32_t cand_max, float* cand_boxes, float* cand_scores,
                          int32_t* cand_class, int32_t* cand_rows, int32_t* cand_counts, int32_t* cand_total,
                          float* probs, void* stream);
int pe_boxhead_finalize(const float* head, int32_t head_stride, int32_t N, int32_t per_image,
                        int32_t num_classes, int32_t cand_max, int32_t max_det, int32_t fix_vars,
                        const float* probs, const float* cand_boxes, const float* cand_scores,
                        const int32_t* cand_class, const int32_t* cand_rows, const int32_t* keep,
                        const int32_t* keep_counts, const int32_t* image_hw, const int32_t* out_hw,
                        float* det_boxes, float* det_scores, int32_t* det_classes, float* det_logits,
                        float* det_probs, float* det_vars, int32_t* det_rows, int32_t* det_counts,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Native COCO bbox evaluator (HOST code, multithreaded; no GPU involved).  Replaces COCOeval.evaluate /
 * computeIoU / evaluateImg / accumulate of the reference's vendored pycocotools
 * (detectron2/pycocotools/cocoeval.py:85-191, 236-421, Params :500-536) and pycocotools 2.0.4 `_mask.iou`
 * (bbIou) for iouType "bbox"; caller: FLIR_evaluation.py:496-563 (_evaluate_predictions_on_coco).
 *   Ground truth rows (annotation order): gt_img / gt_cat = DENSE indices into the sorted image-id / category-id
 *   lists, gt_box [n_gt,4] xywh float64, gt_area, gt_crowd (uint8), gt_id (annotation ids; an id of 0 counts
 *   as "unmatched", like the reference).  Detection rows (result order; detection id = row + 1, area = w*h):
 *   dt_img, dt_cat, dt_box [n_dt,4] xywh, dt_score.
 *   iou_thrs [T], rec_thrs [R], max_dets [M] ascending, area_rng [A,2].  num_threads <= 0: hardware concurrency (<= 32).
 *   Out: precision [T,R,K,A,M] and recall [T,K,A,M] float64, -1 where the reference leaves -1.
 * ------------------------------------------------------------------------------------------- */
int pe_cocoeval_bbox(const int32_t* gt_img, const int32_t* gt_cat, const double* gt_box, const double* gt_area,
                     const uint8_t* gt_crowd, const int64_t* gt_id, int64_t n_gt, const int32_t* dt_img,
                     const int32_t* dt_cat, const double* dt_box, const double* dt_score, int64_t n_dt,
                     int32_t n_imgs, int32_t n_cats, const double* iou_thrs, int32_t T, const double* rec_thrs,
                     int32_t R, const int32_t* max_dets, int32_t M, const double* area_rng, int32_t A,
                     int32_t num_threads, double* precision, double* recall);

#ifdef __cplusplus
}
#endif
#endif /* PROBEN_HIP_H */