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
 * One wavefront per image; max_rows_per_image sizes the per-wavefront LDS slab (<= 2048).
 * ------------------------------------------------------------------------------------------- */
int pe_proben_fuse_batch(const double* boxes,     /* [Ntot,4] xyxy */
                         const double* scores,    /* [Ntot] */
                         const double* probs,     /* [Ntot,K] */
                         const double* variances, /* [Ntot] */
                         const int32_t* classes,  /* [Ntot] */
                         const int32_t* offsets,  /* [B+1] */
                         int32_t num_images, int32_t num_classes, int32_t max_rows_per_image,
                         int32_t score_mode, int32_t box_mode, double iou_thresh,
                         double frame_w, double frame_h, /* class-band shift: 640, 512 */
                         double* out_boxes,              /* [Ntot,4] */
                         float* out_scores,              /* [Ntot] */
                         float* out_classes,             /* [Ntot] */
                         int32_t* out_keep,              /* [Ntot] row index (image-local) of each pivot */
                         int32_t* out_counts,            /* [B] */
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PROBEN_HIP_H */
