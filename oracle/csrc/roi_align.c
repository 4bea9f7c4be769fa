/* Oracle (test infrastructure only): ROIAlign forward, float32, NCHW - a plain-C restatement of
 * the arithmetic of the reference's CPU kernel
 *   detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:22-114  (sample positions / bilinear weights)
 *   detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:116-218 (per-ROI average pooling)
 * Same operation order in float (T = float): start + ph*bin + (iy+.5f)*bin/grid; samples outside
 * [-1, H] x [-1, W] contribute 0; coordinates clamped at 0 and at the last row/column;
 * output = sum / max(grid_h*grid_w, 1).  Pinned by tests/test_roi_align.py:26-39's tables
 * (tests/golden/roi_align_ref_tables.json) and by fixtures produced by the compiled reference
 * kernel itself (tests/golden/gen_detector.py).  Build: oracle/build_c.py (gcc -O2 -ffp-contract=off).
 */
#include <math.h>
#include <stdlib.h>

/* rois: [K,5] = (batch_index, x1, y1, x2, y2); returns 0, or -1 on a negative-size ROI with aligned != 0
 * (the reference asserts there). */
int oracle_roi_align_forward(const float* input, int N, int C, int H, int W, const float* rois, int K,
                             float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                             int aligned, float* output) {
    (void)N;
    for (int n = 0; n < K; ++n) {
        const float* r = rois + (size_t)n * 5;
        const int b = (int)r[0];
        const float offset = aligned ? 0.5f : 0.0f;
        const float start_w = r[1] * spatial_scale - offset;
        const float start_h = r[2] * spatial_scale - offset;
        const float end_w = r[3] * spatial_scale - offset;
        const float end_h = r[4] * spatial_scale - offset;
        float roi_w = end_w - start_w;
        float roi_h = end_h - start_h;
        if (aligned) {
            if (!(roi_w >= 0 && roi_h >= 0)) return -1;
        } else {
            roi_w = roi_w > 1.f ? roi_w : 1.f;
            roi_h = roi_h > 1.f ? roi_h : 1.f;
        }
        const float bin_h = roi_h / (float)pooled_h;
        const float bin_w = roi_w / (float)pooled_w;
        const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / pooled_h);
        const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / pooled_w);
        const int cnt = grid_h * grid_w > 1 ? grid_h * grid_w : 1;
        const float count = (float)cnt;
        const size_t ns = (size_t)pooled_h * pooled_w * (grid_h > 0 ? grid_h : 0) * (grid_w > 0 ? grid_w : 0);
        int* pos = (int*)malloc((ns ? ns : 1) * 4 * sizeof(int));
        float* wgt = (float*)malloc((ns ? ns : 1) * 4 * sizeof(float));
        size_t k = 0;
        for (int ph = 0; ph < pooled_h; ++ph)
            for (int pw = 0; pw < pooled_w; ++pw)
                for (int iy = 0; iy < grid_h; ++iy) {
                    const float yy = start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)grid_h;
                    for (int ix = 0; ix < grid_w; ++ix, ++k) {
                        const float xx = start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)grid_w;
                        float x = xx, y = yy;
                        if (y < -1.0 || y > H || x < -1.0 || x > W) {
                            pos[4 * k] = pos[4 * k + 1] = pos[4 * k + 2] = pos[4 * k + 3] = 0;
                            wgt[4 * k] = wgt[4 * k + 1] = wgt[4 * k + 2] = wgt[4 * k + 3] = 0.f;
                            continue;
                        }
                        if (y <= 0) y = 0;
                        if (x <= 0) x = 0;
                        int y_low = (int)y, x_low = (int)x, y_high, x_high;
                        if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
                        if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
                        const float ly = y - y_low, lx = x - x_low;
                        const float hy = 1.f - ly, hx = 1.f - lx;
                        pos[4 * k] = y_low * W + x_low;   wgt[4 * k] = hy * hx;
                        pos[4 * k + 1] = y_low * W + x_high;  wgt[4 * k + 1] = hy * lx;
                        pos[4 * k + 2] = y_high * W + x_low;  wgt[4 * k + 2] = ly * hx;
                        pos[4 * k + 3] = y_high * W + x_high; wgt[4 * k + 3] = ly * lx;
                    }
                }
        for (int c = 0; c < C; ++c) {
            const float* in = input + ((size_t)b * C + c) * H * W;
            float* out = output + ((size_t)n * C + c) * pooled_h * pooled_w;
            size_t q = 0;
            for (int p = 0; p < pooled_h * pooled_w; ++p) {
                float acc = 0.f;
                for (int s = 0; s < grid_h * grid_w; ++s, ++q)
                    acc += wgt[4 * q] * in[pos[4 * q]] + wgt[4 * q + 1] * in[pos[4 * q + 1]] +
                           wgt[4 * q + 2] * in[pos[4 * q + 2]] + wgt[4 * q + 3] * in[pos[4 * q + 3]];
                out[p] = acc / count;
            }
        }
        free(pos);
        free(wgt);
    }
    return 0;
}

/* Oracle: ROIAlign backward (the adjoint of the forward above), float32, NCHW - restates
 *   detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:221-285 (weights of one sample; a sample outside [-1, H] x [-1, W]
 *   contributes nothing) and :287-394 (every pooled element hands grad * w / count to the four corners of each of its
 *   samples, visited in (roi, channel, ph, pw, iy, ix) order with plain float adds).
 * grad_output [K,C,ph,pw] contiguous, grad_input [N,C,H,W] is ACCUMULATED into (the caller zeroes it).
 * Pinned by tests/test_oracle_roi_align_backward.py: on small shapes the matrix of the (reference-pinned) forward is
 * probed column by column and this routine must be its transpose. */
int oracle_roi_align_backward(const float* grad_output, int N, int C, int H, int W, const float* rois, int K,
                              float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                              float* grad_input) {
    (void)N;
    for (int n = 0; n < K; ++n) {
        const float* r = rois + (size_t)n * 5;
        const int b = (int)r[0];
        const float offset = aligned ? 0.5f : 0.0f;
        const float start_w = r[1] * spatial_scale - offset, start_h = r[2] * spatial_scale - offset;
        const float end_w = r[3] * spatial_scale - offset, end_h = r[4] * spatial_scale - offset;
        float roi_w = end_w - start_w, roi_h = end_h - start_h;
        if (aligned) {
            if (!(roi_w >= 0 && roi_h >= 0)) return -1;
        } else {
            roi_w = roi_w > 1.f ? roi_w : 1.f;
            roi_h = roi_h > 1.f ? roi_h : 1.f;
        }
        const float bin_h = roi_h / (float)pooled_h, bin_w = roi_w / (float)pooled_w;
        const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / pooled_h);
        const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / pooled_w);
        const float count = (float)(grid_h * grid_w);   /* the backward divides by the raw product (0 samples -> no adds) */
        for (int c = 0; c < C; ++c) {
            float* gin = grad_input + ((size_t)b * C + c) * H * W;
            const float* gout = grad_output + ((size_t)n * C + c) * pooled_h * pooled_w;
            for (int ph = 0; ph < pooled_h; ++ph)
                for (int pw = 0; pw < pooled_w; ++pw) {
                    const float g = gout[ph * pooled_w + pw];
                    for (int iy = 0; iy < grid_h; ++iy) {
                        const float yy = start_h + ph * bin_h + (float)(iy + .5f) * bin_h / (float)grid_h;
                        for (int ix = 0; ix < grid_w; ++ix) {
                            const float xx = start_w + pw * bin_w + (float)(ix + .5f) * bin_w / (float)grid_w;
                            float x = xx, y = yy;
                            if (y < -1.0 || y > H || x < -1.0 || x > W) continue;
                            if (y <= 0) y = 0;
                            if (x <= 0) x = 0;
                            int y_low = (int)y, x_low = (int)x, y_high, x_high;
                            if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
                            if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
                            const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
                            gin[y_low * W + x_low] += g * (hy * hx) / count;
                            gin[y_low * W + x_high] += g * (hy * lx) / count;
                            gin[y_high * W + x_low] += g * (ly * hx) / count;
                            gin[y_high * W + x_high] += g * (ly * lx) / count;
                        }
                    }
                }
        }
    }
    return 0;
}
