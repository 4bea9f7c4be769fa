"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

ResizeTransform.apply_image for 3-channel images (detectron2/data/transforms/transform.py:81-98):
    Image.fromarray(img.astype(np.uint8)).resize((new_w, new_h), Image.BILINEAR)
i.e. Pillow's two-pass fixed-point resampler.  Pillow is a third-party dependency of the reference (pinned 9.2.0 in
probEn.yml) and not part of /root/reference; the algorithm below restates libImaging/Resample.c
(precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc - unchanged between
Pillow 3.x and 12.x) and is PINNED against the Pillow installed in the build container (12.2.0) by
tests/test_oracle_resize.py and tests/golden/pil_resize.npz:
  * per output coordinate: centre = (x + 0.5) * scale, support = max(scale, 1), taps xmin .. xmax-1 with triangle
    weights, normalised in float64, converted to 22-bit fixed point with round-half-away-from-zero;
  * horizontal pass over all source rows into a uint8 image, then the vertical pass (each pass rounds:
    (2^21 + sum) >> 22, clipped to 0..255).
The 4- and 6-channel inputs go through cv2.resize on float64 in the reference (transform.py:82-91); OpenCV is not
available here: `cv2_linear_resize_f64` restates its published INTER_LINEAR rule, PARITY UNPINNED."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """-> bounds [out,2] (xmin, count) int32 and fixed-point weights [out, ksize] int32 (Resample.c:precompute_coeffs)."""
    scale = filterscale = float(in_size) / float(out_size)
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            v = 1.0 - v if v < 1.0 else 0.0
            w[x] = v
            ww += v
        if ww != 0.0:
            for x in range(xmax):
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One resampling pass along `axis` of a uint8 [H, W, C] image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.int64)
    for o in range(bounds.shape[0]):
        xmin, n = bounds[o]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for i in range(n):
            acc += src[xmin + i] * int(kk[o, i])
        out[o] = acc >> PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def pil_bilinear_resize_u8(img, new_h, new_w):
    """img uint8 [H, W, C] -> uint8 [new_h, new_w, C] exactly like Image.fromarray(img).resize((new_w, new_h), BILINEAR)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    out = img
    if new_w != w:
        out = _pass(out, *pil_bilinear_coeffs(w, new_w), axis=1)   # horizontal first (ImagingResample)
    if new_h != h:
        out = _pass(out, *pil_bilinear_coeffs(h, new_h), axis=0)
    return out


def cv2_linear_resize_f64(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h)) with the default INTER_LINEAR on a floating-point [H, W, C] image - the
    reference's branch for 4- and 6-channel inputs (transform.py:82-91).  PARITY UNPINNED: OpenCV (4.6.0 in
    probEn.yml) is a third-party dependency that is neither in /root/reference nor installed here; this restates
    the published algorithm of imgproc/resize.cpp (resizeGeneric_ / HResizeLinear / VResizeLinear for CV_64F):
    fx = float((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx; taps left of the image or on / right of the last
    column collapse onto the edge pixel; float32 weights (1 - fx, fx); horizontal then vertical pass in float64."""
    img = np.asarray(img, dtype=np.float64)
    h, w = img.shape[:2]

    def taps(n_in, n_out):
        scale = float(n_in) / float(n_out)
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo], f[lo] = 0, 0.0
        hi = s >= n_in - 1
        s[hi], f[hi] = n_in - 1, 0.0
        return s, np.minimum(s + 1, n_in - 1), (np.float32(1.0) - f).astype(np.float64), f.astype(np.float64)

    x0, x1, ax0, ax1 = taps(w, new_w)
    y0, y1, ay0, ay1 = taps(h, new_h)
    rows = img[:, x0] * ax0[None, :, None] + img[:, x1] * ax1[None, :, None]          # horizontal pass
    return rows[y0] * ay0[:, None, None] + rows[y1] * ay1[:, None, None]             # vertical pass
