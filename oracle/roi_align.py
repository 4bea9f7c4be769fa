"""Oracle ROIAlign forward (float32, NCHW) - ctypes wrapper over oracle/csrc/roi_align.c,
which restates detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:22-218.  Test infrastructure only."""
import ctypes

import torch

from .build_c import build

_lib = None


def _L():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_roi_align_forward.restype = ctypes.c_int
    return _lib


def roi_align_forward(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned):
    """Same signature as the reference's `_C.roi_align_forward` (layers/csrc/ROIAlign/ROIAlign.h:54-84)."""
    x = input.detach().contiguous().float()
    r = rois.detach().contiguous().float()
    N, C, H, W = x.shape
    K = r.shape[0]
    out = torch.zeros((K, C, pooled_h, pooled_w), dtype=torch.float32)
    if K == 0:
        return out
    st = _L().oracle_roi_align_forward(
        ctypes.c_void_p(x.data_ptr()), N, C, H, W, ctypes.c_void_p(r.data_ptr()), K,
        ctypes.c_float(spatial_scale), pooled_h, pooled_w, sampling_ratio, int(bool(aligned)),
        ctypes.c_void_p(out.data_ptr()))
    if st != 0:
        raise RuntimeError("ROIs in ROIAlign cannot have non-negative size!")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_h, pooled_w, batch_size, channels, height, width, sampling_ratio, aligned):
    """Same signature as the reference's `_C.roi_align_backward` (layers/csrc/ROIAlign/ROIAlign.h:86-115): grad [K,C,ph,pw]
    -> grad_input [N,C,H,W] float32."""
    g = grad.detach().contiguous().float()
    r = rois.detach().contiguous().float()
    K = r.shape[0]
    out = torch.zeros((batch_size, channels, height, width), dtype=torch.float32)
    if K == 0:
        return out
    L = _L()
    L.oracle_roi_align_backward.restype = ctypes.c_int
    st = L.oracle_roi_align_backward(
        ctypes.c_void_p(g.data_ptr()), batch_size, channels, height, width, ctypes.c_void_p(r.data_ptr()), K,
        ctypes.c_float(spatial_scale), pooled_h, pooled_w, sampling_ratio, int(bool(aligned)), ctypes.c_void_p(out.data_ptr()))
    if st != 0:
        raise RuntimeError("ROIs in ROIAlign do not have non-negative size!")
    return out
