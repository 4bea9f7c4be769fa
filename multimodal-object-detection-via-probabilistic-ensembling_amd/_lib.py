"""ctypes binding of libproben_hip.so (include/proben_hip.h).  Fails loudly."""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libproben_hip.so")
_lib = None

c_void_p, c_int, c_double, c_float = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double, ctypes.c_float
c_size_t = ctypes.c_size_t

# name -> argtypes (restype is always int unless listed in _RESTYPE)
SIGNATURES = {
    "pe_version": [],
    "pe_proben_pack_detections": [c_void_p] * 6 + [c_int] * 6 + [c_void_p] * 9,
    "pe_proben_fuse_batch": [c_void_p] * 8 + [c_int] * 5 + [c_double] * 3 + [c_void_p] * 5 + [c_void_p],
    "pe_conv2d_nhwc_f16": [c_void_p] * 5 + [c_int] * 14 + [c_void_p],
    "pe_conv_wd_supported": [c_int] * 6,
    "pe_conv_wd_pack_weights": [c_void_p] * 2 + [c_int] * 3 + [c_void_p],
    "pe_conv3x3_wd_f16": [c_void_p] * 4 + [c_int] * 7 + [c_void_p],
    "pe_conv_wd_pack_tail": [c_void_p] * 2 + [c_int] * 2 + [c_void_p],
    "pe_bottleneck_tail_wd_f16": [c_void_p] * 7 + [c_int] * 5 + [c_void_p],
    "pe_bneck64_packed_bytes": [c_int] * 2,
    "pe_bneck64_pack": [c_void_p] * 6,
    "pe_bneck64_f16": [c_void_p] * 9 + [c_int] * 5 + [c_void_p],
    "pe_conv_wd_pack_head": [c_void_p] * 2 + [c_int] * 2 + [c_void_p],
    "pe_conv3x3_wd_rpn_head_f16": [c_void_p] * 6 + [c_int] * 4 + [c_void_p],
    "pe_preprocess_pack": [c_void_p] + [c_int] * 11 + [c_void_p] * 4,
    "pe_preprocess_pack_batch": [c_void_p] + [c_int] * 12 + [c_void_p] * 4,
    "pe_preprocess_pack_pil_u8": [c_void_p] + [c_int] * 11 + [c_void_p] * 2 + [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 2,
    "pe_maxpool3x3s2_nhwc": [c_void_p] * 2 + [c_int] * 4 + [c_void_p],
    "pe_subsample2_nhwc": [c_void_p] * 2 + [c_int] * 4 + [c_void_p],
    "pe_stem_conv7x7_maxpool_f16": [c_void_p] * 4 + [c_int] * 3 + [c_void_p],
    "pe_nms_scratch_bytes": [c_int, c_int],
    "pe_nms_batched": [c_void_p] * 5 + [c_int, c_int, c_float, c_int, c_int] + [c_void_p] * 3 + [c_size_t, c_void_p],
    "pe_rpn_select_topk": [c_void_p] * 4 + [c_int] * 4 + [c_void_p, c_float] + [c_void_p] * 4 + [c_int, c_void_p, c_size_t, c_void_p],
    "pe_rpn_scratch_bytes": [c_void_p, c_int, c_int],
    "pe_box2box_apply_deltas": [c_void_p] * 2 + [c_int] * 2 + [c_void_p, c_float, c_void_p, c_void_p],
    "pe_grid_anchors": [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p],
    "pe_gather_boxes": [c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 3,
    "pe_roi_align_nhwc": [c_void_p] * 3 + [c_int] * 4 + [c_void_p] + [c_int] * 3 + [c_void_p] + [c_int] * 4 + [c_void_p] * 3,
    "pe_roi_align_nhwc_sorted": [c_void_p] * 3 + [c_int] * 4 + [c_void_p] + [c_int] + [c_void_p] + [c_int] * 4 + [c_void_p] * 4,
    "pe_sgd_momentum_f32": [c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_float] * 4 + [c_int, c_void_p],
    "pe_roi_align_backward_nhwc": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "pe_boxhead_candidates": [c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 4 + [c_float, c_float, c_int] + [c_void_p] * 8,
    "pe_cocoeval_bbox": [c_void_p] * 6 + [ctypes.c_int64] + [c_void_p] * 4 + [ctypes.c_int64, c_int, c_int, c_void_p, c_int,
                         c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "pe_boxhead_finalize": [c_void_p] + [c_int] * 7 + [c_void_p] * 18,
}
_RESTYPE = {"pe_last_error": ctypes.c_char_p, "pe_nms_scratch_bytes": ctypes.c_size_t, "pe_rpn_scratch_bytes": ctypes.c_size_t,
             "pe_bneck64_packed_bytes": ctypes.c_size_t}


# exported for tests/ and scripts/ only (csrc/test_hooks.h) - not in include/proben_hip.h
TEST_HOOKS = {"pe_test_set_conv_policy": [c_int, c_int], "pe_test_set_wd9_mode": [c_int], "pe_test_wd9_takes": [c_int] * 5, "pe_test_wd9_head_takes": [c_int] * 3, "pe_test_set_wd9_wgs": [c_int, c_int], "pe_test_set_ring_wgs": [c_int], "pe_test_set_roi_fast": [c_int], "pe_test_set_nms_presorted": [c_int]}


class HipLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library; raises HipLibraryError when it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  proben_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.pe_last_error.restype = ctypes.c_char_p
        L.pe_last_error.argtypes = []
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, ctypes.c_int)
        _lib = L
    return _lib


LAB_HOOKS = {"pe_test_set_ring_ablation": [c_int]}     # exported by the LAB build only (`python -m proben_amd.build --lab`, -DPE_LAB)


def test_hooks():
    """The library with the measurement hooks of csrc/test_hooks.h typed (tests/ and scripts/ only)."""
    L = lib()
    for name, args in LAB_HOOKS.items():
        if hasattr(L, name):
            getattr(L, name).argtypes, getattr(L, name).restype = args, ctypes.c_int
    for name, args in TEST_HOOKS.items():
        fn = getattr(L, name)
        fn.argtypes, fn.restype = args, ctypes.c_int
    return L


def check(status, what):
    if status != 0:
        raise HipLibraryError(f"{what} failed ({status}): {lib().pe_last_error().decode()}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipLibraryError("proben_amd kernels need device tensors (MODEL.DEVICE=cuda); got a CPU tensor. "
                                  "There is no CPU fallback in the product path.")


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
