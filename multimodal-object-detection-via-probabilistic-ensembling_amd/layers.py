"""Python op surface kept drop-in with the reference's `detectron2.layers`
(batched_nms: layers/nms.py:20-37, ROIAlign: layers/roi_align.py:51-96) plus thin typed
wrappers over the C-ABI kernels used by the detector.  No CPU fallbacks."""
import ctypes

import numpy as np
import torch

from . import _lib

_scratch = {}
PROFILE = None  # bench.py's roofline leg sets this to a list and gets one record per conv launch


def _get_scratch(nbytes, device):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _scratch[key] = buf
    return buf


# ------------------------------------------------------------------------------------------------
# NMS
# ------------------------------------------------------------------------------------------------
def nms_batched_raw(boxes, scores, idxs, counts, valid, iou_threshold, mode, max_out):
    """boxes [B,n,4] f32, scores [B,n] f32, idxs [B,n] i32|None, counts [B] i32|None, valid [B,n] u8|None.
    Returns keep [B,max_out] i32 (input row ids, score-descending), keep_counts [B] i32."""
    _lib.require_cuda(boxes, scores)
    B, n = scores.shape
    dev = boxes.device
    keep = torch.empty((B, max_out), dtype=torch.int32, device=dev)
    kcnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    if B == 0:
        return keep, kcnt
    L = _lib.lib()
    nbytes = L.pe_nms_scratch_bytes(B, max(n, 1))
    scratch = _get_scratch(nbytes, dev)
    st = L.pe_nms_batched(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(idxs), _lib.ptr(counts), _lib.ptr(valid),
                          B, n, float(iou_threshold), int(mode), int(max_out), _lib.ptr(keep), _lib.ptr(kcnt),
                          _lib.ptr(scratch), scratch.numel(), _lib.stream())
    _lib.check(st, "pe_nms_batched")
    return keep, kcnt


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Drop-in for detectron2.layers.batched_nms (layers/nms.py:20-37): boxes [N,4], scores [N],
    idxs [N] -> int64 keep indices sorted by score descending.  torchvision's dispatch rule is kept:
    coordinate trick up to 20000 box elements on a GPU, one NMS per class beyond that.  Arbitrary boxes are fine: with
    negative coordinates the trick's class bands can overlap and the kernel then compares all pairs like torchvision
    (csrc/nms.hip, "generic" images)."""
    assert boxes.shape[-1] == 4
    _lib.require_cuda(boxes, scores, idxs)
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    mode = 1 if boxes.numel() > 20000 else 0
    b = boxes.detach().float().contiguous().view(1, n, 4)
    s = scores.detach().float().contiguous().view(1, n)
    # class ids may arrive as float tensors (demo_probEn.py:57): torchvision does idxs.to(boxes)
    i = idxs.detach().to(torch.int32).contiguous().view(1, n)
    keep, cnt = nms_batched_raw(b, s, i, None, None, iou_threshold, mode, n)
    return keep[0, : int(cnt.item())].to(torch.int64)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms equivalent (re-exported by layers/nms.py:6)."""
    return batched_nms(boxes, scores, torch.zeros(len(scores), dtype=torch.int32, device=boxes.device), iou_threshold)


# ------------------------------------------------------------------------------------------------
# ROIAlign
# ------------------------------------------------------------------------------------------------
ROI_SORT = True    # A/B switch of bench.py --roi-sort (results are identical either way)


def roi_align_nhwc(feats, rois, *, scales, pooled, sampling_ratio=0, aligned=True, counts=None, per_image=0,
                   num_rois=None, out=None, want_levels=False, sort=None):
    """feats: list (1 or 4) of NHWC tensors (fp16 or fp32, same dtype/C); rois: [R,5] or boxes [N,per_image,4].
    sort (boxes form only): the workgroups take the ROIs in (level, Morton cell) order, one contiguous stretch per XCD
    (pe_roi_align_nhwc_sorted) - the same results at the same places, fewer feature bytes fetched."""
    f0 = feats[0]
    _lib.require_cuda(f0, rois)
    N, C = f0.shape[0], f0.shape[3]
    dtype = 0 if f0.dtype == torch.float16 else 1
    have_b = rois.dim() == 2 and rois.shape[1] == 5
    R = rois.shape[0] if have_b else (num_rois if num_rois is not None else rois.shape[0] * rois.shape[1])
    if out is None:
        out = torch.empty((R, pooled[0], pooled[1], C), dtype=f0.dtype, device=f0.device)
    lv = torch.empty((R,), dtype=torch.int32, device=f0.device) if want_levels else None
    nl = len(feats)
    ptrs = (ctypes.c_void_p * nl)(*[f.data_ptr() for f in feats])
    hw = (ctypes.c_int32 * (2 * nl))(*sum([[f.shape[1], f.shape[2]] for f in feats], []))
    sc = (ctypes.c_float * nl)(*scales)
    if (ROI_SORT if sort is None else sort) and not have_b and per_image > 0 and R == N * per_image:
        order = torch.empty((R,), dtype=torch.int32, device=f0.device)
        st = _lib.lib().pe_roi_align_nhwc_sorted(ptrs, hw, sc, nl, N, C, dtype, _lib.ptr(rois.contiguous()), int(per_image),
                                                 _lib.ptr(counts), pooled[0], pooled[1], int(sampling_ratio), int(bool(aligned)),
                                                 _lib.ptr(out), _lib.ptr(lv), _lib.ptr(order), _lib.stream())
        _lib.check(st, "pe_roi_align_nhwc_sorted")
        return (out, lv) if want_levels else out
    st = _lib.lib().pe_roi_align_nhwc(ptrs, hw, sc, nl, N, C, dtype, _lib.ptr(rois.contiguous()), int(have_b), R,
                                      int(per_image), _lib.ptr(counts), pooled[0], pooled[1], int(sampling_ratio),
                                      int(bool(aligned)), _lib.ptr(out), _lib.ptr(lv), _lib.stream())
    _lib.check(st, "pe_roi_align_nhwc")
    return (out, lv) if want_levels else out


def roi_align_backward_nhwc(grad_output, rois, feat_shapes, *, scales, sampling_ratio=0, aligned=True, counts=None, per_image=0,
                            num_rois=None):
    """Adjoint of roi_align_nhwc: grad_output [R,ph,pw,C] (fp16 or fp32) -> list of fp32 [N,H_l,W_l,C] gradients, one per
    level of `feat_shapes` (1 or 4 (N,H,W,C) tuples); training half of layers/roi_align.py:26-42 / poolers.py:180-235."""
    _lib.require_cuda(grad_output, rois)
    g = grad_output.contiguous()
    R, ph, pw, C = g.shape
    dtype = 0 if g.dtype == torch.float16 else 1
    if dtype == 1:
        g = g.float()
    have_b = rois.dim() == 2 and rois.shape[1] == 5
    N = feat_shapes[0][0]
    grads = [torch.zeros(tuple(fs), dtype=torch.float32, device=g.device) for fs in feat_shapes]
    nl = len(grads)
    ptrs = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in grads])
    hw = (ctypes.c_int32 * (2 * nl))(*sum([[fs[1], fs[2]] for fs in feat_shapes], []))
    sc = (ctypes.c_float * nl)(*scales)
    st = _lib.lib().pe_roi_align_backward_nhwc(_lib.ptr(g), dtype, hw, sc, nl, N, C, _lib.ptr(rois.contiguous()), int(have_b),
                                               R if num_rois is None else num_rois, int(per_image), _lib.ptr(counts), ph, pw,
                                               int(sampling_ratio), int(bool(aligned)), ptrs, _lib.stream())
    _lib.check(st, "pe_roi_align_backward_nhwc")
    return grads


class _ROIAlignFunction(torch.autograd.Function):
    """layers/roi_align.py:10-49 (`_ROIAlign`): forward and backward both run the gfx950 kernels (NHWC inside)."""

    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio, aligned):
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(output_size), spatial_scale, sampling_ratio, aligned, tuple(input.shape), input.dtype)
        C = input.shape[1]
        x = input.detach().permute(0, 2, 3, 1).contiguous()
        if x.dtype not in (torch.float16, torch.float32):
            x = x.float()
        pad = (-C) % (8 if x.dtype == torch.float16 else 4)
        if pad:
            x = torch.nn.functional.pad(x, (0, pad))
        out = roi_align_nhwc([x], rois.float(), scales=[spatial_scale], pooled=tuple(output_size), sampling_ratio=sampling_ratio, aligned=aligned)
        return out[..., :C].permute(0, 3, 1, 2).contiguous().to(input.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        output_size, spatial_scale, sampling_ratio, aligned, in_shape, in_dtype = ctx.cfg
        N, C, H, W = in_shape
        g = grad_output.detach().permute(0, 2, 3, 1).contiguous().float()
        pad = (-C) % 4
        if pad:
            g = torch.nn.functional.pad(g, (0, pad))
        (gin,) = roi_align_backward_nhwc(g, rois.float(), [(N, H, W, C + pad)], scales=[spatial_scale], sampling_ratio=sampling_ratio, aligned=aligned)
        return gin[..., :C].permute(0, 3, 1, 2).contiguous().to(in_dtype), None, None, None, None, None


class ROIAlign:
    """Drop-in for detectron2.layers.ROIAlign (layers/roi_align.py:51-96): NCHW input, rois [K,5],
    returns [K,C,ph,pw] in the input dtype; differentiable with respect to the input like the reference's autograd Function.
    (The detector itself stays in NHWC and calls roi_align_nhwc.)"""

    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned

    def __call__(self, input, rois):
        assert rois.dim() == 2 and rois.size(1) == 5
        _lib.require_cuda(input, rois)
        return _ROIAlignFunction.apply(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)

    forward = __call__

    def __repr__(self):
        return (f"ROIAlign(output_size={self.output_size}, spatial_scale={self.spatial_scale}, "
                f"sampling_ratio={self.sampling_ratio}, aligned={self.aligned})")


# ------------------------------------------------------------------------------------------------
# conv / gemm
# ------------------------------------------------------------------------------------------------
def conv2d_nhwc(x, weight, bias, *, kernel, stride=1, relu=False, residual=None, residual_mode=0,
                out=None, out_f32=False, cout_store=0, out_stride=0, cout=None):
    """x [N,H,W,Cin] fp16 NHWC; weight packed [Cout,KH,KW,Cin] fp16 ([Cout,8,8,4] for the 7x7 stem);
    bias fp32 [Cout] or None.  Returns NHWC fp16 (or fp32 [N,Ho,Wo,out_stride] when out_f32)."""
    _lib.require_cuda(x, weight)
    N, H, W, Cin = x.shape
    Cout = cout if cout is not None else weight.shape[0]
    if kernel == 1:
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    elif kernel == 3:
        Ho, Wo = H, W
    else:
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    if out is None:
        if out_f32:
            out = torch.empty((N, Ho, Wo, out_stride or Cout), dtype=torch.float32, device=x.device)
        else:
            out = torch.empty((N, Ho, Wo, out_stride or Cout), dtype=torch.float16, device=x.device)
    rh, rw = (residual.shape[1], residual.shape[2]) if residual is not None else (0, 0)
    st = _lib.lib().pe_conv2d_nhwc_f16(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(out),
                                       N, H, W, Cin, Cout, kernel, stride, int(relu), int(residual_mode), rh, rw,
                                       int(out_f32), int(cout_store), int(out_stride), _lib.stream())
    _lib.check(st, "pe_conv2d_nhwc_f16")
    if PROFILE is not None:  # bench.py roofline leg: remember the launch so it can be replayed back-to-back
        variant = conv_variant_name(N * Ho * Wo, Cout, kernel, Cin if kernel == 1 else 0, stride=stride, residual_mode=residual_mode,
                                    out_f32=out_f32, has_bias=bias is not None, cout_store=cout_store, out_stride=out_stride, in_pixels=N * H * W)
        cin_real = 3 if kernel == 7 else Cin
        shape = f"N{N} {H}x{W} Cin{Cin} Cout{Cout} k{kernel} s{stride} res{residual_mode} f32{int(out_f32)}"
        M = N * Ho * Wo
        in_elems = M * Cin if kernel == 1 else N * H * W * (3 if kernel == 7 else Cin)   # a strided 1x1 reads only its samples
        nbytes = (in_elems * 2 + weight.numel() * 2 + M * (cout_store or Cout) * (4 if out_f32 else 2)
                  + (M * Cout * 2 if residual_mode == 1 else (residual.numel() * 2 if residual_mode == 2 else 0)))
        PROFILE.append({"variant": variant, "shape": shape, "flops": 2.0 * N * Ho * Wo * Cout * kernel * kernel * cin_real,
                        "bytes": float(nbytes),
                        "replay": (lambda: conv2d_nhwc(x, weight, bias, kernel=kernel, stride=stride, relu=relu, residual=residual,
                                                       residual_mode=residual_mode, out=out, out_f32=out_f32,
                                                       cout_store=cout_store, out_stride=out_stride, cout=cout))})
    return out


def conv_wd_supported(kernel, stride, H, W, Cin, Cout):
    """True when the weights-direct kernel (csrc/conv_wd.h) takes this geometry."""
    return bool(_lib.lib().pe_conv_wd_supported(int(kernel), int(stride), int(H), int(W), int(Cin), int(Cout)))


def conv_wd_pack(weight):
    """[Cout,3,3,Cin] fp16 device tensor -> the fragment-ordered copy pe_conv3x3_wd_f16 streams (same size)."""
    _lib.require_cuda(weight)
    Cout, kh, kw, Cin = weight.shape
    assert kh == 3 and kw == 3 and weight.dtype == torch.float16 and weight.is_contiguous()
    packed = torch.empty(Cout * 9 * Cin, dtype=torch.float16, device=weight.device)
    _lib.check(_lib.lib().pe_conv_wd_pack_weights(_lib.ptr(weight), _lib.ptr(packed), Cout, Cin, 3, _lib.stream()),
               "pe_conv_wd_pack_weights")
    return packed


def conv3x3_wd(x, packed, bias, cout, *, relu=False, out=None, out_stride=0):
    """3x3 / stride 1 / pad 1 through the weights-direct kernel.  x [N,H,W,Cin] fp16 NHWC, packed = conv_wd_pack(w),
    bias fp32 [cout] (required).  Returns [N,H,W,out_stride or cout] fp16."""
    _lib.require_cuda(x, packed, bias)
    N, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((N, H, W, out_stride or cout), dtype=torch.float16, device=x.device)
    st = _lib.lib().pe_conv3x3_wd_f16(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(out), N, H, W, Cin, cout,
                                      int(relu), int(out_stride), _lib.stream())
    _lib.check(st, "pe_conv3x3_wd_f16")
    if PROFILE is not None:
        M = N * H * W
        wd9 = _lib.test_hooks().pe_test_wd9_takes(N, H, W, Cin, cout)      # which kernel generation took the launch (same bits either way)
        PROFILE.append({"variant": "conv3x3_wd9_kernel<8, 4, 5, 0>" if wd9 else "conv3x3_wd_kernel<1, 4, 4, 4, 0, 0>",
                        "shape": f"N{N} {H}x{W} Cin{Cin} Cout{cout} k3 s1 res0 f320",
                        "flops": 2.0 * M * cout * 9 * Cin, "bytes": float(M * Cin * 2 + cout * 9 * Cin * 2 + M * cout * 2),
                        "replay": (lambda: conv3x3_wd(x, packed, bias, cout, relu=relu, out=out, out_stride=out_stride))})
    return out


def conv_wd_pack_tail(weight2d):
    """conv3 weight [tail_cout, 256] fp16 (BN folded) -> fragment records in the fused bottleneck tail's stream order."""
    _lib.require_cuda(weight2d)
    cout, C = weight2d.shape
    assert weight2d.dtype == torch.float16 and weight2d.is_contiguous()
    packed = torch.empty(cout * C, dtype=torch.float16, device=weight2d.device)
    _lib.check(_lib.lib().pe_conv_wd_pack_tail(_lib.ptr(weight2d), _lib.ptr(packed), cout, C, _lib.stream()), "pe_conv_wd_pack_tail")
    return packed


def bottleneck_tail_wd(x, packed3x3, bias3x3, packed_tail, tail_bias, residual, tail_cout, out=None):
    """relu(conv3(relu(conv2(x))) + residual) of a BottleneckBlock in one launch.  x [N,H,W,Cin] fp16 (conv1's output),
    conv2: Cin -> 256, conv3: 256 -> tail_cout; residual [N,H,W,tail_cout] fp16 or None.  Returns [N,H,W,tail_cout] fp16."""
    _lib.require_cuda(x, packed3x3, bias3x3, packed_tail, tail_bias)
    N, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((N, H, W, tail_cout), dtype=torch.float16, device=x.device)
    if residual is not None:
        assert residual.is_contiguous() and tuple(residual.shape) == (N, H, W, tail_cout)
    st = _lib.lib().pe_bottleneck_tail_wd_f16(_lib.ptr(x), _lib.ptr(packed3x3), _lib.ptr(bias3x3), _lib.ptr(packed_tail), _lib.ptr(tail_bias),
                                              _lib.ptr(residual), _lib.ptr(out), N, H, W, Cin, int(tail_cout), _lib.stream())
    _lib.check(st, "pe_bottleneck_tail_wd_f16")
    if PROFILE is not None:
        M = N * H * W
        PROFILE.append({"variant": "conv3x3_wd_kernel<1, 4, 4, 4, 0, 2>",
                        "shape": f"N{N} {H}x{W} Cin{Cin} Cout256->{tail_cout} k3+k1 s1 res{int(residual is not None)} f320",
                        "flops": 2.0 * M * 256 * 9 * Cin + 2.0 * M * tail_cout * 256,
                        "bytes": float(M * Cin * 2 + 256 * 9 * Cin * 2 + tail_cout * 256 * 2 + M * tail_cout * 2 * (2 if residual is not None else 1)),
                        "replay": (lambda: bottleneck_tail_wd(x, packed3x3, bias3x3, packed_tail, tail_bias, residual, tail_cout, out=out))})
    return out


def bneck64_pack(w2, w3, wsc=None, w1n=None):
    """Fragment-ordered weight stream of `bneck64`: w2 [64,3,3,64], w3 [256,64] (or [256,1,1,64]), wsc [256,64] | None (first
    block's shortcut convolution), w1n [64,256] | None (the next block's conv1); all fp16, BN folded."""
    _lib.require_cuda(w2, w3)
    assert tuple(w2.shape) == (64, 3, 3, 64) and w3.numel() == 256 * 64 and w2.dtype == torch.float16
    for w in (w2, w3, wsc, w1n):
        assert w is None or (w.is_contiguous() and w.dtype == torch.float16 and w.numel() in (64 * 576, 256 * 64))
    nbytes = _lib.lib().pe_bneck64_packed_bytes(int(wsc is not None), int(w1n is not None))
    packed = torch.empty(nbytes // 2, dtype=torch.float16, device=w2.device)
    _lib.check(_lib.lib().pe_bneck64_pack(_lib.ptr(w2), _lib.ptr(w3), _lib.ptr(wsc), _lib.ptr(w1n), _lib.ptr(packed), _lib.stream()), "pe_bneck64_pack")
    return packed


def bneck64(t1, shortcut_src, packed, bias2, bias3, bias_sc=None, bias1n=None, out=None, t1_next=None):
    """out = relu(conv3(relu(conv2_3x3(t1))) + shortcut) of a 64-wide bottleneck block and, when bias1n is given (packed with
    w1n), t1_next = relu(conv1_next(out)), in one launch.  shortcut_src: x [N,H,W,256] (identity) or, when bias_sc is given
    (packed with wsc), the shortcut convolution's input [N,H,W,64].  Returns (out, t1_next | None)."""
    _lib.require_cuda(t1, shortcut_src, packed, bias2, bias3)
    N, H, W, C = t1.shape
    has_sc, has_next = bias_sc is not None, bias1n is not None
    assert C == 64 and t1.is_contiguous() and shortcut_src.is_contiguous() and t1.dtype == torch.float16
    assert tuple(shortcut_src.shape) == (N, H, W, 64 if has_sc else 256)
    assert packed.numel() * 2 == _lib.lib().pe_bneck64_packed_bytes(int(has_sc), int(has_next)), "packed stream does not match the requested stages"
    if out is None:
        out = torch.empty((N, H, W, 256), dtype=torch.float16, device=t1.device)
    if has_next and t1_next is None:
        t1_next = torch.empty((N, H, W, 64), dtype=torch.float16, device=t1.device)
    st = _lib.lib().pe_bneck64_f16(_lib.ptr(t1), _lib.ptr(shortcut_src), _lib.ptr(packed), _lib.ptr(bias2), _lib.ptr(bias3), _lib.ptr(bias_sc),
                                   _lib.ptr(bias1n), _lib.ptr(out), _lib.ptr(t1_next if has_next else None), N, H, W, int(has_sc), int(has_next),
                                   _lib.stream())
    _lib.check(st, "pe_bneck64_f16")
    if PROFILE is not None:
        M = N * H * W
        PROFILE.append({"variant": f"bneck64_kernel<{int(has_sc)}, {int(has_next)}>", "shape": f"N{N} {H}x{W} 64->64->256{'+sc' if has_sc else '+id'}{'->64' if has_next else ''} f320",
                        "flops": 2.0 * M * 64 * (576 + 256 + (64 if has_sc else 0) + (256 if has_next else 0)),
                        "bytes": float(M * 2 * (64 + (64 if has_sc else 256) + 256 + (64 if has_next else 0)) + packed.numel() * 2),
                        "replay": (lambda: bneck64(t1, shortcut_src, packed, bias2, bias3, bias_sc, bias1n, out=out, t1_next=t1_next))})
    return out, (t1_next if has_next else None)


def conv_wd_pack_head(weight2d):
    """[rows <= 16, 256] fp16 head weight (objectness + anchor deltas) -> the fragment-ordered 16 KiB block of the fused RPN head."""
    _lib.require_cuda(weight2d)
    rows, C = weight2d.shape
    assert weight2d.dtype == torch.float16 and weight2d.is_contiguous()
    packed = torch.empty(4 * 4 * 64 * 8, dtype=torch.float16, device=weight2d.device)
    _lib.check(_lib.lib().pe_conv_wd_pack_head(_lib.ptr(weight2d), _lib.ptr(packed), rows, C, _lib.stream()), "pe_conv_wd_pack_head")
    return packed


def conv3x3_wd_rpn_head(x, packed, bias, packed_head, head_bias16, out=None):
    """StandardRPNHead for one level in one launch: relu(conv3x3(x)) (256 channels, never stored) -> 1x1 head ->
    fp32 [N,H,W,16] (3 objectness logits, 12 deltas, 1 pad)."""
    _lib.require_cuda(x, packed, bias, packed_head, head_bias16)
    N, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((N, H, W, 16), dtype=torch.float32, device=x.device)
    st = _lib.lib().pe_conv3x3_wd_rpn_head_f16(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(packed_head), _lib.ptr(head_bias16),
                                               _lib.ptr(out), N, H, W, Cin, _lib.stream())
    _lib.check(st, "pe_conv3x3_wd_rpn_head_f16")
    if PROFILE is not None:
        M = N * H * W
        wd9 = _lib.test_hooks().pe_test_wd9_head_takes(N, H, W)
        PROFILE.append({"variant": "conv3x3_wd9_kernel<8, 4, 9, 1>" if wd9 else "conv3x3_wd_kernel<1, 4, 4, 4, 0, 1>",
                        "shape": f"N{N} {H}x{W} Cin{Cin} Cout256+head k3 s1 res0 f321",
                        "flops": 2.0 * M * 256 * 9 * Cin + 2.0 * M * 15 * 256, "bytes": float(M * Cin * 2 + 256 * 9 * Cin * 2 + M * 15 * 4),
                        "replay": (lambda: conv3x3_wd_rpn_head(x, packed, bias, packed_head, head_bias16, out=out))})
    return out


DEFAULT_CONV_POLICY = 329   # tile_bits of csrc/test_hooks.h pe_test_set_conv_policy the library starts with (tests restore it)


def conv_variant_name(M, Cout, kernel, K=0, *, stride=1, residual_mode=0, out_f32=False, has_bias=True, cout_store=0, out_stride=0, in_pixels=None):
    """Name of the kernel pe_conv2d_nhwc_f16 dispatches to under the DEFAULT policy (mirrors conv2_dispatch in
    csrc/conv_igemm2.hip; matches the rocprofv3 kernel names)."""
    bn = 64 if Cout <= 64 else 128
    if kernel == 7:
        return "conv_igemm_kernel<128, 64, 2>"          # 7x7 stem, register-staged kernel
    if kernel == 1:
        ring = (stride in (1, 2) and not out_f32 and has_bias and Cout % 256 == 0
                and (K >= (512 if stride == 1 else 256) if residual_mode == 0 else (stride == 1 and K >= 128 and M * Cout * 2 < 2 ** 31))
                and (cout_store or Cout) == Cout and (in_pixels or M) * K * 2 < 2 ** 31 and Cout * K * 2 < 2 ** 31 and M * (out_stride or Cout) * 2 < 2 ** 31)
        if ring:
            return "conv1x1_ring_kernel"                # persistent loader / consumer kernel (csrc/conv1x1_ring.hip)
        if K >= 4096 and Cout % 256 == 0 and ((M + 255) // 256) * (Cout // 256) >= 224:
            return "conv_big_kernel<0>"                 # 256x256 two-stage kernel (long-K GEMM)
        return f"conv_igemm2_kernel<128, {bn}, 0, 1>"   # LDS-DMA 1x1 / GEMM
    if bn == 64:
        return "conv3x3rb_kernel<128, 64>"
    big = ((M + 255) // 256) * ((Cout + 127) // 128) >= 512
    return f"conv3x3rb_kernel<{256 if big else 128}, 128>"  # kw-reuse 3x3, double-buffered weight tile


def linear_f16(x, weight, bias, *, relu=False, out_f32=False, cout_store=0, out_stride=0):
    """x [M,K] fp16, weight [Cout,K] fp16 -> [M,Cout]: the conv kernel with H = W = 1."""
    M, K = x.shape
    y = conv2d_nhwc(x.view(M, 1, 1, K), weight.view(weight.shape[0], 1, 1, K), bias, kernel=1, relu=relu,
                    out_f32=out_f32, cout_store=cout_store, out_stride=out_stride)
    return y.view(M, -1)


def maxpool3x3s2_nhwc(x):
    N, H, W, C = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().pe_maxpool3x3s2_nhwc(_lib.ptr(x), _lib.ptr(out), N, H, W, C, _lib.stream()), "pe_maxpool3x3s2_nhwc")
    return out


def stem_conv_pool(x, weight_shifted, bias):
    """Fused BasicStem: conv7x7/2 (+folded BN) + ReLU + max_pool2d(3, 2, 1)  (backbone/resnet.py:375-384).
    x [N,H,W,4] fp16 (H, W multiples of 4); weight_shifted [64,7,8,4] fp16 with tap 0 zero (weights.pack_stem_fused)."""
    N, H, W, C = x.shape
    assert C == 4 and tuple(weight_shifted.shape) == (64, 7, 8, 4) and x.dtype == torch.float16
    out = torch.empty((N, H // 4, W // 4, 64), dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().pe_stem_conv7x7_maxpool_f16(_lib.ptr(x), _lib.ptr(weight_shifted), _lib.ptr(bias), _lib.ptr(out),
                                                      N, H, W, _lib.stream()), "pe_stem_conv7x7_maxpool_f16")
    return out


def subsample2_nhwc(x):
    N, H, W, C = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().pe_subsample2_nhwc(_lib.ptr(x), _lib.ptr(out), N, H, W, C, _lib.stream()), "pe_subsample2_nhwc")
    return out


def preprocess_pack(src, dst, *, src_kind, ch0, nch, flip_rgb, dst_hw, mean, std):
    """src: one image (HWC u8 / HWC f32 / CHW f32 device tensor); dst: [pad_h,pad_w,4] fp16 view."""
    _lib.require_cuda(src, dst)
    if src_kind == 2:
        c, h, w = src.shape
    else:
        h, w, c = src.shape
    m = (ctypes.c_float * 4)(*(list(mean) + [0.0] * (4 - len(mean))))
    s = (ctypes.c_float * 4)(*(list(std) + [1.0] * (4 - len(std))))
    st = _lib.lib().pe_preprocess_pack(_lib.ptr(src), src_kind, h, w, c, ch0, nch, int(flip_rgb), dst_hw[0], dst_hw[1],
                                       dst.shape[0], dst.shape[1], m, s, _lib.ptr(dst), _lib.stream())
    _lib.check(st, "pe_preprocess_pack")


_PIL_TABLES = {}


def _pil_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    if key not in _PIL_TABLES:
        from .data import pil_bilinear_tables
        t = torch.from_numpy(pil_bilinear_tables(in_size, out_size)).to(device)
        torch.cuda.current_stream(t.device).synchronize()   # the cache is shared by every stream: publish a COMPLETE table
        _PIL_TABLES[key] = t
    return _PIL_TABLES[key]


def preprocess_pack_pil_u8(src, dst, *, ch0, nch, flip_rgb, dst_hw, mean, std):
    """src: [N,H,W,C] uint8 batch (or one [H,W,C] image with dst [pad_h,pad_w,4]); Pillow-exact bilinear resize to
    dst_hw, then normalise / pad / NHWC4 fp16 pack (the reference's 3-channel path, transform.py:92-97)."""
    _lib.require_cuda(src, dst)
    if src.dim() == 3:
        src, dst = src.unsqueeze(0), dst.unsqueeze(0)
    assert src.dtype == torch.uint8 and src.is_contiguous()
    n, h, w, c = src.shape
    xt, yt = _pil_tables(w, dst_hw[1], src.device), _pil_tables(h, dst_hw[0], src.device)
    m = (ctypes.c_float * 4)(*(list(mean) + [0.0] * (4 - len(mean))))
    s = (ctypes.c_float * 4)(*(list(std) + [1.0] * (4 - len(std))))
    st = _lib.lib().pe_preprocess_pack_pil_u8(_lib.ptr(src), n, h, w, c, ch0, nch, int(flip_rgb), dst_hw[0], dst_hw[1],
                                              dst.shape[1], dst.shape[2], m, s, _lib.ptr(xt), xt.shape[1] - 2,
                                              _lib.ptr(yt), yt.shape[1] - 2, _lib.ptr(dst), _lib.stream())
    _lib.check(st, "pe_preprocess_pack_pil_u8")


def preprocess_pack_batch(src, dst, *, src_kind, ch0, nch, flip_rgb, dst_hw, mean, std):
    """src: [N,H,W,C] (u8 / f32) or [N,C,H,W] f32 batch of equally sized images; dst: [N,pad_h,pad_w,4] fp16."""
    _lib.require_cuda(src, dst)
    n = src.shape[0]
    if src_kind == 2:
        c, h, w = src.shape[1:]
    else:
        h, w, c = src.shape[1:]
    m = (ctypes.c_float * 4)(*(list(mean) + [0.0] * (4 - len(mean))))
    s = (ctypes.c_float * 4)(*(list(std) + [1.0] * (4 - len(std))))
    st = _lib.lib().pe_preprocess_pack_batch(_lib.ptr(src), n, src_kind, h, w, c, ch0, nch, int(flip_rgb), dst_hw[0], dst_hw[1],
                                             dst.shape[1], dst.shape[2], m, s, _lib.ptr(dst), _lib.stream())
    _lib.check(st, "pe_preprocess_pack_batch")
