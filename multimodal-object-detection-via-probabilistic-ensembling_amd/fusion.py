"""ProbEn late fusion on the GPU - host side.

Mirrors the reference's call surface (demo/FLIR/demo_probEn.py):
  fusion(method, info_1, info_2, info_3='')   :189-196  (one image, Python lists in)
and adds the batched form the MI355X path actually uses:
  fuse_batch(...)                             one launch, one 1024-thread workgroup per image.
The arithmetic lives in csrc/proben.hip behind pe_proben_fuse_batch.
"""
import numpy as np
import torch

from . import _lib

SCORE_MODES = {"probEn": 0, "avg": 1, "max": 2, "probEn_binary": 3}
BOX_MODES = {"v-avg": 0, "s-avg": 1, "avg": 2, "argmax": 3}
FRAME_W, FRAME_H = 640.0, 512.0  # class-band shift hard-coded by the reference (demo_probEn.py:100-103)


def fuse_batch(boxes, scores, probs, variances, classes, offsets, score_fusion="probEn", box_fusion="v-avg",
               max_rows=None, iou_thresh=0.5, frame=(FRAME_W, FRAME_H), row_counts=None, passthrough=None):
    """Fuse B images in one launch.

    boxes f64 [Ntot,4], scores f64 [Ntot], probs f64 [Ntot,K], variances f64 [Ntot],
    classes i32 [Ntot], offsets i32 [B+1] - all CUDA tensors, rows of each image already
    concatenated in detector order.  Returns a dict of device tensors:
      boxes f64 [Ntot,4], scores f32 [Ntot], classes f32 [Ntot], keep i32 [Ntot], counts i32 [B];
    image b's fused rows are [offsets[b], offsets[b]+counts[b]).
    """
    _lib.require_cuda(boxes, scores, probs, variances, classes, offsets)
    if score_fusion == "max" and box_fusion == "argmax":
        raise ValueError("('max','argmax') is the class-aware NMS route: use fusion()/nms_fuse_batch")
    B = offsets.numel() - (0 if row_counts is not None else 1)
    ntot = boxes.shape[0]
    K = probs.shape[1] if probs is not None and probs.dim() == 2 else 1
    boxes = boxes.contiguous().double()
    scores = scores.contiguous().double()
    probs = probs.contiguous().double() if probs is not None else None
    variances = variances.reshape(-1).contiguous().double()
    classes = classes.contiguous().to(torch.int32)
    offsets = offsets.contiguous().to(torch.int32)
    if max_rows is None:
        assert row_counts is None, "max_rows must be given with row_counts (avoids a host sync)"
        max_rows = int((offsets[1:] - offsets[:-1]).max().item()) if B > 0 else 1
    max_rows = max(int(max_rows), 1)
    dev = boxes.device
    out = {
        "boxes": torch.empty((ntot, 4), dtype=torch.float64, device=dev),
        "scores": torch.empty((ntot,), dtype=torch.float32, device=dev),
        "classes": torch.empty((ntot,), dtype=torch.float32, device=dev),
        "keep": torch.empty((ntot,), dtype=torch.int32, device=dev),
        "counts": torch.zeros((max(B, 1),), dtype=torch.int32, device=dev)[:B],
    }
    st = _lib.lib().pe_proben_fuse_batch(
        _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(probs), _lib.ptr(variances), _lib.ptr(classes),
        _lib.ptr(offsets), _lib.ptr(row_counts), _lib.ptr(passthrough), B, K, max_rows, SCORE_MODES[score_fusion], BOX_MODES[box_fusion],
        float(iou_thresh), float(frame[0]), float(frame[1]),
        _lib.ptr(out["boxes"]), _lib.ptr(out["scores"]), _lib.ptr(out["classes"]), _lib.ptr(out["keep"]),
        _lib.ptr(out["counts"]), _lib.stream())
    _lib.check(st, "pe_proben_fuse_batch")
    return out


def pack_infos(per_image_infos, device="cuda"):
    """per_image_infos: list (images) of lists (detectors) of reference-style dicts
    {bbox, score, class, prob, vars}.  Returns the flat device tensors + offsets."""
    bb, ss, cc, pp, vv, offs = [], [], [], [], [], [0]
    K = None
    for infos in per_image_infos:
        for d in infos:
            if d and len(d["prob"]) > 0:
                K = len(d["prob"][0])
                break
        if K:
            break
    K = K or 3
    for infos in per_image_infos:
        n = 0
        for d in infos:
            if not d or len(d["bbox"]) == 0:
                continue
            bb.append(np.asarray(d["bbox"], dtype=np.float64).reshape(-1, 4))
            ss.append(np.asarray(d["score"], dtype=np.float64).reshape(-1))
            cc.append(np.asarray(d["class"], dtype=np.int32).reshape(-1))
            pp.append(np.asarray(d["prob"], dtype=np.float64).reshape(-1, K))
            vv.append(np.asarray(d["vars"], dtype=np.float64).reshape(-1))
            n += len(ss[-1])
        offs.append(offs[-1] + n)

    def cat(xs, shape, dt):
        return torch.from_numpy(np.concatenate(xs) if xs else np.zeros(shape, dtype=dt)).to(device)

    return (cat(bb, (0, 4), np.float64), cat(ss, (0,), np.float64), cat(pp, (0, K), np.float64),
            cat(vv, (0,), np.float64), cat(cc, (0,), np.int32),
            torch.tensor(offs, dtype=torch.int32, device=device))


def fusion(method, info_1, info_2, info_3=""):
    """Drop-in for the reference's ``fusion`` (demo_probEn.py:189-196).

    Returns (out_boxes, out_scores, out_class): boxes as a list of float64 ndarrays [4]
    (or a float32 Tensor [n,4] on the ('max','argmax') route), scores / classes as float32
    CPU tensors - the reference's return types."""
    infos = [info_1, info_2] + ([info_3] if info_3 else [])
    if method[0] == "max" and method[1] == "argmax":
        from .layers import batched_nms
        boxes = torch.tensor(sum([list(d["bbox"]) for d in infos], []), dtype=torch.float32).reshape(-1, 4)
        scores = torch.tensor(sum([list(d["score"]) for d in infos], []), dtype=torch.float32)
        classes = torch.tensor(sum([list(d["class"]) for d in infos], []), dtype=torch.float32)
        keep = batched_nms(boxes.cuda(), scores.cuda(), classes.cuda(), 0.5).cpu()
        return boxes[keep], scores[keep], classes[keep]
    b, s, p, v, c, offs = pack_infos([infos])
    out = fuse_batch(b, s, p, v, c, offs, method[0], method[1])
    m = int(out["counts"][0].item())
    if m < 0:
        raise RuntimeError("fusion: too many rows for one image")
    boxes = out["boxes"][:m].cpu().numpy()
    return [boxes[i] for i in range(m)], out["scores"][:m].cpu(), out["classes"][:m].cpu()


def fuse_detections(dets, score_fusion="probEn", box_fusion="v-avg", max_class=2, iou_thresh=0.5):
    """Device-to-device stage fusion: `dets` = the result dicts of 2 or 3 detectors run on the SAME batch
    (rcnn.GeneralizedRCNN.forward_batch).  Packs their detections into ProbEn rows (classes <= max_class,
    like the JSON writer demo_FLIR_save_predictions.py:148-155), applies the reference's per-image case
    split (0 detectors -> nothing, 1 -> passthrough, >= 2 -> fusion; demo_probEn.py:237-267) and fuses.
    No host synchronisation.  Returns a dict: boxes f64 [B*S,4], scores f32, classes f32, counts i32 [B],
    offsets i32 [B], stride S = len(dets) * D."""
    import ctypes
    nd = len(dets)
    B, D = dets[0]["scores"].shape
    # the box heads' candidate-cap bookkeeping travels with the result (no kernel here): check_candidate_overflow() looks
    # at it at the consumer's first host synchronisation
    overflow_src = [(d["cand_total"], d["cand_max"]) for d in dets if "cand_total" in d]
    K = dets[0]["prob_score"].shape[2]
    dev = dets[0]["scores"].device
    S = nd * D

    def arr(key):
        return (ctypes.c_void_p * nd)(*[d[key].data_ptr() for d in dets])
    ob = torch.empty((B * S, 4), dtype=torch.float64, device=dev)
    os_ = torch.empty((B * S,), dtype=torch.float64, device=dev)
    op = torch.empty((B * S, K), dtype=torch.float64, device=dev)
    ov = torch.empty((B * S,), dtype=torch.float64, device=dev)
    oc = torch.empty((B * S,), dtype=torch.int32, device=dev)
    ooff = torch.empty((B,), dtype=torch.int32, device=dev)
    ocnt = torch.empty((B,), dtype=torch.int32, device=dev)
    osingle = torch.empty((B,), dtype=torch.int32, device=dev)
    st = _lib.lib().pe_proben_pack_detections(arr("boxes"), arr("scores"), arr("classes"), arr("prob_score"), arr("vars"),
                                             arr("counts"), nd, B, D, K, max_class, S, _lib.ptr(ob), _lib.ptr(os_),
                                             _lib.ptr(op), _lib.ptr(ov), _lib.ptr(oc), _lib.ptr(ooff), _lib.ptr(ocnt),
                                             _lib.ptr(osingle), _lib.stream())
    _lib.check(st, "pe_proben_pack_detections")
    if score_fusion == "max" and box_fusion == "argmax":
        from .layers import nms_batched_raw
        b32 = ob.float().view(B, S, 4)
        keep, kcnt = nms_batched_raw(b32, os_.float().view(B, S), oc.view(B, S), ocnt, None, iou_thresh, 0, S)
        # images where only ONE detector fired are passed through untouched by the reference (demo_probEn.py:239-254)
        single = osingle.bool()
        keep = torch.where(single[:, None], torch.arange(S, dtype=torch.int32, device=dev).expand(B, S), keep)
        kcnt = torch.where(single, ocnt, kcnt)
        # same contract as the ProbEn route: image b's fused rows are [offsets[b], offsets[b] + counts[b]) of the flat
        # arrays (score-descending), so late_fusion.fused_rows_device / comm.all_gather_fused_rows need no special case
        g = (torch.arange(B, device=dev).view(B, 1) * S + keep.clamp(0, S - 1).long()).view(-1)
        return {"boxes": ob[g], "scores": os_.float()[g], "classes": oc.float()[g], "counts": kcnt, "keep": keep,
                "offsets": ooff, "stride": S, "in_counts": ocnt, "nms_route": True, "cand_overflow_src": overflow_src}
    out = fuse_batch(ob, os_, op, ov, oc, ooff, score_fusion, box_fusion, max_rows=S, iou_thresh=iou_thresh,
                     row_counts=ocnt, passthrough=osingle)
    out["offsets"], out["stride"], out["in_counts"] = ooff, S, ocnt
    out["cand_overflow_src"] = overflow_src
    return out


def check_candidate_overflow(result):
    """Raise if a box head met more (proposal, class) candidates above SCORE_THRESH_TEST than its NMS stage holds
    (rcnn._roi_heads: cand_total > cand_max) - the device-to-device routes (forward_batch -> FramePairPipeline ->
    fuse_detections) would otherwise lose those detections silently, in proposal order, where the reference keeps all.
    `result`: a forward_batch dict or a fuse_detections dict.  Synchronises with the DEVICE (not just the current stream: the
    counters are written on the detectors' side streams of FramePairPipeline, which the caller's stream need not have waited for):
    call it where the consumer reads results anyway (GeneralizedRCNN.to_instances does the same check for the Instances route)."""
    src = result.get("cand_overflow_src")
    if src is None and "cand_total" in result:
        src = [(result["cand_total"], result["cand_max"])]
    if src and any(t.is_cuda for t, _ in src):
        torch.cuda.synchronize(src[0][0].device)
    for tot, cmax in src or []:
        worst = int(tot.max())
        if worst > cmax:
            raise RuntimeError(f"box head: {worst} (proposal, class) candidates pass the score threshold on one image but the "
                               f"NMS stage holds {cmax}: detections would be dropped in proposal order (the reference keeps all). "
                               "Raise SCORE_THRESH_TEST or lower POST_NMS_TOPK_TEST.")
