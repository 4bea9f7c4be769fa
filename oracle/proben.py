"""Oracle: ProbEn late fusion (NumPy float64), test infrastructure only.

Restates the arithmetic of the reference's per-image fusion:
  demo/FLIR/demo_probEn.py:79-90   prepare_data        (concatenate detector lists)
  demo/FLIR/demo_probEn.py:92-187  nms_bayesian        (greedy clustering + fusion)
  demo/FLIR/demo_probEn.py:32-42   bayesian_fusion_multiclass
  demo/FLIR/demo_probEn.py:24-30   bayesian_fusion     (binary form, K = 1)
  demo/FLIR/demo_probEn.py:73-77   weighted_box_fusion
  demo/FLIR/demo_probEn.py:20-22   avg_bbox_fusion
  demo/FLIR/demo_probEn.py:44-71   nms_1               (the ('max','argmax') route)
  demo/FLIR/demo_probEn.py:189-196 fusion              (dispatch)
  demo/FLIR/demo_probEn.py:198-298 apply_late_fusion_and_evaluate (per-image driver: which lists are fused, what the evaluator gets)

Pinned by tests/golden/proben_*.npz, generated in the build container by
running the reference functions themselves (tests/golden/gen_proben.py); the
driver by tests/golden/p5_cases.json (gen_p5.py: the reference's function run
on three prediction dicts with a recording evaluator).

Tie rule (reference: ``scores.argsort()[::-1]``, an unstable sort whose tie
order depends on the NumPy build): score descending, then ORIGINAL INDEX
DESCENDING - what the reference's pinned NumPy 1.23 yields for short arrays
(insertion sort is stable, then reversed).  The HIP kernel uses the same rule.
"""
import numpy as np

SCORE_MODES = {"probEn": 0, "avg": 1, "max": 2, "probEn_binary": 3}
BOX_MODES = {"v-avg": 0, "s-avg": 1, "avg": 2, "argmax": 3}


def order_desc(scores):
    """Indices sorted by score descending, ties by original index descending."""
    scores = np.asarray(scores, dtype=np.float64)
    return np.argsort(scores, kind="stable")[::-1]


def _seq_sum(values):
    """Left-to-right float64 sum (cluster order).  The reference's np.sum is
    sequential for axis-0 reductions and for < 8 elements; the kernel sums in
    this order too."""
    acc = np.zeros_like(np.asarray(values[0], dtype=np.float64))
    for v in values:
        acc = acc + v
    return acc


def fuse_score(score_mode, probs, scores, pivot_class):
    """probs [m,K], scores [m] in cluster order (matches first, pivot last).
    Returns (score, class) as float64 / float."""
    m, K = probs.shape
    if score_mode == "probEn":
        # demo_probEn.py:32-42 - background column = 1 - sum(p); product of
        # per-detector posteriors in log space; normalise; max INCLUDING background.
        full = np.zeros((m, K + 1))
        full[:, :K] = probs
        full[:, K] = 1.0 - _seq_sum(list(probs.T))
        with np.errstate(divide="ignore", invalid="ignore"):
            logs = np.log(full)
        s = np.exp(_seq_sum(list(logs)))
        with np.errstate(divide="ignore", invalid="ignore"):
            s = s / _seq_sum(list(s))
        return float(np.max(s)), float(np.argmax(s))
    if score_mode == "probEn_binary":
        # demo_probEn.py:24-30 (unused in the reference's FLIR script; K = 1 form)
        with np.errstate(divide="ignore", invalid="ignore"):
            pos = np.exp(_seq_sum(list(np.log(scores))))
            neg = np.exp(_seq_sum(list(np.log(1.0 - scores))))
            return float(pos / (pos + neg)), float(pivot_class)
    if score_mode == "avg":
        return float(_seq_sum(list(scores)) / m), float(pivot_class)
    if score_mode == "max":
        # max over the WHOLE prob matrix, not over `score` (demo_probEn.py:151)
        return float(np.max(probs)), float(pivot_class)
    raise ValueError(score_mode)


def fuse_box(box_mode, boxes, scores, variances):
    """boxes [m,4], scores [m], variances [m] in cluster order."""
    m = boxes.shape[0]
    if box_mode in ("v-avg", "s-avg"):
        with np.errstate(divide="ignore", invalid="ignore"):
            w = (1.0 / variances) if box_mode == "v-avg" else scores
            w = w / _seq_sum(list(w))
            return _seq_sum(list(boxes * w[:, None]))
    if box_mode == "avg":
        return _seq_sum(list(boxes)) / m
    if box_mode == "argmax":
        return boxes[int(np.argmax(scores))].copy()
    raise ValueError(box_mode)


def nms_bayesian(boxes, scores, classes, probs, variances, thresh, score_mode, box_mode,
                 frame_w=640.0, frame_h=512.0):
    """Greedy class-aware clustering + fusion of one image's concatenated rows.

    boxes [N,4] xyxy, scores [N], classes [N], probs [N,K], variances [N] -> float64.
    Returns (keep [M] int, out_scores [M] f64, out_boxes [M,4] f64, out_classes [M] f64).
    """
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float64).reshape(-1)
    classes = np.asarray(classes, dtype=np.float64).reshape(-1)
    probs = np.asarray(probs, dtype=np.float64).reshape(len(scores), -1)
    variances = np.asarray(variances, dtype=np.float64).reshape(-1)
    n = len(scores)
    # class separation by shifting into disjoint frames, legacy "+1" areas
    x1 = boxes[:, 0] + classes * frame_w
    y1 = boxes[:, 1] + classes * frame_h
    x2 = boxes[:, 2] + classes * frame_w
    y2 = boxes[:, 3] + classes * frame_h
    areas = (x2 - x1 + 1.0) * (y2 - y1 + 1.0)
    order = order_desc(scores)
    alive = np.ones(n, dtype=bool)
    keep, out_s, out_b, out_c = [], [], [], []
    for pos in range(n):
        i = order[pos]
        if not alive[i]:
            continue
        alive[i] = False
        rest = order[pos + 1:]
        rest = rest[alive[rest]]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1.0)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1.0)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        matched = rest[ovr > thresh]
        # the reference keeps only rows with ovr <= thresh: NaN rows vanish too
        alive[rest[~(ovr <= thresh)]] = False
        keep.append(int(i))
        if len(matched) > 0:
            members = np.concatenate([matched, [i]])  # pivot LAST
            s, c = fuse_score(score_mode, probs[members], scores[members], classes[i])
            b = fuse_box(box_mode, boxes[members], scores[members], variances[members])
        else:
            s, c, b = float(scores[i]), float(classes[i]), boxes[i].copy()
        out_s.append(s)
        out_c.append(c)
        out_b.append(b)
    return (np.asarray(keep, dtype=np.int64), np.asarray(out_s, dtype=np.float64),
            np.asarray(out_b, dtype=np.float64).reshape(-1, 4), np.asarray(out_c, dtype=np.float64))


def concat_infos(infos):
    """prepare_data (demo_probEn.py:79-90): rows of detector 1, then 2, then 3."""
    infos = [d for d in infos if d]
    K = None
    for d in infos:
        if len(d["prob"]) > 0:
            K = len(d["prob"][0])
    K = K or 3
    boxes = np.concatenate([np.asarray(d["bbox"], dtype=np.float64).reshape(-1, 4) for d in infos])
    scores = np.concatenate([np.asarray(d["score"], dtype=np.float64).reshape(-1) for d in infos])
    classes = np.concatenate([np.asarray(d["class"], dtype=np.float64).reshape(-1) for d in infos])
    probs = np.concatenate([np.asarray(d["prob"], dtype=np.float64).reshape(-1, K) for d in infos])
    variances = np.concatenate([np.asarray(d["vars"], dtype=np.float64).reshape(-1) for d in infos])
    return boxes, scores, classes, probs, variances


def fusion(method, info_1, info_2, info_3=""):
    """Same call as the reference's ``fusion`` (demo_probEn.py:189-196).

    Returns (out_boxes, out_scores, out_class):
      ('max','argmax'): float32 arrays from class-aware NMS at 0.5 (nms_1 route);
      otherwise float64 boxes [M,4], float32 scores, float32 classes."""
    from .nms import batched_nms_f32
    boxes, scores, classes, probs, variances = concat_infos([info_1, info_2, info_3])
    if method[0] == "max" and method[1] == "argmax":
        b32 = boxes.astype(np.float32)
        s32 = scores.astype(np.float32)
        c32 = classes.astype(np.float32)
        keep = batched_nms_f32(b32, s32, c32, 0.5)
        return b32[keep], s32[keep], c32[keep]
    keep, s, b, c = nms_bayesian(boxes, scores, classes, probs, variances, 0.5, method[0], method[1])
    return b, s.astype(np.float32), c.astype(np.float32)


def late_fusion_rows(det_1, det_2, method, det_3="", img_folder="../../../Datasets/FLIR/val/thermal_8_bit/", image_hw=(512, 640)):
    """The per-image driver (demo_probEn.py:198-298) up to `evaluator.process`: one record per image that reaches the evaluator.
      * the loop runs over det_2's images (:205); a detector "fired" when its box list is non-empty (:224,236);
      * nobody fired -> the image is skipped (:239-240); one fired -> its list passes through unchanged, detector 1 first, then 2,
        then 3 (:242-254); two of three -> fusion of the two non-empty lists in detector order (:256-266); all -> fusion of all (:268-269);
      * file_name = img_folder + detector 1's image name up to its first '.' + '.jpeg' (:271); image_id is detector 2's (:283);
        height / width come from the image file (:272-273; FLIR thermal frames are 512 x 640);
      * boxes reach `Boxes` as float64 and are stored as float32 (structures/boxes.py:147-149), scores / classes as float32
        (`torch.Tensor(list)`, :246-247; the fused route's are float32 already)."""
    rows = []
    dets = [det_1, det_2] + ([det_3] if det_3 else [])
    for i in range(len(det_2["image"])):
        infos = [{"img_name": d["image"][i], "bbox": d["boxes"][i], "score": d["scores"][i], "class": d["classes"][i],
                  "prob": d["probs"][i], "vars": d["vars"][i]} for d in dets]
        live = [x for x in infos if len(x["bbox"]) > 0]
        if not live:
            continue
        if len(live) == 1:
            b = np.asarray(live[0]["bbox"], dtype=np.float64)
            sc = np.asarray(live[0]["score"], dtype=np.float32)
            c = np.asarray(live[0]["class"], dtype=np.float32)
        else:
            b, sc, c = fusion(method, *live)
        rows.append({"file_name": img_folder + det_1["image"][i].split(".")[0] + ".jpeg", "image_id": det_2["image_id"][i],
                     "height": image_hw[0], "width": image_hw[1], "boxes": np.asarray(b, dtype=np.float64).astype(np.float32).reshape(-1, 4),
                     "scores": np.asarray(sc, dtype=np.float32), "classes": np.asarray(c, dtype=np.float32)})
    return rows
