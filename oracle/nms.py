"""Oracle: greedy NMS / batched NMS in float32 (test infrastructure only).

The reference calls torchvision 0.13.0 (pinned in probEn.yml:172; NOT vendored
under the reference tree) at detectron2/layers/nms.py:16-37.  Consumers:
modeling/proposal_generator/rpn_outputs.py:147, modeling/roi_heads/fast_rcnn.py:130,
demo/FLIR/demo_probEn.py:64.  This restates torchvision's published algorithm:

  nms:         sort by score descending; walk in that order; a box survives if no
               earlier survivor has IoU > thr with it;
               IoU = inter / (area_a + area_b - inter), area = (x2-x1)*(y2-y1),
               inter = max(0, xx2-xx1) * max(0, yy2-yy1), all float32, no "+1".
  batched_nms: "coordinate trick" (boxes + idx * (max_coord + 1)) when
               boxes.numel() <= 4000 on CPU / 20000 on GPU, else one nms per class
               ("vanilla"), result re-sorted by score descending.

PARITY: INDIRECTLY PINNED (round 4).  torchvision itself is absent from the reference tree and from the build container, but the
reference holds two statements of horizontal greedy NMS that its own tests equate with torchvision's: the Python
`reference_horizontal_nms` (tests/test_nms_rotated.py:11-33, asserted == nms_rotated at 0 degrees, :89-101) and the C++ rotated
kernel at 0 degrees (layers/csrc/nms_rotated/nms_rotated_cpu.cpp:7-60 + box_iou_rotated_utils.h:315-340, asserted ==
torchvision batched_nms for IoU 0.2 / 0.5 / 0.8, :45-66).  tests/golden/gen_nms.py EXECUTES both on this repo's fixtures (a 4 624-box
RPN-like set, a dense 1 500-box set, the reference test's own recipe; IoU 0.5 / 0.7) and tests/test_oracle_nms.py requires this
file to reproduce their keep lists index for index (the HIP kernel likewise, tests/test_ops_gpu.py).  Not covered by reference-held
code: the coordinate-trick / per-class dispatch thresholds of batched_nms (torchvision's, restated) and the order of exactly tied
scores (the fixtures are tie-free).
Tie rule: score descending, then index ascending (stable descending sort).
"""
import numpy as np


def order_desc_stable(scores):
    scores = np.asarray(scores)
    # stable descending == stable ascending on negated keys (NaN-free inputs)
    return np.argsort(-scores.astype(np.float64), kind="stable")


def nms_f32(boxes, scores, thr):
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    n = len(scores)
    if n == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = order_desc_stable(scores)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr32 = np.float32(thr)
    zero = np.float32(0)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        rest = rest[~suppressed[rest]]
        if len(rest) == 0:
            continue
        w = np.maximum(zero, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]))
        h = np.maximum(zero, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr32]] = True
    return np.asarray(keep, dtype=np.int64)


def batched_nms_f32(boxes, scores, idxs, thr, device_type="cuda", mode=None):
    """mode: None -> torchvision's own dispatch rule for `device_type`;
    'trick' / 'vanilla' to force one."""
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    idxs = np.asarray(idxs)
    if boxes.size == 0:
        return np.zeros((0,), dtype=np.int64)
    if mode is None:
        limit = 4000 if device_type == "cpu" else 20000
        mode = "vanilla" if boxes.size > limit else "trick"
    if mode == "trick":
        max_coord = boxes.max()
        offsets = idxs.astype(np.float32) * (max_coord + np.float32(1))
        return nms_f32(boxes + offsets[:, None], scores, thr)
    keep_mask = np.zeros(len(scores), dtype=bool)
    for c in np.unique(idxs):
        sel = np.nonzero(idxs == c)[0]
        keep_mask[sel[nms_f32(boxes[sel], scores[sel], thr)]] = True
    kept = np.nonzero(keep_mask)[0]
    return kept[order_desc_stable(scores[kept])]
