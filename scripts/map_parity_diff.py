"""Which decisions flip between the oracle (the reference's fp32 CPU arithmetic) and the HIP detector?

For every evaluation set of tests/parity_map.py the HIP detector runs twice: as shipped (SCORE_THRESH_TEST 0.5) and with the threshold at
0.05 (diagnostic: candidates below 0.5 never suppress anything above it in the score-ordered NMS, so the >= 0.5 part of the output is the
same and the rest shows what WOULD have been there).  Every oracle detection without a same-class IoU >= 0.9 partner on the device
("oracle-only") is put into exactly one class, first match wins:
  class_flip      the device has a detection at IoU >= 0.9 with another class
  threshold_flip  the device's low-threshold run has the same-class partner (IoU >= 0.9) with a score below 0.5
  nms_or_box      the device has a same-class detection at 0.5 <= IoU < 0.9: another member of the cluster won the final NMS, or the
                  same proposal regressed to a visibly different box
  missing         no counterpart even at threshold 0.05: the proposal did not reach the box head (RPN top-k / RPN NMS), or its score
                  moved by more than 0.45
The device-only detections get the mirror classes with what the committed oracle rows allow (the oracle was run at 0.5 only):
class_flip, near_threshold (device score < 0.5 + 3 sigma of the matched pairs' score difference), nms_or_box, unexplained.
Per class: counts, the score margin |s - 0.5| of its members, and the AP it carries (AP of a detector's rows minus AP of the same rows
without the class's detections).  Plus the pooled table: all sets as ONE dataset.
    python scripts/map_parity_diff.py [out.json]      (GPU box; ~1 min)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import proben_amd  # noqa: E402,F401
from parity_map import NAMES, coco_stats, evaluation_sets, hip_rows, match_signed  # noqa: E402


def iou_to(r, q):
    iw = (np.minimum(r[3], q[:, 3]) - np.maximum(r[1], q[:, 1])).clip(0)
    ih = (np.minimum(r[4], q[:, 4]) - np.maximum(r[2], q[:, 2])).clip(0)
    inter = iw * ih
    return inter / ((r[3] - r[1]) * (r[4] - r[2]) + (q[:, 3] - q[:, 1]) * (q[:, 4] - q[:, 2]) - inter + 1e-9)


def unmatched(a, b, n_frames, iou_min=0.9):
    """indices of rows of `a` without a same-class IoU >= iou_min partner in `b` (greedy, the matching of parity_map.match_signed)"""
    out = []
    for f in range(n_frames):
        ia, ib = np.nonzero(a[:, 0] == f)[0], np.nonzero(b[:, 0] == f)[0]
        used = np.zeros(len(ib), bool)
        for i in ia:
            cand = np.nonzero((~used) & (b[ib, 6] == a[i, 6]))[0]
            if len(cand):
                iou = iou_to(a[i], b[ib[cand]])
                j = int(iou.argmax())
                if iou[j] >= iou_min:
                    used[cand[j]] = True
                    continue
            out.append(i)
    return np.asarray(out, dtype=np.int64)


def classify(rows, idx, other, other_low, score_sigma):
    """rows[idx] are one detector's unmatched detections; `other` the other detector's rows at 0.5, `other_low` at 0.05 (or None)"""
    cls = {}
    for i in idx:
        r = rows[i]
        o = other[other[:, 0] == r[0]]
        iou = iou_to(r, o) if len(o) else np.zeros(0)
        same = o[:, 6] == r[6] if len(o) else np.zeros(0, bool)
        if len(o) and (iou[~same] >= 0.9).any():
            c = "class_flip"
        else:
            c = None
            if other_low is not None:
                ol = other_low[(other_low[:, 0] == r[0]) & (other_low[:, 6] == r[6])]
                if len(ol):
                    il = iou_to(r, ol)
                    if ((il >= 0.9) & (ol[:, 5] < 0.5)).any():
                        c = "threshold_flip"
            elif r[5] < 0.5 + 3 * score_sigma:
                c = "near_threshold"
            if c is None:
                c = "nms_or_box" if len(o) and ((iou >= 0.5) & (iou < 0.9) & same).any() else ("missing" if other_low is not None else "unexplained")
        cls[int(i)] = c
    return cls


def summarise(rows, cls, gts, base):
    out = {}
    for name in sorted(set(cls.values())):
        members = np.asarray([i for i, c in cls.items() if c == name], dtype=np.int64)
        keep = np.ones(len(rows), bool)
        keep[members] = False
        without = coco_stats(gts, rows[keep])
        out[name] = {"count": int(len(members)), "score_margin_median": float(np.median(np.abs(rows[members, 5] - 0.5))),
                     "score_margin_p90": float(np.percentile(np.abs(rows[members, 5] - 0.5), 90)),
                     "ap_carried_points": {n: float((base[k] - without[k]) * 100) for k, n in enumerate(NAMES[:3])}}
    return out


def main():
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_map_flips.json")
    sd, sets = evaluation_sets(os.path.join(ROOT, "tests", "golden"))
    model = GeneralizedRCNN(DetectorConfig(), sd)
    low = GeneralizedRCNN(DetectorConfig(score_thresh=0.05), sd)
    rec = {"sets": {}, "iou_match": 0.9, "low_threshold": 0.05}
    pool_gt, pool_o, pool_h, off = [], [], [], 0
    tot_o, tot_h = {}, {}
    for name, frames, gts, ora in sets:
        hip, hip_low = hip_rows(model, frames), hip_rows(low, frames)
        n = len(frames)
        ds, _, _, _ = match_signed(ora, hip, n)
        sigma = float(ds.std())
        only_o, only_h = unmatched(ora, hip, n), unmatched(hip, ora, n)
        co = classify(ora, only_o, hip, hip_low, sigma)
        ch = classify(hip, only_h, ora, None, sigma)
        so, sh = coco_stats(gts, ora), coco_stats(gts, hip)
        rec["sets"][name] = {"oracle_detections": int(len(ora)), "hip_detections": int(len(hip)), "matched_score_diff_sigma": sigma,
                             "delta_points": {nn: float((sh[k] - so[k]) * 100) for k, nn in enumerate(NAMES)},
                             "oracle_only": summarise(ora, co, gts, so), "hip_only": summarise(hip, ch, gts, sh)}
        for tot, cl in ((tot_o, co), (tot_h, ch)):
            for c in cl.values():
                tot[c] = tot.get(c, 0) + 1
        pool_gt += list(gts)
        for rows, pool in ((ora, pool_o), (hip, pool_h)):
            r = rows.copy()
            r[:, 0] += off
            pool.append(r)
        off += n
        print(name, {k: v["count"] for k, v in rec["sets"][name]["oracle_only"].items()}, {k: v["count"] for k, v in rec["sets"][name]["hip_only"].items()}, flush=True)
    po, ph = coco_stats(pool_gt, np.concatenate(pool_o)), coco_stats(pool_gt, np.concatenate(pool_h))
    rec["pooled"] = {"frames": off, "objects": int(sum(len(g[0]) for g in pool_gt)), "oracle": {n: float(po[k] * 100) for k, n in enumerate(NAMES)},
                     "hip": {n: float(ph[k] * 100) for k, n in enumerate(NAMES)}, "delta_points": {n: float((ph[k] - po[k]) * 100) for k, n in enumerate(NAMES)}}
    d = np.asarray([[s["delta_points"][n] for n in NAMES] for s in rec["sets"].values()])
    rec["per_set_delta"] = {"n_sets": int(len(d)), "mean": {n: float(d[:, k].mean()) for k, n in enumerate(NAMES)},
                            "std": {n: float(d[:, k].std(ddof=1)) for k, n in enumerate(NAMES)},
                            "standard_error": {n: float(d[:, k].std(ddof=1) / len(d) ** 0.5) for k, n in enumerate(NAMES)}}
    rec["totals"] = {"oracle_only": tot_o, "hip_only": tot_h}
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(rec, open(out_path, "w"), indent=1)
    print(json.dumps({k: rec[k] for k in ("totals", "pooled", "per_set_delta")}, indent=1))


if __name__ == "__main__":
    main()
