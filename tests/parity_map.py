"""Shared measurement code of the mAP-parity harness (tests/test_parity_map_gpu.py asserts on it, scripts/map_parity.py writes its
record to gpurun_out/ for profiles/): fixture loading, the COCO protocol, the HIP detector's rows on an evaluation set, per-detection
matching with SIGNED differences."""
import os

import numpy as np
import torch

NAMES = ["AP", "AP50", "AP75", "APs", "APm", "APl"]
NORTH_STAR_POINTS = 0.1          # "mAP within 1e-3" on the 0-100 scale


def coco_stats(gts, rows, hw=(512, 640)):
    from proben_amd import evaluation
    images = [{"id": i, "height": hw[0], "width": hw[1], "file_name": f"{i}.jpeg"} for i in range(len(gts))]
    cats = [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"}, {"id": 3, "name": "car"}]
    anns, aid = [], 1
    for i, (b, c) in enumerate(gts):
        for bb, cc in zip(b, c):
            w, h = float(bb[2] - bb[0]), float(bb[3] - bb[1])
            anns.append({"id": aid, "image_id": i, "category_id": int(cc) + 1, "bbox": [float(bb[0]), float(bb[1]), w, h], "area": w * h, "iscrowd": 0})
            aid += 1
    dets = [{"image_id": int(r[0]), "category_id": int(r[6]) + 1, "bbox": [float(r[1]), float(r[2]), float(r[3] - r[1]), float(r[4] - r[2])],
             "score": float(r[5])} for r in rows]
    ev = evaluation.COCOevalBBox({"images": images, "annotations": anns, "categories": cats}, dets, impl="native")
    ev.evaluate()
    ev.accumulate()
    return np.asarray(ev.summarize(printer=None), dtype=np.float64)


def load_fixture(golden_dir):
    import proben_amd  # noqa: F401
    from proben_amd.synthetic import labelled_frames, synthetic_state_dict
    z = np.load(os.path.join(golden_dir, "pseudo_heads_r101.npz"))
    sd = synthetic_state_dict(int(z["depth"]), 3, 3, seed=int(z["seed"]))
    for k in z.files:
        if "/" in k:
            sd[k.replace("/", ".")] = torch.from_numpy(z[k])
    frames, gts = labelled_frames(int(z["n_eval"]), seed=int(z["eval_seed"]))
    return z, sd, frames, gts


def evaluation_sets(golden_dir):
    """[(name, frames, ground truth, oracle rows)]: the fixture's own set + the extra sets of scripts/map_parity_sets.py (if generated)."""
    from proben_amd.synthetic import labelled_frames
    z, sd, frames, gts = load_fixture(golden_dir)
    sets = [(f"seed {int(z['eval_seed'])}", frames, gts, z["oracle_rows"])]
    extra = os.path.join(golden_dir, "pseudo_heads_r101_sets.npz")
    if os.path.exists(extra):
        e = np.load(extra)
        for k in sorted(e.files):
            if k.startswith("rows_"):
                fr, gt = labelled_frames(int(e["n_frames"]), seed=int(k[5:]))
                sets.append((f"seed {k[5:]}", fr, gt, e[k]))
    return sd, sets


def hip_rows(model, frames, batch=16):
    from proben_amd.data import resize_shortest_edge_shape
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    rows = []
    for b0 in range(0, len(frames), batch):
        fr = torch.from_numpy(frames[b0:b0 + batch]).cuda()
        det = model.forward_batch(fr, out_sizes=[(512, 640)] * len(fr), resize_to=new_hw)
        cnt = det["counts"].cpu().tolist()
        for i, c in enumerate(cnt):
            bx, sc, cl = det["boxes"][i, :c].cpu().numpy(), det["scores"][i, :c].cpu().numpy(), det["classes"][i, :c].cpu().numpy()
            rows += [[b0 + i, *bx[j], sc[j], cl[j]] for j in range(c)]
    return np.asarray(rows, dtype=np.float32).reshape(-1, 7)


def match_signed(ora, hip, n_frames, iou_min=0.9):
    """Greedy per-frame matching (same class, IoU >= iou_min).  Returns signed differences hip - oracle of the matched pairs:
    scores [m], boxes [m, 4] (x1, y1, x2, y2), and the unmatched counts."""
    ds, db, un_o, un_h = [], [], 0, 0
    for f in range(n_frames):
        o, h = ora[ora[:, 0] == f], hip[hip[:, 0] == f]
        used = np.zeros(len(h), bool)
        for r in o:
            cand = np.nonzero((~used) & (h[:, 6] == r[6]))[0]
            if len(cand) == 0:
                un_o += 1
                continue
            q = h[cand]
            iw = (np.minimum(r[3], q[:, 3]) - np.maximum(r[1], q[:, 1])).clip(0)
            ih = (np.minimum(r[4], q[:, 4]) - np.maximum(r[2], q[:, 2])).clip(0)
            inter = iw * ih
            iou = inter / ((r[3] - r[1]) * (r[4] - r[2]) + (q[:, 3] - q[:, 1]) * (q[:, 4] - q[:, 2]) - inter + 1e-9)
            j = int(iou.argmax())
            if iou[j] >= iou_min:
                used[cand[j]] = True
                ds.append(q[j, 5] - r[5])
                db.append(q[j, 1:5] - r[1:5])
            else:
                un_o += 1
        un_h += int((~used).sum())
    return np.asarray(ds, dtype=np.float64), np.asarray(db, dtype=np.float64).reshape(-1, 4), un_o, un_h


def measure(golden_dir):
    """The whole measurement: per evaluation set the oracle's and the HIP detector's AP table against the same ground truth, the
    deltas, their mean / std over the sets, and the signed per-coordinate box offset of the matched detections."""
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    sd, sets = evaluation_sets(golden_dir)
    model = GeneralizedRCNN(DetectorConfig(), sd)
    rec = {"model": "R101-FPN, seeded random backbone / FCs, fitted RPN + box predictor (tests/golden/gen_pseudo_heads.py)",
           "north_star_tolerance_points": NORTH_STAR_POINTS, "sets": {}}
    deltas, all_db, all_ds = [], [], []
    for name, frames, gts, ora_rows in sets:
        rows = hip_rows(model, frames)
        ora, hip = coco_stats(gts, ora_rows), coco_stats(gts, rows)
        ds, db, un_o, un_h = match_signed(ora_rows, rows, len(frames))
        deltas.append((hip - ora)[:6] * 100)
        all_db.append(db)
        all_ds.append(ds)
        rec["sets"][name] = {"frames": int(len(frames)), "ground_truth_objects": int(sum(len(g[0]) for g in gts)),
                             "oracle_detections": int(len(ora_rows)), "hip_detections": int(len(rows)),
                             "oracle": {n: float(ora[i] * 100) for i, n in enumerate(NAMES)}, "hip": {n: float(hip[i] * 100) for i, n in enumerate(NAMES)},
                             "delta": {n: float(deltas[-1][i]) for i, n in enumerate(NAMES)},
                             "matched_pairs": int(len(ds)), "oracle_only": un_o, "hip_only": un_h}
    d = np.asarray(deltas)
    db, ds = np.concatenate(all_db), np.concatenate(all_ds)
    rec["delta_mean"] = {n: float(d[:, i].mean()) for i, n in enumerate(NAMES)}
    rec["delta_std"] = {n: float(d[:, i].std(ddof=1)) if len(d) > 1 else None for i, n in enumerate(NAMES)}
    rec["n_sets"] = int(len(d))
    rec["matched_pairs_signed"] = {
        "pairs": int(len(ds)), "score_diff_mean": float(ds.mean()), "score_diff_std_of_mean": float(ds.std() / max(len(ds), 1) ** 0.5),
        "box_offset_mean_px": {c: float(db[:, i].mean()) for i, c in enumerate(["x1", "y1", "x2", "y2"])},
        "box_offset_std_of_mean_px": {c: float(db[:, i].std() / max(len(db), 1) ** 0.5) for i, c in enumerate(["x1", "y1", "x2", "y2"])},
        "box_abs_max_coord_median_px": float(np.median(np.abs(db).max(1))) if len(db) else None}
    return rec


# ------------------------------------------------------------------------------------------------------------------------
# FUSED mAP (BASELINE's metric: the AP of the rows ProbEn makes of two detectors' lists).  demo/FLIR/demo_probEn.py:198-298 ->
# detectron2/evaluation/FLIR_evaluation.py:249-310.  Fixtures: tests/golden/gen_fused_map.py (second pseudo-trained detector on the
# RGB rendering of the same scenes + the oracle's rows of BOTH detectors with class probabilities and variances).
# ------------------------------------------------------------------------------------------------------------------------
FUSED_METHODS = (("probEn", "v-avg"), ("avg", "s-avg"))


def load_fused_fixture(golden_dir, max_sets=None):
    """(state dicts (thermal, rgb), [(name, thermal frames, rgb frames, ground truth, oracle rows thermal, oracle rows rgb)]).
    `max_sets`: render only the first few sets' frames (the CPU consistency test needs one; rendering 24 sets takes a quarter of an hour on 8 cores)."""
    import proben_amd  # noqa: F401
    from proben_amd.synthetic import labelled_frames, labelled_frames_rgb, synthetic_state_dict
    sds = []
    for fn in ("pseudo_heads_r101.npz", "pseudo_heads_r101_rgb.npz"):
        z = np.load(os.path.join(golden_dir, fn))
        sd = synthetic_state_dict(int(z["depth"]), 3, 3, seed=int(z["seed"]))
        for k in z.files:
            if "/" in k:
                sd[k.replace("/", ".")] = torch.from_numpy(z[k])
        sds.append(sd)
    e = np.load(os.path.join(golden_dir, "fused_map_sets.npz"))
    n = int(e["n_frames"])
    sets = []
    for k in sorted(e.files):
        if k.startswith("t_") and "r_" + k[2:] in e.files:
            seed = int(k[2:])
            ft, gts = labelled_frames(n, seed=seed)
            fr, _ = labelled_frames_rgb(n, seed=seed)
            sets.append((f"seed {seed}", ft, fr, gts, e[k], e["r_" + k[2:]]))
            if max_sets is not None and len(sets) >= max_sets:
                break
    return sds, sets


def _info(rows, f):
    r = rows[rows[:, 0] == f]
    return {"img_name": str(f), "bbox": r[:, 1:5].astype(np.float64), "score": r[:, 5].astype(np.float64), "class": r[:, 6].astype(np.int64),
            "prob": r[:, 7:10].astype(np.float64), "vars": r[:, 10:11].astype(np.float64)}


def oracle_fused_rows(rows_t, rows_r, n_frames, method):
    """The reference's per-image driver on the oracle detectors' lists (oracle.proben.late_fusion_rows' case split: nobody fired ->
    skipped, one fired -> passed through, both -> fusion; detector order thermal, RGB) -> rows (frame, x1, y1, x2, y2, score, class)
    with the float32 containers of the reference's Instances."""
    from oracle import proben as O
    out = []
    for f in range(n_frames):
        live = [x for x in (_info(rows_t, f), _info(rows_r, f)) if len(x["score"])]
        if not live:
            continue
        if len(live) == 1:
            b, s, c = live[0]["bbox"], live[0]["score"].astype(np.float32), live[0]["class"].astype(np.float32)
        else:
            b, s, c = O.fusion(list(method), *live)
        b = np.asarray(b, dtype=np.float64).astype(np.float32).reshape(-1, 4)
        out += [[f, *b[j], s[j], c[j]] for j in range(len(s))]
    return np.asarray(out, dtype=np.float32).reshape(-1, 7)


def hip_detections(models, frames_t, frames_r, batch=16):
    """Both detectors on every frame pair through the product's FramePairPipeline (each on its own stream): the per-batch result dicts,
    left on the device."""
    from proben_amd.data import resize_shortest_edge_shape
    from proben_amd.pipeline import FramePairPipeline
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    pipe = FramePairPipeline(models, fuse=False)
    out = []
    for b0 in range(0, len(frames_t), batch):
        ft, fr = torch.from_numpy(frames_t[b0:b0 + batch]).cuda(), torch.from_numpy(frames_r[b0:b0 + batch]).cuda()
        dets, _ = pipe([ft, fr], [(512, 640)] * len(ft), new_hw)
        pipe.wait((dets, None))
        out.append((b0, dets))
    torch.cuda.synchronize()
    return out


def detector_rows(batches, which):
    rows = []
    for b0, dets in batches:
        d = dets[which]
        for i, c in enumerate(d["counts"].cpu().tolist()):
            b_, s_, c_ = d["boxes"][i, :c].cpu().numpy(), d["scores"][i, :c].cpu().numpy(), d["classes"][i, :c].cpu().numpy()
            rows += [[b0 + i, *b_[j], s_[j], c_[j]] for j in range(c)]
    return np.asarray(rows, dtype=np.float32).reshape(-1, 7)


def hip_fused_rows(batches, method):
    """The product route of configs[2] behind the detectors: pe_proben_pack_detections -> pe_proben_fuse_batch (fusion.fuse_detections, what
    FramePairPipeline calls) on the device-resident detection lists -> the fused rows of every frame, boxes in the float32 of `Instances`."""
    from proben_amd import fusion as F
    rows = []
    for b0, dets in batches:
        fused = F.fuse_detections(dets, method[0], method[1])
        F.check_candidate_overflow(fused)
        cnt, off = fused["counts"].cpu().tolist(), fused["offsets"].cpu().tolist()
        bx, sc, cl = fused["boxes"].float().cpu().numpy(), fused["scores"].cpu().numpy(), fused["classes"].cpu().numpy()
        for i, (c, o) in enumerate(zip(cnt, off)):
            assert c >= 0, "a fused image ran out of rows"
            rows += [[b0 + i, *bx[o + j], sc[o + j], cl[o + j]] for j in range(c)]
    return np.asarray(rows, dtype=np.float32).reshape(-1, 7)


def _iou_to(r, q):
    iw = (np.minimum(r[3], q[:, 3]) - np.maximum(r[1], q[:, 1])).clip(0)
    ih = (np.minimum(r[4], q[:, 4]) - np.maximum(r[2], q[:, 2])).clip(0)
    inter = iw * ih
    return inter / ((r[3] - r[1]) * (r[4] - r[2]) + (q[:, 3] - q[:, 1]) * (q[:, 4] - q[:, 2]) - inter + 1e-9)


def _unmatched(a, b, n_frames, iou_min=0.9):
    out = []
    for f in range(n_frames):
        ia, ib = np.nonzero(a[:, 0] == f)[0], np.nonzero(b[:, 0] == f)[0]
        used = np.zeros(len(ib), bool)
        for i in ia:
            cand = np.nonzero((~used) & (b[ib, 6] == a[i, 6]))[0]
            if len(cand):
                iou = _iou_to(a[i], b[ib[cand]])
                j = int(iou.argmax())
                if iou[j] >= iou_min:
                    used[cand[j]] = True
                    continue
            out.append(i)
    return np.asarray(out, dtype=np.int64)


def classify_fused(rows, idx, other, own_inputs_only):
    """rows[idx]: one route's fused rows without a same-class IoU >= 0.9 partner among the other route's fused rows.  First match wins:
      class_flip         the other route has a fused row at IoU >= 0.9 with another class
      inherited          one of THIS route's own detector-level detections that the other route's detectors do not have (same class, IoU >=
                         0.9) overlaps the row at IoU >= 0.5: the difference was there before ProbEn (a detector's final-NMS winner flip,
                         threshold flip or missing proposal - profiles/r05_map_flips.json - handed through or averaged in)
      fusion_level       the detectors' lists agree around the row, the other route has a same-class fused row at 0.5 <= IoU < 0.9: a
                         cluster-membership decision (IoU against 0.5, score order) went the other way, or the averaged box moved
      unexplained        nothing of the above"""
    cls = {}
    for i in idx:
        r = rows[i]
        o = other[other[:, 0] == r[0]]
        iou = _iou_to(r, o) if len(o) else np.zeros(0)
        same = o[:, 6] == r[6] if len(o) else np.zeros(0, bool)
        mine = own_inputs_only[(own_inputs_only[:, 0] == r[0]) & (own_inputs_only[:, 6] == r[6])]
        if len(o) and (iou[~same] >= 0.9).any():
            c = "class_flip"
        elif len(mine) and (_iou_to(r, mine) >= 0.5).any():
            c = "inherited"
        elif len(o) and ((iou >= 0.5) & (iou < 0.9) & same).any():
            c = "fusion_level"
        else:
            c = "unexplained"
        cls[int(i)] = c
    return cls


def measure_fused(golden_dir, methods=FUSED_METHODS, flips=True, max_sets=None):
    """Per method and evaluation set: AP table of oracle-detectors -> oracle.proben (the reference's route) and of HIP detectors ->
    pe_proben_fuse_batch (the product's route) against the same ground truth; deltas, mean, standard error; the flip classes of the
    fused rows (and, for scale, the same deltas of the two detectors alone on these sets)."""
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    sds, sets = load_fused_fixture(golden_dir, max_sets=max_sets)      # max_sets: the first few sets (in seed order) only
    models = [GeneralizedRCNN(DetectorConfig(), sd) for sd in sds]
    rec = {"models": "two R101-FPN, seeded random backbones (seeds 1 / 2), RPN + box predictor fitted on the thermal / RGB rendering of the same "
                     "scenes (tests/golden/gen_pseudo_heads.py, gen_fused_map.py; 20 % of the objects invisible to the RGB detector)",
           "north_star_tolerance_points": NORTH_STAR_POINTS, "methods": {}}
    per_method = {m: {"sets": {}, "deltas": [], "totals": {"oracle_only": {}, "hip_only": {}}, "pool_gt": [], "pool_o": [], "pool_h": []} for m in methods}
    off = 0
    for name, ft, fr, gts, ot, orr in sets:
        n = len(ft)
        batches = hip_detections(models, ft, fr)
        ht, hr = detector_rows(batches, 0), detector_rows(batches, 1)
        if flips:      # detector-level rows only one side has (same class, IoU >= 0.9), both detectors together
            o_in_only = np.concatenate([ot[_unmatched(ot[:, :7], ht, n), :7], orr[_unmatched(orr[:, :7], hr, n), :7]])
            h_in_only = np.concatenate([ht[_unmatched(ht, ot[:, :7], n)], hr[_unmatched(hr, orr[:, :7], n)]])
        for method in methods:
            acc = per_method[method]
            ora = oracle_fused_rows(ot, orr, n, method)
            hip = hip_fused_rows(batches, method)
            so, sh = coco_stats(gts, ora), coco_stats(gts, hip)
            acc["deltas"].append((sh - so)[:6] * 100)
            ds, db, un_o, un_h = match_signed(ora, hip, n)
            srec = {"frames": n, "ground_truth_objects": int(sum(len(g[0]) for g in gts)), "oracle_fused_rows": int(len(ora)), "hip_fused_rows": int(len(hip)),
                    "nan_scores": [int(np.isnan(ora[:, 5]).sum()), int(np.isnan(hip[:, 5]).sum())],
                    "oracle": {k: float(so[i] * 100) for i, k in enumerate(NAMES)}, "hip": {k: float(sh[i] * 100) for i, k in enumerate(NAMES)},
                    "delta": {k: float(acc["deltas"][-1][i]) for i, k in enumerate(NAMES)}, "matched_pairs": int(len(ds)), "oracle_only": un_o, "hip_only": un_h,
                    "matched_score_diff_sigma": float(np.nanstd(ds)) if len(ds) else None,
                    "matched_box_abs_max_coord_median_px": float(np.median(np.abs(db).max(1))) if len(db) else None}
            if method == methods[0]:      # the detectors alone on the same frames (method-independent)
                for tag, o_, h_ in (("thermal", ot[:, :7], ht), ("rgb", orr[:, :7], hr)):
                    a, b = coco_stats(gts, o_), coco_stats(gts, h_)
                    srec["detector_" + tag] = {"oracle": {k: float(a[i] * 100) for i, k in enumerate(NAMES[:3])},
                                               "delta": {k: float((b[i] - a[i]) * 100) for i, k in enumerate(NAMES[:3])}}
            if flips:
                co = classify_fused(ora, _unmatched(ora, hip, n), hip, o_in_only)
                ch = classify_fused(hip, _unmatched(hip, ora, n), ora, h_in_only)
                for key, cl in (("oracle_only", co), ("hip_only", ch)):
                    cnt = {}
                    for c in cl.values():
                        cnt[c] = cnt.get(c, 0) + 1
                        acc["totals"][key][c] = acc["totals"][key].get(c, 0) + 1
                    srec["flip_classes_" + key] = cnt
                srec["detector_level_unmatched"] = {"oracle_only": int(len(o_in_only)), "hip_only": int(len(h_in_only)), "oracle_detections": int(len(ot) + len(orr)),
                                                    "hip_detections": int(len(ht) + len(hr))}
            acc["sets"][name] = srec
            acc["pool_gt"] += list(gts)
            for rows, pool in ((ora, acc["pool_o"]), (hip, acc["pool_h"])):
                r = rows.copy()
                r[:, 0] += off
                pool.append(r)
        off += n
    for method in methods:
        acc = per_method[method]
        d = np.asarray(acc["deltas"])
        mrec = {"sets": acc["sets"], "n_sets": int(len(d)), "delta_mean": {k: float(d[:, i].mean()) for i, k in enumerate(NAMES)},
                "delta_std": {k: float(d[:, i].std(ddof=1)) if len(d) > 1 else None for i, k in enumerate(NAMES)},
                "delta_standard_error": {k: float(d[:, i].std(ddof=1) / len(d) ** 0.5) if len(d) > 1 else None for i, k in enumerate(NAMES)}}
        po, ph = coco_stats(acc["pool_gt"], np.concatenate(acc["pool_o"])), coco_stats(acc["pool_gt"], np.concatenate(acc["pool_h"]))
        mrec["pooled"] = {"frames": off, "oracle": {k: float(po[i] * 100) for i, k in enumerate(NAMES)}, "delta": {k: float((ph[i] - po[i]) * 100) for i, k in enumerate(NAMES)}}
        if flips:
            mrec["flip_class_totals"] = acc["totals"]
        rec["methods"]["/".join(method)] = mrec
    return rec
