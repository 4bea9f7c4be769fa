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
