"""Where do the HIP detector and the oracle differ on the pseudo-trained fixture (tests/golden/pseudo_heads_r101.npz)?
Per-detection matching of the two lists: counts, matched pairs (same class, IoU >= 0.9), signed score / box differences,
what the unmatched detections look like.    python scripts/map_parity_diff.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import proben_amd  # noqa: E402,F401
from test_parity_map_gpu import coco_stats, load_fixture  # noqa: E402
from proben_amd.data import resize_shortest_edge_shape  # noqa: E402
from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN  # noqa: E402

z, sd, frames, gts = load_fixture(os.path.join(ROOT, "tests", "golden"))
ora = z["oracle_rows"]
model = GeneralizedRCNN(DetectorConfig(), sd)
new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
rows = []
for b0 in range(0, len(frames), 16):
    fr = torch.from_numpy(frames[b0:b0 + 16]).cuda()
    det = model.forward_batch(fr, out_sizes=[(512, 640)] * len(fr), resize_to=new_hw)
    cnt = det["counts"].cpu().tolist()
    for i, c in enumerate(cnt):
        bx, sc, cl = det["boxes"][i, :c].cpu().numpy(), det["scores"][i, :c].cpu().numpy(), det["classes"][i, :c].cpu().numpy()
        rows += [[b0 + i, *bx[j], sc[j], cl[j]] for j in range(c)]
hip = np.asarray(rows, dtype=np.float32).reshape(-1, 7)
print("detections: oracle", len(ora), "hip", len(hip))
ds, db, un_o, un_h = [], [], [], []
for f in range(len(frames)):
    o, h = ora[ora[:, 0] == f], hip[hip[:, 0] == f]
    used = np.zeros(len(h), bool)
    for r in o:
        best, bj = 0, -1
        for j, q in enumerate(h):
            if used[j] or q[6] != r[6]:
                continue
            x1, y1, x2, y2 = max(r[1], q[1]), max(r[2], q[2]), min(r[3], q[3]), min(r[4], q[4])
            inter = max(0, x2 - x1) * max(0, y2 - y1)
            iou = inter / ((r[3] - r[1]) * (r[4] - r[2]) + (q[3] - q[1]) * (q[4] - q[2]) - inter + 1e-9)
            if iou > best:
                best, bj = iou, j
        if best >= 0.9:
            used[bj] = True
            ds.append(h[bj, 5] - r[5]); db.append(np.abs(h[bj, 1:5] - r[1:5]).max())
        else:
            un_o.append(r[5])
    un_h += list(h[~used, 5])
ds, db = np.asarray(ds), np.asarray(db)
print(f"matched {len(ds)}: score diff (hip - oracle) mean {ds.mean():+.5f} median {np.median(ds):+.5f} |max| {np.abs(ds).max():.4f}; box |max diff| median {np.median(db):.3f} p99 {np.percentile(db, 99):.3f} px")
print(f"oracle-only {len(un_o)}: scores", np.round(np.sort(un_o)[:12], 3), "... median", np.median(un_o) if un_o else None)
print(f"hip-only {len(un_h)}: scores", np.round(np.sort(un_h)[:12], 3), "... median", np.median(un_h) if un_h else None)
so, sh = coco_stats(gts, ora), coco_stats(gts, hip)
print("AP/AP50/AP75 oracle", np.round(so[:3] * 100, 3), "hip", np.round(sh[:3] * 100, 3))
# what if the HIP list is scored with the ORACLE's scores on the matched pairs?  (separates score noise from box / set differences)
