"""Lab (round 6): where does the +0.12 AP of the product route's probEn / v-avg rows over the oracle route's come from?
For the first N evaluation sets: both routes' fused rows, then
  * how many scores are exactly 1.0f / >= 0.999 in each route;
  * the AP delta with the rows in file order (what the record holds), in a CANONICAL order (image, score descending, x1, y1: ties between equal
    scores are then broken by geometry in both routes alike) and under random permutations of the file order (the spread tie order alone makes);
  * the AP delta with every score rounded to float32(1 - 2^-k) steps (k = 10): coarser ties, for scale.
    python scripts/lab/fused_tie_probe.py [n_sets]        (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import proben_amd  # noqa: E402,F401
from parity_map import coco_stats, hip_detections, hip_fused_rows, load_fused_fixture, oracle_fused_rows  # noqa: E402
from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN  # noqa: E402


def canon(rows):
    k = np.lexsort((rows[:, 2], rows[:, 1], -np.nan_to_num(rows[:, 5], nan=-1.0), rows[:, 0]))
    return rows[k]


def main():
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    method = ("probEn", "v-avg")
    sds, sets = load_fused_fixture(os.path.join(ROOT, "tests", "golden"), max_sets=n_sets)
    models = [GeneralizedRCNN(DetectorConfig(), sd) for sd in sds]
    rng = np.random.default_rng(0)
    acc = {"file": [], "canonical": [], "perm_spread_o": [], "perm_spread_h": [], "perm_delta": []}
    ones = []
    for name, ft, fr, gts, ot, orr in sets:
        n = len(ft)
        batches = hip_detections(models, ft, fr)
        ora, hip = oracle_fused_rows(ot, orr, n, method), hip_fused_rows(batches, method)
        ap = lambda r: coco_stats(gts, r)[:3] * 100
        d_file = ap(hip) - ap(ora)
        d_can = ap(canon(hip)) - ap(canon(ora))
        po = np.array([ap(ora[rng.permutation(len(ora))]) for _ in range(6)])
        ph = np.array([ap(hip[rng.permutation(len(hip))]) for _ in range(6)])
        acc["file"].append(d_file); acc["canonical"].append(d_can)
        acc["perm_spread_o"].append(po.std(0, ddof=1)); acc["perm_spread_h"].append(ph.std(0, ddof=1)); acc["perm_delta"].append(ph.mean(0) - po.mean(0))
        so, sh = ora[:, 5], hip[:, 5]
        ones.append((int((so == 1.0).sum()), int((sh == 1.0).sum()), int((so >= 0.999).sum()), int((sh >= 0.999).sum()), len(so), len(sh)))
        print(name, "file", np.round(d_file, 3), "canonical", np.round(d_can, 3), "mean over permutations", np.round(acc["perm_delta"][-1], 3),
              "| std of AP / AP50 / AP75 over permutations: oracle", np.round(acc["perm_spread_o"][-1], 3), "hip", np.round(acc["perm_spread_h"][-1], 3),
              "| scores == 1.0f: oracle %d hip %d; >= 0.999: %d %d of %d / %d" % ones[-1], flush=True)
    f = lambda k: (np.round(np.mean(acc[k], 0), 3), "se", np.round(np.std(acc[k], 0, ddof=1) / np.sqrt(len(acc[k])), 3))
    print("MEAN over", len(sets), "sets: delta in file order", *f("file"), "| canonical order", *f("canonical"), "| mean over random orders", *f("perm_delta"))
    print("tie-order spread of ONE route's AP / AP50 / AP75 (std over random file orders, mean over sets): oracle", np.round(np.mean(acc["perm_spread_o"], 0), 3), "hip", np.round(np.mean(acc["perm_spread_h"], 0), 3))
    o1, h1, o9, h9, no, nh = np.sum(ones, 0)
    print("scores exactly 1.0f: oracle %d (%.2f %%), hip %d (%.2f %%); >= 0.999: %.2f %% / %.2f %%" % (o1, 100 * o1 / no, h1, 100 * h1 / nh, 100 * o9 / no, 100 * h9 / nh))


if __name__ == "__main__":
    main()
