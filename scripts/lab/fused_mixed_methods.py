"""Lab (round 6): which half of probEn / v-avg carries the +0.12 AP of the product route?  The two MIXED method pairs over the same sets.
    python scripts/lab/fused_mixed_methods.py [n_sets]        (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import proben_amd  # noqa: E402,F401
from parity_map import measure_fused  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rec = measure_fused(os.path.join(ROOT, "tests", "golden"), methods=[("probEn", "s-avg"), ("avg", "v-avg")], flips=False, max_sets=n)
for m, r in rec["methods"].items():
    print(m, "sets", r["n_sets"], "mean delta AP / AP50 / AP75", [round(r["delta_mean"][k], 3) for k in ("AP", "AP50", "AP75")],
          "se", [round(r["delta_standard_error"][k], 3) for k in ("AP", "AP50", "AP75")], "pooled", [round(r["pooled"]["delta"][k], 3) for k in ("AP", "AP50", "AP75")])
