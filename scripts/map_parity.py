"""Collector of the mAP-parity measurement (runs on the GPU box): tests/parity_map.py::measure -> gpurun_out/r05_map_parity.json
(copy to profiles/).    python scripts/map_parity.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import proben_amd  # noqa: E402,F401
from parity_map import measure  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_map_parity.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
rec = measure(os.path.join(ROOT, "tests", "golden"))
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("n_sets", "delta_mean", "delta_std", "matched_pairs_signed")}, indent=1))
