"""The FUSED-mAP record: tests/parity_map.py::measure_fused -> gpurun_out/r06_fused_map.json (copied to profiles/).
    python scripts/fused_map.py [out.json]        (GPU box; ~1 minute per evaluation set and method)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import proben_amd  # noqa: E402,F401
from parity_map import measure_fused  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06_fused_map.json")
    rec = measure_fused(os.path.join(ROOT, "tests", "golden"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"), indent=1)
    for m, r in rec["methods"].items():
        print(m, "sets", r["n_sets"], "mean", {k: round(v, 3) for k, v in r["delta_mean"].items()}, "se", {k: (round(v, 3) if v is not None else None) for k, v in r["delta_standard_error"].items()},
              "pooled", {k: round(v, 3) for k, v in r["pooled"]["delta"].items()}, r.get("flip_class_totals"))
        for name, s in r["sets"].items():
            print("  ", name, "oracle AP/AP50/AP75 %.2f %.2f %.2f" % tuple(s["oracle"][k] for k in ("AP", "AP50", "AP75")), "delta", {k: round(v, 3) for k, v in s["delta"].items() if k in ("AP", "AP50", "AP75")},
                  "rows", s["oracle_fused_rows"], s["hip_fused_rows"], s.get("detector_thermal"), s.get("detector_rgb"))


if __name__ == "__main__":
    main()
