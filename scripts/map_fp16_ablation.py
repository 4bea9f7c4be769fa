"""Which fp16 rounding buys the HIP detector's delta-mAP?  (zero GPU cost.)

The oracle (torch CPU fp32 = the reference's arithmetic) is run on the mAP-parity fixture's frames (tests/golden/pseudo_heads_r101.npz,
tests/test_parity_map_gpu.py) with its TEST-ONLY rounding emulation (oracle/detector.py::EMU) in four settings and every variant is
scored against the SAME ground truth with the same COCO protocol:
    fp32            the reference (the committed oracle rows: re-scored, not re-run)
    fp16            everything the device rounds: input, folded weights, every backbone / FPN activation, head activations
    fp16 + resid32  the same, but the residual stream (block outputs) stays fp32: only convolution OPERANDS are rounded
    fp16 backbone   backbone / FPN rounded, RPN head + box head in fp32
Usage:  python scripts/map_fp16_ablation.py [--frames 256] [--threads 8] [--out profiles/r04_map_fp16_ablation.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_map_fp16_ablation.json"))
    ap.add_argument("--variants", default="fp16,fp16_resid32,fp16_backbone")
    args = ap.parse_args()
    from PIL import Image
    import proben_amd  # noqa: F401
    from oracle import detector as D
    from proben_amd.data import resize_shortest_edge_shape
    from parity_map import coco_stats, load_fixture
    torch.set_num_threads(args.threads)
    z, sd, frames, gts = load_fixture(os.path.join(ROOT, "tests", "golden"))
    n = min(args.frames, len(frames))
    frames, gts = frames[:n], gts[:n]
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    spec = D.DetectorSpec(depth=int(z["depth"]))
    names = ["AP", "AP50", "AP75", "APs", "APm", "APl"]
    ora_rows = z["oracle_rows"]
    ora_rows = ora_rows[ora_rows[:, 0] < n]
    base = coco_stats(gts, ora_rows)
    rec = {"frames": n, "ground_truth_objects": int(sum(len(g[0]) for g in gts)),
           "fp32": {"detections": int(len(ora_rows)), **{k: float(base[i] * 100) for i, k in enumerate(names)}}}
    settings = {"fp16": {"backbone": True, "heads": True}, "fp16_resid32": {"backbone": True, "heads": True, "resid32": True},
                "fp16_backbone": {"backbone": True}}
    for name in args.variants.split(","):
        D.EMU = settings[name]
        rows, t0 = [], time.time()
        for i in range(n):
            r = np.array(Image.fromarray(frames[i]).resize((new_hw[1], new_hw[0]), Image.BILINEAR))
            o = D.forward([torch.from_numpy(r).permute(2, 0, 1).float().contiguous()], sd, spec, out_sizes=[(512, 640)])[0]
            b, s, c = o["boxes"].numpy(), o["scores"].numpy(), o["classes"].numpy()
            rows += [[i, *b[j], s[j], c[j]] for j in range(len(s))]
            if i % 32 == 31:
                print(f"{name}: {i + 1}/{n} frames, {time.time() - t0:.0f} s", flush=True)
        D.EMU = None
        rows = np.asarray(rows, dtype=np.float32).reshape(-1, 7)
        st = coco_stats(gts, rows)
        rec[name] = {"detections": int(len(rows)), **{k: float(st[i] * 100) for i, k in enumerate(names)},
                     "delta_vs_fp32": {k: float((st[i] - base[i]) * 100) for i, k in enumerate(names)}}
        np.save(os.path.splitext(args.out)[0] + f"_{name}_rows.npy", rows)
        json.dump(rec, open(args.out, "w"), indent=1)
        print(json.dumps(rec[name], indent=1), flush=True)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
