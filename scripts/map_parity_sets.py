"""More evaluation sets for the mAP-parity measurement ("4 disjoint 256-frame sets -> mean +- std of the deltas").
Runs the ORACLE (torch CPU fp32 = the reference's arithmetic) with the pseudo-trained fixture weights (tests/golden/
pseudo_heads_r101.npz) on `labelled_frames(n, seed)` for new seeds and stores its detections next to the fixture:
tests/golden/pseudo_heads_r101_sets.npz {rows_<seed>: [n_det, 7] (frame, x1, y1, x2, y2, score, class)}.  Build container only
(~12 minutes per 256-frame set on 8 cores); tests/test_parity_map_gpu.py scores the HIP detector against the same sets.
    python scripts/map_parity_sets.py --seeds 7001,7002,7003"""
import argparse
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
    ap.add_argument("--seeds", default="7001,7002,7003")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    from PIL import Image
    import proben_amd  # noqa: F401
    from oracle import detector as D
    from proben_amd.data import resize_shortest_edge_shape
    from proben_amd.synthetic import labelled_frames
    from test_parity_map_gpu import load_fixture
    torch.set_num_threads(args.threads)
    z, sd, _, _ = load_fixture(os.path.join(ROOT, "tests", "golden"))
    spec = D.DetectorSpec(depth=int(z["depth"]))
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    out_path = os.path.join(ROOT, "tests", "golden", "pseudo_heads_r101_sets.npz")
    save = dict(np.load(out_path)) if os.path.exists(out_path) else {}
    for seed in [int(s) for s in args.seeds.split(",") if s]:
        if f"rows_{seed}" in save:
            continue
        assert seed != int(z["eval_seed"]) and seed != int(z["seed"]), "evaluation sets must be disjoint from the fixture's and the fitting set"
        frames, _ = labelled_frames(args.frames, seed=seed)
        rows, t0 = [], time.time()
        for i in range(len(frames)):
            r = np.array(Image.fromarray(frames[i]).resize((new_hw[1], new_hw[0]), Image.BILINEAR))
            o = D.forward([torch.from_numpy(r).permute(2, 0, 1).float().contiguous()], sd, spec, out_sizes=[(512, 640)])[0]
            b, s, c = o["boxes"].numpy(), o["scores"].numpy(), o["classes"].numpy()
            rows += [[i, *b[j], s[j], c[j]] for j in range(len(s))]
            if i % 64 == 63:
                print(f"seed {seed}: {i + 1}/{len(frames)} frames, {time.time() - t0:.0f} s", flush=True)
        save[f"rows_{seed}"] = np.asarray(rows, dtype=np.float32).reshape(-1, 7)
        save["n_frames"] = np.int64(args.frames)
        np.savez_compressed(out_path, **save)
        print(f"seed {seed}: {len(rows)} oracle detections", flush=True)


if __name__ == "__main__":
    main()
