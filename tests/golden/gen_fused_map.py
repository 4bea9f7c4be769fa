"""Build-container generator for the FUSED-mAP harness (reference: demo/FLIR/demo_probEn.py:198-298 ->
detectron2/evaluation/FLIR_evaluation.py:249-310).  BASELINE's metric is the AP of the rows ProbEn makes of TWO detectors'
lists; tests/golden/gen_pseudo_heads.py pins one detector.  This script
  1. fits a SECOND pseudo-trained R101-FPN (weight seed 2, the 'RGB camera' rendering of the same scenes:
     proben_amd.synthetic.labelled_frames_rgb, 20 % of the objects invisible to it) with gen_pseudo_heads.main ->
     tests/golden/pseudo_heads_r101_rgb.npz (the eight fitted tensors only);
  2. runs the ORACLE (oracle/detector.py = restatement of the reference's CPU path) with both detectors on disjoint 256-frame sets
     and stores what `demo_FLIR_save_predictions.py:133-176` would have written per detection - box, score, class, the K class
     probabilities and the predicted variance - as rows (frame, x1, y1, x2, y2, score, class, p0, p1, p2, var) ->
     tests/golden/fused_map_sets.npz {t_<seed>: thermal detector, r_<seed>: RGB detector}.
tests/test_parity_map_gpu.py feeds these rows to oracle.proben (oracle route) and the HIP detectors' lists to pe_proben_fuse_batch
(product route) and scores both against the same ground truth.  ~10 minutes for the fit + ~25 minutes per set on 8 cores.
    python tests/golden/gen_fused_map.py --seeds 7002,7003,7004,7005"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RGB_HEADS = os.path.join(HERE, "pseudo_heads_r101_rgb.npz")
OUT = os.path.join(HERE, "fused_map_sets.npz")
RGB_SEED = 2


def state_dict(path):
    import proben_amd  # noqa: F401
    from proben_amd.synthetic import synthetic_state_dict
    z = np.load(path)
    sd = synthetic_state_dict(int(z["depth"]), 3, 3, seed=int(z["seed"]))
    for k in z.files:
        if "/" in k:
            sd[k.replace("/", ".")] = torch.from_numpy(z[k])
    return sd, int(z["depth"])


def oracle_rows(frames, sd, depth, tag):
    from PIL import Image
    from oracle import detector as D
    from proben_amd.data import resize_shortest_edge_shape
    spec = D.DetectorSpec(depth=depth)
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    rows, t0 = [], time.time()
    for i in range(len(frames)):
        r = np.array(Image.fromarray(frames[i]).resize((new_hw[1], new_hw[0]), Image.BILINEAR))
        o = D.forward([torch.from_numpy(r).permute(2, 0, 1).float().contiguous()], sd, spec, out_sizes=[(512, 640)])[0]
        b, s, c, p, v = (o[k].numpy() for k in ("boxes", "scores", "classes", "prob_score", "vars"))
        rows += [[i, *b[j], s[j], c[j], *p[j], v[j].reshape(-1)[0]] for j in range(len(s))]
        if i % 64 == 63:
            print(f"{tag}: {i + 1}/{len(frames)} frames, {time.time() - t0:.0f} s", flush=True)
    return np.asarray(rows, dtype=np.float32).reshape(-1, 11)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="7002,7003,7004,7005")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from proben_amd.synthetic import labelled_frames, labelled_frames_rgb
    if not os.path.exists(RGB_HEADS):
        import gen_pseudo_heads
        gen_pseudo_heads.main(out=RGB_HEADS, seed=RGB_SEED, frames_fn=labelled_frames_rgb, n_eval=0)
        torch.set_num_threads(args.threads)
    sd_t, depth = state_dict(os.path.join(HERE, "pseudo_heads_r101.npz"))
    sd_r, _ = state_dict(RGB_HEADS)
    save = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for seed in [int(s) for s in args.seeds.split(",") if s]:
        assert seed != 7001, "7001 is the fitting set"
        for tag, fn, sd in (("t", labelled_frames, sd_t), ("r", labelled_frames_rgb, sd_r)):
            key = f"{tag}_{seed}"
            if key in save:
                continue
            frames, _ = fn(args.frames, seed=seed)
            save[key] = oracle_rows(frames, sd, depth, key)
            save["n_frames"] = np.int64(args.frames)
            np.savez_compressed(OUT, **save)
            print(f"{key}: {len(save[key])} oracle detections", flush=True)


if __name__ == "__main__":
    main()
