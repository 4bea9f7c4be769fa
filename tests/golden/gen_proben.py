"""Generate tests/golden/proben_cases.npz by running the REFERENCE's own
``fusion`` (demo/FLIR/demo_probEn.py:189-196) on seeded synthetic detections.

Run in the build container only:  python tests/golden/gen_proben.py
The .npz holds inputs and the reference's outputs (data only).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import load_reference_proben  # noqa: E402

SCORE = ["probEn", "avg", "max"]
BOX = ["v-avg", "s-avg", "avg", "argmax"]


def synth_detector(rng, n, base=None, jitter=3.0, k=3):
    """n detections: boxes inside 640x512, probs with max > 0.5, class = argmax."""
    if base is not None and len(base) > 0:
        take = rng.integers(0, len(base), size=n)
        boxes = base[take] + rng.normal(0, jitter, size=(n, 4))
    else:
        x1 = rng.uniform(0, 560, n)
        y1 = rng.uniform(0, 440, n)
        w = rng.uniform(10, 160, n)
        h = rng.uniform(10, 160, n)
        boxes = np.stack([x1, y1, np.minimum(x1 + w, 640), np.minimum(y1 + h, 512)], 1)
    boxes = np.clip(boxes, 0, [640, 512, 640, 512])
    probs = []
    while len(probs) < n:
        p = rng.dirichlet([1, 1, 1, 0.3])[:k]
        if p.max() > 0.5:
            probs.append(p)
    probs = np.asarray(probs).reshape(n, k)
    cls = probs.argmax(1)
    score = probs.max(1)
    var = rng.uniform(0.5, 3.0, size=(n, 1))
    # round-trip through float32 like the JSON written from f32 tensors
    f = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)  # noqa: E731
    return {"bbox": f(boxes), "score": f(score), "class": cls.astype(np.int64),
            "prob": f(probs), "vars": f(var)}


def to_info(d, name="img"):
    return {"img_name": name, "bbox": d["bbox"].tolist(), "score": d["score"].tolist(),
            "class": d["class"].tolist(), "class_logits": d["prob"].tolist(),
            "prob": d["prob"].tolist(), "vars": d["vars"].tolist()}


def make_cases():
    rng = np.random.default_rng(20260927)
    cases = []
    # 2-detector random cases with cross-detector near-duplicates
    for n1, n2 in [(6, 5), (30, 25), (100, 100), (1, 1), (2, 40)]:
        d1 = synth_detector(rng, n1)
        nd = int(0.5 * n2)
        d2a = synth_detector(rng, nd, base=d1["bbox"]) if nd else None
        d2b = synth_detector(rng, n2 - nd)
        d2 = {k: np.concatenate([d2a[k], d2b[k]]) for k in d2b} if nd else d2b
        # near-duplicates mostly share the class of their source: copy from a random d1 row
        cases.append([d1, d2])
    # 3-detector cases
    for n in [(8, 7, 9), (60, 50, 70), (100, 100, 100)]:
        d1 = synth_detector(rng, n[0])
        d2 = synth_detector(rng, n[1], base=d1["bbox"])
        d3 = synth_detector(rng, n[2], base=d1["bbox"], jitter=5.0)
        cases.append([d1, d2, d3])
    # one big cluster (m >= 8) of the same class + class disagreement inside it
    d1 = synth_detector(rng, 12, base=np.array([[100.0, 100, 220, 260]]), jitter=2.0)
    d1["class"][:] = 1
    d1["prob"][:, :] = np.float32(0.05)
    d1["prob"][:, 1] = np.linspace(0.55, 0.85, 12).astype(np.float32)
    d1["score"] = d1["prob"][:, 1].copy()
    d2 = synth_detector(rng, 12, base=np.array([[100.0, 100, 220, 260]]), jitter=2.0)
    d2["class"][:] = 1  # same class band so they cluster ...
    d2["prob"][:, :] = np.float32(0.01)
    d2["prob"][:, 0] = np.float32(0.44)  # ... but probability mass on class 0 / background
    d2["prob"][:, 1] = np.linspace(0.51, 0.53, 12).astype(np.float32)
    d2["score"] = d2["prob"][:, 1].copy()
    cases.append([d1, d2])
    # ties: equal scores in different (non-interacting) classes / far-apart boxes
    d1 = synth_detector(rng, 6)
    d1["bbox"] = np.array([[10, 10, 60, 60], [200, 10, 260, 70], [400, 10, 470, 80],
                           [10, 300, 70, 380], [200, 300, 280, 400], [400, 300, 500, 420]], dtype=np.float64)
    d1["score"][:] = np.float32(0.75)
    d2 = {k: v.copy() for k, v in d1.items()}
    d2["bbox"] = d2["bbox"] + 2.0
    d2["score"][:] = np.float32(0.625)
    cases.append([d1, d2])
    return cases


def main():
    ref = load_reference_proben()
    out = {}
    cases = make_cases()
    out["num_cases"] = np.int64(len(cases))
    for ci, dets in enumerate(cases):
        out[f"c{ci}_ndet"] = np.int64(len(dets))
        for di, d in enumerate(dets):
            for k, v in d.items():
                out[f"c{ci}_d{di}_{k}"] = v
        infos = [to_info(d) for d in dets]
        for sm in SCORE:
            for bm in BOX:
                if sm == "max" and bm == "argmax":
                    continue  # nms_1 route needs torchvision (absent): oracle-only
                b, s, c = ref.fusion([sm, bm], *infos)
                out[f"c{ci}_{sm}_{bm}_boxes"] = np.asarray(b, dtype=np.float64).reshape(-1, 4)
                out[f"c{ci}_{sm}_{bm}_scores"] = s.numpy().astype(np.float32)
                out[f"c{ci}_{sm}_{bm}_classes"] = c.numpy().astype(np.float32)
    # the unused binary form (demo_probEn.py:24-30), needed for K = 1 (KAIST)
    rng = np.random.default_rng(7)
    vecs = [rng.uniform(0.05, 0.95, size=m) for m in (2, 3, 5)]
    out["binary_in"] = np.concatenate(vecs)
    out["binary_len"] = np.asarray([len(v) for v in vecs])
    out["binary_out"] = np.asarray([ref.bayesian_fusion(v) for v in vecs])
    path = os.path.join(HERE, "proben_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
