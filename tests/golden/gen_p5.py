"""Generate tests/golden/p5_cases.json: the REFERENCE's per-image late-fusion driver
(`apply_late_fusion_and_evaluate`, demo/FLIR/demo_probEn.py:198-298) run here on three synthetic prediction dicts (the J1 schema of
demo_FLIR_save_predictions.py:166-176), 2- and 3-detector, all 11 (score, box) combinations the reference can run without torchvision
- build container only.

What is stubbed and what is the reference's: the function body, `fusion`, `prepare_data`, `nms_bayesian` and the `Instances` / `Boxes`
containers are the reference's own code (loaded from /root/reference); `cv2.imread` (it only supplies H, W) returns a 512 x 640 x 3
array and the evaluator is a recorder whose `process(inputs, outputs)` keeps what the driver hands to FLIREvaluator:
(file_name, image_id, height, width, pred_boxes as float32, scores, pred_classes).  The fixture holds the three input dicts and, per
run, those records - data only."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402
from gen_proben import BOX, SCORE, synth_detector  # noqa: E402

ref = H.load_reference_proben()          # placeholders for cv2 / detectron2 / torchvision while the script body is executed ...
H.install_detectron2_standins()          # ... then the reference's real containers
import torch  # noqa: E402
from detectron2.structures import Boxes, Instances  # noqa: E402

ref.Boxes, ref.Instances = Boxes, Instances


class _CV2:
    @staticmethod
    def imread(path):
        return np.zeros((512, 640, 3), dtype=np.uint8)


ref.cv2 = _CV2


class Recorder:
    def __init__(self):
        self.rows = []

    def reset(self):
        self.rows = []

    def process(self, inputs, outputs):
        for i, o in zip(inputs, outputs):
            inst = o["instances"]
            self.rows.append({"file_name": i["file_name"], "image_id": i["image_id"], "height": int(i["height"]), "width": int(i["width"]),
                              "image_shape": list(i["image"].shape),
                              "boxes": inst.pred_boxes.tensor.numpy().astype(np.float32).reshape(-1, 4).tolist(),
                              "boxes_dtype": str(inst.pred_boxes.tensor.dtype),
                              "scores": np.asarray(inst.scores, dtype=np.float32).tolist(), "scores_dtype": str(inst.scores.dtype),
                              "classes": np.asarray(inst.pred_classes, dtype=np.float32).tolist(), "classes_dtype": str(inst.pred_classes.dtype)})

    def evaluate(self, out_eval_path=None):
        return {"recorded": len(self.rows)}


def j1(per_image, det, names, ids):
    """per_image[i][det] = synth_detector dict (or None = no detection) -> one prediction dict of the J1 schema"""
    out = {k: [] for k in ("image", "boxes", "scores", "classes", "image_id", "class_logits", "probs", "vars")}
    for i, dets in enumerate(per_image):
        d = dets[det]
        out["image"].append(names[det][i])
        out["image_id"].append(ids[det][i])
        if d is None:
            for k in ("boxes", "scores", "classes", "class_logits", "probs", "vars"):
                out[k].append([])
            continue
        out["boxes"].append(d["bbox"].tolist())
        out["scores"].append(d["score"].tolist())
        out["classes"].append(d["class"].tolist())
        out["class_logits"].append(np.log(np.concatenate([d["prob"], 1 - d["prob"].sum(1, keepdims=True)], 1) + 1e-9).tolist())
        out["probs"].append(d["prob"].tolist())
        out["vars"].append(d["vars"].tolist())
    return out


def main():
    rng = np.random.default_rng(20260928)
    # which detectors fire per image: every case of the driver's split (:237-267)
    fire = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1)]
    per_image = []
    for f in fire:
        d1 = synth_detector(rng, int(rng.integers(1, 25)))
        d2 = synth_detector(rng, int(rng.integers(1, 25)), base=d1["bbox"])
        d3 = synth_detector(rng, int(rng.integers(1, 25)), base=d1["bbox"], jitter=5.0)
        per_image.append([d if on else None for d, on in zip((d1, d2, d3), f)])
    n = len(per_image)
    # detector 1 names its files .jpg, detectors 2 / 3 .jpeg; the ids of the three files differ on purpose: the driver takes det_2's
    names = [[f"FLIR_{9000 + i:05d}.jpg" for i in range(n)], [f"FLIR_{9000 + i:05d}.jpeg" for i in range(n)], [f"FLIR_{9000 + i:05d}.jpeg" for i in range(n)]]
    ids = [[1000 + i for i in range(n)], [i * 3 + 7 for i in range(n)], [5000 + i for i in range(n)]]
    dets = [j1(per_image, d, names, ids) for d in range(3)]
    out = {"source": "demo/FLIR/demo_probEn.py:198-298 apply_late_fusion_and_evaluate executed on synthetic prediction dicts "
                     f"(numpy {np.__version__}, torch {torch.__version__}); cv2.imread -> zeros(512, 640, 3); evaluator = recorder",
           "fire": fire, "det_1": dets[0], "det_2": dets[1], "det_3": dets[2], "runs": {}}
    cfg = None
    for sm in SCORE:
        for bm in BOX:
            if sm == "max" and bm == "argmax":
                continue        # nms_1 needs torchvision (absent here): covered by the oracle's restatement only
            for kdet in (2, 3):
                rec = Recorder()
                res = ref.apply_late_fusion_and_evaluate(cfg, rec, dets[0], dets[1], [sm, bm], det_3=dets[2] if kdet == 3 else "")
                assert res == {"recorded": len(rec.rows)}
                out["runs"][f"{sm}_{bm}_{kdet}"] = rec.rows
    path = os.path.join(HERE, "p5_cases.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in list(out["runs"].items())[:4]})
    r = out["runs"]["probEn_v-avg_3"][0]
    print({k: (v if not isinstance(v, list) or len(v) < 4 else "...") for k, v in r.items()})


if __name__ == "__main__":
    main()
