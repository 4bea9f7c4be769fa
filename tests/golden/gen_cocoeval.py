"""Generate tests/golden/cocoeval_case.json: synthetic GT + detections and the 12 COCO stats, per-class AP
computed by the REFERENCE's vendored evaluator (detectron2/pycocotools/{coco,cocoeval}.py) - build container
only.  The one third-party piece, pycocotools' C `_mask.iou` (bbIou), is absent: the stand-in is
proben_amd.evaluation.bbox_iou_xywh, so this fixture pins the matching / accumulation / summary logic, not
the IoU arithmetic itself ("parity unpinned" there)."""
import contextlib
import io
import json
import os
import sys
import types

import numpy as np

if not hasattr(np, "float"):
    np.float = float  # removed alias still used by the reference's vendored cocoeval.py (accumulate)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as H  # noqa: E402

H.install_detectron2_standins()
import proben_amd  # noqa: E402,F401
from proben_amd.evaluation import bbox_iou_xywh  # noqa: E402

m = types.ModuleType("pycocotools._mask")
m.iou = lambda dt, gt, iscrowd: bbox_iou_xywh(dt, gt, iscrowd)
m.encode = m.decode = m.area = m.merge = m.frPyObjects = m.toBbox = None
sys.modules["pycocotools._mask"] = m
sys.modules["pycocotools"]._mask = m
from detectron2.pycocotools.coco import COCO  # noqa: E402
from detectron2.pycocotools.cocoeval import COCOeval  # noqa: E402


def synth(seed=3, n_img=12):
    rng = np.random.default_rng(seed)
    images = [{"id": 100 + i, "file_name": f"thermal_8_bit/im{i}.jpeg", "height": 512, "width": 640} for i in range(n_img)]
    cats = [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"}, {"id": 3, "name": "car"}]
    anns, dets = [], []
    aid = 1
    for im in images:
        for _ in range(int(rng.integers(2, 7))):
            w, h = rng.uniform(8, 220), rng.uniform(8, 200)
            x, y = rng.uniform(0, 640 - w), rng.uniform(0, 512 - h)
            c = int(rng.integers(1, 4))
            crowd = int(rng.random() < 0.1)
            anns.append({"id": aid, "image_id": im["id"], "category_id": c, "bbox": [x, y, w, h], "area": w * h, "iscrowd": crowd})
            aid += 1
            if rng.random() < 0.8:  # a detection near this GT
                j = rng.normal(0, 4 + 0.05 * w, 4)
                dets.append({"image_id": im["id"], "category_id": c if rng.random() < 0.9 else int(rng.integers(1, 4)),
                             "bbox": [x + j[0], y + j[1], max(w + j[2], 2), max(h + j[3], 2)], "score": float(np.float32(rng.uniform(0.5, 1)))})
        for _ in range(int(rng.integers(0, 4))):  # false positives
            w, h = rng.uniform(8, 150), rng.uniform(8, 150)
            dets.append({"image_id": im["id"], "category_id": int(rng.integers(1, 4)),
                         "bbox": [rng.uniform(0, 640 - w), rng.uniform(0, 512 - h), w, h], "score": float(np.float32(rng.uniform(0.5, 0.9)))})
    dets[3]["score"] = dets[7]["score"]  # a score tie
    return {"images": images, "annotations": anns, "categories": cats}, dets


def main():
    gt, dets = synth()
    path = "/tmp/pe_coco_gt.json"
    json.dump(gt, open(path, "w"))
    with contextlib.redirect_stdout(io.StringIO()):
        coco = COCO(path)
        dt = coco.loadRes(json.loads(json.dumps(dets)))
        ev = COCOeval(coco, dt, "bbox")
        ev.evaluate()
        ev.accumulate()
        ev.summarize()
    prec = ev.eval["precision"]
    per_class = []
    for k in range(prec.shape[2]):
        p = prec[:, :, k, 0, -1]
        p = p[p > -1]
        per_class.append(float(np.mean(p)) if p.size else float("nan"))
    out = {"gt": gt, "dets": dets, "stats": [float(s) for s in ev.stats], "per_class_ap": per_class,
           "precision_sum": float(prec[prec > -1].sum()), "recall_sum": float(ev.eval["recall"][ev.eval["recall"] > -1].sum())}
    p = os.path.join(HERE, "cocoeval_case.json")
    json.dump(out, open(p, "w"))
    print("wrote", p, os.path.getsize(p), out["stats"])


if __name__ == "__main__":
    main()
