"""The per-image late-fusion driver (demo/FLIR/demo_probEn.py:198-298 `apply_late_fusion_and_evaluate`) and
the prediction-JSON interchange (demo/FLIR/demo_FLIR_save_predictions.py:83-176), batched for the GPU.

J1 schema (kept readable / writable for drop-in): one dict of parallel per-image lists
  image, boxes [n][4], scores [n], classes [n], image_id, class_logits [n][K+1], probs [n][K], vars [n][1]
with detections of classes > 2 dropped."""
import json
import time

import numpy as np
import torch

from . import fusion as F
from .structures import Boxes, Instances

J1_KEYS = ["image", "boxes", "scores", "classes", "image_id", "class_logits", "probs", "vars"]


def predictions_to_j1(file_names, image_ids, instances_list, max_class=2):
    """list of Instances (CPU or GPU) -> the reference's prediction dict (demo_FLIR_save_predictions.py:133-176)."""
    out = {k: [] for k in J1_KEYS}
    for name, iid, inst in zip(file_names, image_ids, instances_list):
        inst = inst.to("cpu")
        boxes = inst.pred_boxes.tensor.tolist()
        scores = inst.scores.tolist()
        classes = inst.pred_classes.tolist()
        logits = inst.class_logits.tolist() if inst.has("class_logits") else [[] for _ in boxes]
        probs = inst.prob_score.tolist() if inst.has("prob_score") else [[] for _ in boxes]
        var = inst.vars.tolist() if inst.has("vars") else [[1.0] for _ in boxes]
        keep = [j for j in range(len(boxes)) if classes[j] <= max_class]
        out["image"].append(name)
        out["boxes"].append([boxes[j] for j in keep])
        out["scores"].append([scores[j] for j in keep])
        out["classes"].append([classes[j] for j in keep])
        out["image_id"].append(iid)
        out["class_logits"].append([logits[j] for j in keep])
        out["probs"].append([probs[j] for j in keep])
        out["vars"].append([var[j] for j in keep])
    return out


def write_j1(path, pred):
    with open(path, "w") as f:
        json.dump(pred, f, indent=2)


def read_j1(path):
    with open(path) as f:
        d = json.load(f)
    for k in J1_KEYS:
        assert k in d, f"{path}: missing key '{k}' of the prediction schema"
    return d


def shard_j1(det, index_range):
    """The slice of a prediction dict one rank works on (contiguous InferenceSampler block: rank order == image order)."""
    lo, hi = (index_range.start, index_range.stop) if len(index_range) else (0, 0)
    return {k: v[lo:hi] for k, v in det.items()}


def _info(det, i):
    return {"img_name": det["image"][i], "bbox": det["boxes"][i], "score": det["scores"][i], "class": det["classes"][i],
            "class_logits": det["class_logits"][i], "prob": det["probs"][i], "vars": det["vars"][i]}


def late_fusion(dets, method, device="cuda"):
    """dets: 2 or 3 J1 dicts over the same images (order = detector order).  Returns per-image
    (boxes float64 [m,4] | None, scores f32, classes f32); None = skipped image (no detector fired).
    Case split of demo_probEn.py:237-267: 0 detectors -> skip, 1 -> passthrough, >= 2 -> fusion of the
    non-empty lists in order.  All images needing fusion go through ONE batched launch."""
    n_img = len(dets[1]["image"]) if len(dets) > 1 else len(dets[0]["image"])      # the reference loops over det_2's images (:205)
    results = [None] * n_img
    batch, where = [], []
    for i in range(n_img):
        infos = [_info(d, i) for d in dets]
        live = [x for x in infos if len(x["bbox"]) > 0]
        if len(live) == 0:
            continue
        if len(live) == 1:
            x = live[0]
            results[i] = (np.array(x["bbox"], dtype=np.float64), torch.tensor(x["score"], dtype=torch.float32),
                          torch.tensor(x["class"], dtype=torch.float32))
            continue
        batch.append(live)
        where.append(i)
    if batch:
        if method[0] == "max" and method[1] == "argmax":
            for i, live in zip(where, batch):
                b, s, c = F.fusion(method, *live)
                results[i] = (b.double().numpy(), s, c)
        else:
            b, s, p, v, c, offs = F.pack_infos(batch, device)
            out = F.fuse_batch(b, s, p, v, c, offs, method[0], method[1])
            cnt = out["counts"].cpu().numpy()
            ob, os_, oc = out["boxes"].cpu().numpy(), out["scores"].cpu(), out["classes"].cpu()
            oh = offs.cpu().numpy()
            for j, i in enumerate(where):
                sl = slice(oh[j], oh[j] + cnt[j])
                results[i] = (ob[sl], os_[sl], oc[sl])
    return results


def apply_late_fusion_and_evaluate(cfg, evaluator, det_1, det_2, method, det_3="", image_hw=None, device="cuda",
                                   img_folder="../../../Datasets/FLIR/val/thermal_8_bit/"):
    """Same call as the reference (demo_probEn.py:198).  `image_hw`: {image_id: (H, W)} from the dataset
    json (the reference re-reads every thermal JPEG just for its shape); default 512 x 640 (FLIR).
    `img_folder`: the prefix the reference hard-codes into the `file_name` it hands to the evaluator (:200,271).
    What the evaluator receives per image is pinned by tests/golden/p5_cases.json (the reference's function run with a recording
    evaluator): tests/test_pipeline_gpu.py::test_late_fusion_driver_reproduces_the_references_records."""
    evaluator.reset()
    print("Method: ", method)
    start = time.time()
    dets = [det_1, det_2] + ([det_3] if det_3 else [])
    fused = late_fusion(dets, method, device)
    for i, r in enumerate(fused):
        if r is None:
            continue
        iid = det_2["image_id"][i]
        H, W = (image_hw or {}).get(iid, (512, 640))
        boxes, scores, classes = r
        inst = Instances((H, W))
        inst.pred_boxes = Boxes(torch.as_tensor(np.asarray(boxes), dtype=torch.float32).reshape(-1, 4))
        inst.scores = scores
        inst.pred_classes = classes
        name = img_folder + det_1["image"][i].split(".")[0] + ".jpeg"
        evaluator.process([{"file_name": name, "height": H, "width": W, "image_id": iid}], [{"instances": inst}])
    print("Average time:", (time.time() - start) / max(len(det_2["image"]), 1))
    return evaluator.evaluate()


def fused_rows_device(fused, image_ids, valid_classes=(0, 1, 2, 5, 7, 16)):
    """Evaluation rows of one batch ON THE DEVICE: [n,7] float64 = (image_id, x, y, w, h, score, category_id) with the
    evaluator's class whitelist / remap (FLIR_evaluation.py:313-382) applied - the tensor form that crosses ranks in
    comm.all_gather_rows (one RCCL all-gather instead of the reference's pickled lists over gloo).
    `fused`: result of fusion.fuse_detections (or any dict with boxes [B*S,4], scores, classes, counts, offsets, stride)."""
    B = fused["counts"].numel()
    S = fused["stride"]
    dev = fused["scores"].device
    F.check_candidate_overflow(fused)   # evaluation rows leave for the host / other ranks from here: the one sync point
    slot = torch.arange(S, device=dev).unsqueeze(0)                                  # [1,S]
    live = slot < fused["counts"].unsqueeze(1)                                       # [B,S]
    cls = fused["classes"].view(B, S).to(torch.int64)
    ok = torch.zeros_like(live)
    for c in valid_classes:
        ok |= cls == c
    live &= ok
    cat = torch.where((cls == 5) | (cls == 7), torch.full_like(cls, 2), cls)
    b = fused["boxes"].view(B, S, 4).double()
    ids = torch.as_tensor(image_ids, dtype=torch.float64, device=dev).view(B, 1).expand(B, S)
    rows = torch.stack([ids, b[..., 0], b[..., 1], b[..., 2] - b[..., 0], b[..., 3] - b[..., 1],
                        fused["scores"].view(B, S).double(), cat.double()], dim=2)
    return rows[live]                                                               # image-major, score order kept
