"""End-to-end parity as ONE number: detections of the HIP detector (fp16 MFMA convs, fp32 post-ops) scored against the
detections of the fp32 CPU oracle (= restatement of the reference's MODEL.DEVICE=cpu path) with the COCO evaluator -
oracle rows are the ground truth, HIP rows the detections, AP50 / AP are reported per class and overall.

    python scripts/e2e_parity.py [--images 64] [--depth 101] [--out profiles/r02_e2e_parity.json]

Same seeded synthetic weights and full-size (640x512 -> 800x1000, padded 800x1024) synthetic frames on both sides.
Random-init weights do not separate scores the way a trained model does, so near-ties in top-k / NMS / the 0.5 score
threshold flip under fp16 feature noise: the figure is a LOWER bound for what trained weights would give."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def run(n_images=64, depth=101, seed=1, threads=16, batch=16):
    import proben_amd  # noqa: F401
    from oracle import detector as D
    from proben_amd import evaluation
    from proben_amd.data import resize_shortest_edge_shape
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_images, synthetic_state_dict
    sd = synthetic_state_dict(depth, 3, 3, seed=seed)
    imgs = synthetic_images(n_images, seed=300 + seed)                 # uint8 [n,512,640,3]
    model = GeneralizedRCNN(DetectorConfig(), sd)
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    # ---- HIP: raw frames -> Pillow-exact resize on the GPU -> detector ----
    hip = []
    for b0 in range(0, n_images, batch):
        fr = torch.from_numpy(imgs[b0:b0 + batch]).cuda()
        det = model.forward_batch(fr, out_sizes=[(512, 640)] * len(fr), resize_to=new_hw)
        cnt = det["counts"].cpu().tolist()
        for i, c in enumerate(cnt):
            hip.append((det["boxes"][i, :c].cpu().numpy(), det["scores"][i, :c].cpu().numpy(), det["classes"][i, :c].cpu().numpy()))
    # ---- oracle: the same frames through Pillow (what the reference does for 3-channel inputs) -> CPU fp32 ----
    from PIL import Image
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    spec = D.DetectorSpec(depth=depth)
    t0 = time.time()
    ora = []
    for i in range(n_images):
        r = np.array(Image.fromarray(imgs[i]).resize((new_hw[1], new_hw[0]), Image.BILINEAR))
        x = torch.from_numpy(r).permute(2, 0, 1).float().contiguous()
        o = D.forward([x], sd, spec, out_sizes=[(512, 640)])[0]
        ora.append((o["boxes"].numpy(), o["scores"].numpy(), o["classes"].numpy()))
    cpu_s = time.time() - t0
    # ---- COCO evaluation: oracle = GT, HIP = detections ----
    images = [{"id": i, "height": 512, "width": 640, "file_name": f"{i}.jpeg"} for i in range(n_images)]
    cats = [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"}, {"id": 3, "name": "car"}]
    anns, dets, aid = [], [], 1
    for i in range(n_images):
        for b, s, c in zip(*ora[i]):
            w, h = float(b[2] - b[0]), float(b[3] - b[1])
            anns.append({"id": aid, "image_id": i, "category_id": int(c) + 1, "bbox": [float(b[0]), float(b[1]), w, h], "area": w * h, "iscrowd": 0})
            aid += 1
        for b, s, c in zip(*hip[i]):
            dets.append({"image_id": i, "category_id": int(c) + 1, "bbox": [float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])],
                         "score": float(s)})
    ev = evaluation.COCOevalBBox({"images": images, "annotations": anns, "categories": cats}, dets, impl="native")
    ev.evaluate()
    ev.accumulate()
    stats = ev.summarize(printer=None)
    n_o, n_h = sum(len(o[1]) for o in ora), sum(len(h[1]) for h in hip)
    # how many oracle detections have a HIP detection of the same class at IoU >= 0.9 / score within 0.02
    matched, close = 0, 0
    for (ob, os_, oc), (hb, hs, hc) in zip(ora, hip):
        if len(ob) == 0 or len(hb) == 0:
            continue
        x1 = np.maximum(ob[:, None, 0], hb[None, :, 0]); y1 = np.maximum(ob[:, None, 1], hb[None, :, 1])
        x2 = np.minimum(ob[:, None, 2], hb[None, :, 2]); y2 = np.minimum(ob[:, None, 3], hb[None, :, 3])
        inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
        iou = inter / ((ob[:, 2] - ob[:, 0]) * (ob[:, 3] - ob[:, 1]))[:, None].clip(1e-9)
        iou = inter / (((ob[:, 2] - ob[:, 0]) * (ob[:, 3] - ob[:, 1]))[:, None] + ((hb[:, 2] - hb[:, 0]) * (hb[:, 3] - hb[:, 1]))[None, :] - inter + 1e-12)
        iou = np.where(oc[:, None] == hc[None, :], iou, 0.0)
        j = iou.argmax(1)
        ok = iou.max(1) >= 0.9
        matched += int(ok.sum())
        close += int((np.abs(hs[j] - os_)[ok] < 0.02).sum())
    return {"images": n_images, "depth": depth, "oracle_detections": n_o, "hip_detections": n_h,
            "AP": round(float(stats[0]) * 100, 3), "AP50": round(float(stats[1]) * 100, 3), "AP75": round(float(stats[2]) * 100, 3),
            "oracle_dets_matched_iou90": round(matched / max(n_o, 1), 4), "matched_with_score_within_0.02": round(close / max(matched, 1), 4),
            "oracle_cpu_seconds": round(cpu_s, 1),
            "note": "oracle detections as ground truth, HIP detections scored with the native COCO evaluator; random-init weights"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--depth", type=int, default=101)
    ap.add_argument("--out", type=str, default="")
    a = ap.parse_args()
    res = run(a.images, a.depth)
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
