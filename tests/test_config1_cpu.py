"""BASELINE config 1 (the reference's own CPU-runnable case): thermal-only R50-FPN, 8 images, MODEL.DEVICE=cpu
through the demo_mAP_FLIR plumbing (dataset registration -> detector -> FLIREvaluator -> AP table).
On CPU the detector is the ORACLE (the product has no CPU path); this test pins the plumbing around it."""
import json

import numpy as np
import torch

import proben_amd  # noqa: F401
from oracle import detector as D
from proben_amd import data, evaluation
from proben_amd.structures import Boxes, Instances
from proben_amd.synthetic import synthetic_images, synthetic_state_dict


def test_config1_plumbing_cpu(tmp_path):
    rng = np.random.default_rng(0)
    n_img, H, W = 8, 256, 320  # reduced frames keep the CPU suite short; the path is size-agnostic
    images = [{"id": 10 + i, "file_name": f"thermal_8_bit/f{i}.jpeg", "height": H, "width": W} for i in range(n_img)]
    cats = [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"}, {"id": 3, "name": "car"}]
    anns, aid = [], 1
    for im in images:
        for _ in range(int(rng.integers(2, 7))):
            w, h = rng.uniform(10, 120), rng.uniform(10, 100)
            x, y = rng.uniform(0, W - w), rng.uniform(0, H - h)
            anns.append({"id": aid, "image_id": im["id"], "category_id": int(rng.integers(1, 4)), "bbox": [x, y, w, h],
                         "area": w * h, "iscrowd": 0})
            aid += 1
    gt = tmp_path / "FLIR_thermal_RGBT_pairs_val.json"
    json.dump({"images": images, "annotations": anns, "categories": cats}, open(gt, "w"))
    data.register_coco_instances("flir_cfg1", {}, str(gt), str(tmp_path))
    dicts = data.DatasetCatalog.get("flir_cfg1")
    assert len(dicts) == n_img and data.MetadataCatalog.get("flir_cfg1").thing_classes == ["person", "bicycle", "car"]
    sd = synthetic_state_dict(50, 3, 3, seed=1)
    spec = D.DetectorSpec(depth=50)
    frames = synthetic_images(n_img, H, W, seed=4)
    torch.set_num_threads(8)

    def model(inputs):  # the reference's batch-1 contract: list[dict] -> list[{"instances": Instances}]
        outs = []
        for x in inputs:
            newh, neww = data.resize_shortest_edge_shape(x["height"], x["width"], 400, 667)  # MIN/MAX_SIZE_TEST halved
            im = torch.from_numpy(x["image_np"]).permute(2, 0, 1).float()[None]
            im = torch.nn.functional.interpolate(im, size=(newh, neww), mode="bilinear", align_corners=False)[0]
            o = D.forward([im], sd, spec, out_sizes=[(x["height"], x["width"])])[0]
            inst = Instances((x["height"], x["width"]))
            inst.pred_boxes = Boxes(o["boxes"])
            inst.scores = o["scores"]
            inst.pred_classes = o["classes"]
            outs.append({"instances": inst})
        return outs
    loader = data.build_detection_test_loader(
        dicts, lambda d: {"image_np": frames[d["image_id"] - 10], "height": d["height"], "width": d["width"], "image_id": d["image_id"]})
    assert len(loader) == n_img and all(len(b) == 1 for b in loader)
    ev = evaluation.FLIREvaluator("flir_cfg1", proben_amd.get_cfg(), False, output_dir=str(tmp_path))
    res = evaluation.inference_on_dataset(model, loader, ev)
    assert set(res["bbox"]) >= {"AP", "AP50", "AP75", "APs", "APm", "APl", "AP-person", "AP-bicycle", "AP-car"}
    assert -1.0 <= res["bbox"]["AP50"] <= 100.0
    assert (tmp_path / "coco_instances_results.json").exists()
