"""The Python boundary of SURVEY 8(b) on the GPU: Box2BoxTransform / DefaultAnchorGenerator against the fixtures the
reference's own classes produced (tests/golden/detector_ops.npz), build_model + DetectionCheckpointer, and the two
dataset drivers the reference names as harnesses - demo_mAP_FLIR.py and demo_LAMR_KAIST.py - end to end on synthetic
data written in the reference's directory layouts."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "detector_ops.npz")))


def test_box2box_transform_matches_reference(ops):
    import proben_amd
    b = torch.from_numpy(ops["b2b_boxes"]).cuda()
    for key, w in (("1", (1.0, 1.0, 1.0, 1.0)), ("3", (10.0, 10.0, 5.0, 5.0))):
        t = proben_amd.Box2BoxTransform(w)
        got = t.apply_deltas(torch.from_numpy(ops["b2b_d" + key]).cuda(), b).cpu().numpy()
        # identical expression order; the only difference is the device expf (<= 2 ulp)
        np.testing.assert_allclose(got, ops["b2b_out" + key], rtol=3e-6, atol=1e-4)
    # apply(get(src, dst), src) == dst; tensor math of get_deltas runs wherever the boxes are
    src = b[:16]
    dst = src + torch.tensor([3.0, -2.0, 9.0, 4.0], device="cuda")
    t = proben_amd.Box2BoxTransform((10.0, 10.0, 5.0, 5.0))
    torch.testing.assert_close(t.apply_deltas(t.get_deltas(src, dst), src), dst, rtol=1e-5, atol=1e-3)
    with pytest.raises(proben_amd._lib.HipLibraryError):
        t.apply_deltas(torch.zeros(2, 4), torch.zeros(2, 4))          # CPU tensors: no fallback


def test_default_anchor_generator_matches_reference(ops):
    import proben_amd
    from proben_amd.modeling import ANCHOR_GENERATOR_REGISTRY, ShapeSpec
    cfg = proben_amd.get_cfg()
    ag = ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, [ShapeSpec(stride=s) for s in (4, 8, 16, 32, 64)])
    assert ag.num_cell_anchors == [3] * 5 and ag.box_dim == 4
    feats = [torch.zeros(2, 1, int(h), int(w), device="cuda") for h, w in ops["anchors_grid"]]
    anchors = ag(feats)
    assert len(anchors) == 2 and len(anchors[0]) == 5
    for i, a in enumerate(anchors[1]):
        np.testing.assert_array_equal(a.tensor.cpu().numpy(), ops[f"anchors_l{i}"])     # bit-exact: sums of exact floats
    # tests/test_anchor_generator.py:25-40 of the reference (sizes 32, 64 x ratios .25, 1, 4 on a 1x2 grid, stride 4)
    cfg.MODEL.ANCHOR_GENERATOR.SIZES = [[32, 64]]
    cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.25, 1, 4]]
    ag = proben_amd.DefaultAnchorGenerator(cfg, [ShapeSpec(stride=4)])
    got = ag([torch.zeros(1, 3, 1, 2, device="cuda")])[0][0].tensor.cpu()
    want = torch.tensor([[-32.0, -8.0, 32.0, 8.0], [-16.0, -16.0, 16.0, 16.0], [-8.0, -32.0, 8.0, 32.0],
                         [-64.0, -16.0, 64.0, 16.0], [-32.0, -32.0, 32.0, 32.0], [-16.0, -64.0, 16.0, 64.0],
                         [-28.0, -8.0, 36.0, 8.0], [-12.0, -16.0, 20.0, 16.0], [-4.0, -32.0, 12.0, 32.0],
                         [-60.0, -16.0, 68.0, 16.0], [-28.0, -32.0, 36.0, 32.0], [-12.0, -64.0, 20.0, 64.0]])
    assert torch.allclose(got, want)


def test_build_model_checkpointer_and_predictor_attributes(tmp_path):
    """build_model(cfg) -> model on cuda without weights; DetectionCheckpointer(model).load(.pth) makes it equal to the
    predictor built from the same file; DefaultPredictor carries .cfg .model .metadata .transform_gen .input_format."""
    import proben_amd
    from proben_amd.synthetic import synthetic_images, synthetic_state_dict
    cfg = proben_amd.get_cfg()
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 3
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.5
    cfg.MODEL.ROI_BOX_HEAD.OUTPUT_LOGITS = True
    cfg.MODEL.ROI_HEADS.ENABLE_GAUSSIANNLLOSS = True
    sd = synthetic_state_dict(50, 3, 3, seed=5)
    path = str(tmp_path / "model.pth")
    torch.save({"model": sd}, path)
    model = proben_amd.build_model(cfg)
    proben_amd.DetectionCheckpointer(model).load(path)
    cfg.MODEL.WEIGHTS = path
    pred = proben_amd.DefaultPredictor(cfg)
    assert pred.input_format == "BGR" and pred.cfg is not cfg and pred.model is not None
    img = synthetic_images(1, 256, 320, seed=3)[0]
    assert pred.transform_gen.get_transform(img).apply_image(img).shape == (800, 1000, 3)
    want = pred(img)["instances"]
    x = torch.from_numpy(pred.transform_gen.get_transform(img).apply_image(img).astype("float32").transpose(2, 0, 1))
    got = model([{"image": x, "height": 256, "width": 320}])[0]["instances"]      # the reference's model call contract
    assert len(got) == len(want) and len(got) > 0
    torch.testing.assert_close(got.pred_boxes.tensor, want.pred_boxes.tensor, rtol=0, atol=0)
    assert torch.equal(got.pred_classes, want.pred_classes) and got.vars.shape == (len(got), 1)


def _write_flir(root, n, H, W):
    from PIL import Image
    from proben_amd.synthetic import synthetic_images
    (root / "thermal_8_bit").mkdir(parents=True)
    (root / "RGB").mkdir()
    th, rgb = synthetic_images(n, H, W, seed=31), synthetic_images(n, H + 40, W + 60, seed=32)
    images, anns = [], []
    for i in range(n):
        Image.fromarray(th[i]).save(root / "thermal_8_bit" / f"FLIR_{i:05d}.jpeg", quality=95)
        Image.fromarray(rgb[i]).save(root / "RGB" / f"FLIR_{i:05d}.jpg", quality=95)
        images.append({"id": i, "file_name": f"thermal_8_bit/FLIR_{i:05d}.jpeg", "height": H, "width": W})
        anns.append({"id": i + 1, "image_id": i, "category_id": 1 + i % 3, "bbox": [20, 30, 60, 80], "area": 4800, "iscrowd": 0})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"},
                                                                      {"id": 3, "name": "car"}]},
              open(root / "FLIR_thermal_RGBT_pairs_val.json", "w"))


def test_demo_map_flir_driver_on_the_gpu(tmp_path):
    """cli/demo_mAP_FLIR.py (counterpart of demo/FLIR/demo_mAP_FLIR.py:1-66): DefaultPredictor -> test loader ->
    inference_on_dataset -> FLIREvaluator, on a synthetic FLIR-layout dataset; detections used as their own ground
    truth must score AP50 = 100 (closes the loop predictor -> rows -> COCOeval on the device path)."""
    from proben_amd.cli import demo_mAP_FLIR
    root = tmp_path / "val"
    _write_flir(root, 5, 256, 320)
    res = demo_mAP_FLIR.main(["--dataset_path", str(root), "--fusion_method", "thermal_only", "--outfolder", str(tmp_path / "out"),
                              "--dataset_name", "flir_map_gpu_a"])
    assert "bbox" in res and set(("AP", "AP50", "AP75")) <= set(res["bbox"])
    # second pass: the detector's own output becomes the ground truth
    import proben_amd
    from proben_amd.cli.save_predictions import build_cfg
    from proben_amd.data import read_image
    from proben_amd.opt import config_parser
    args = config_parser(["--dataset_path", str(root), "--fusion_method", "thermal_only"])
    pred = proben_amd.DefaultPredictor(build_cfg(args))
    d = json.load(open(root / "FLIR_thermal_RGBT_pairs_val.json"))
    anns, aid = [], 1
    for im in d["images"]:
        inst = pred(read_image(str(root / im["file_name"]), "BGR"))["instances"].to("cpu")
        for b, c in zip(inst.pred_boxes.tensor.tolist(), inst.pred_classes.tolist()):
            anns.append({"id": aid, "image_id": im["id"], "category_id": c + 1, "bbox": [b[0], b[1], b[2] - b[0], b[3] - b[1]],
                         "area": (b[2] - b[0]) * (b[3] - b[1]), "iscrowd": 0})
            aid += 1
    assert aid > 10
    d["annotations"] = anns
    json.dump(d, open(root / "FLIR_thermal_RGBT_pairs_val.json", "w"))
    res = demo_mAP_FLIR.main(["--dataset_path", str(root), "--fusion_method", "thermal_only", "--outfolder", str(tmp_path / "out"),
                              "--dataset_name", "flir_map_gpu_b"])
    assert res["bbox"]["AP50"] == pytest.approx(100.0, abs=1e-6)


def test_kaist_driver_config5(tmp_path):
    """cli/demo_LAMR_KAIST.main (demo/KAIST/demo_LAMR_KAIST.py:85-145) on a synthetic KAIST tree: single detector and
    the two-detector binary-ProbEn mode (configs[4]); text rows parse back, the variance file has every frame, and
    scoring the detections against themselves gives a log-average miss rate of ~0."""
    from PIL import Image
    from proben_amd.cli import demo_LAMR_KAIST as K
    from proben_amd.synthetic import synthetic_images
    root = tmp_path / "KAIST"
    lines = []
    th, rgb = synthetic_images(4, 256, 320, seed=51), synthetic_images(4, 256, 320, seed=52)
    for i in range(4):
        d = root / "test" / "set06" / f"V00{i % 2}"
        (d / "lwir").mkdir(parents=True, exist_ok=True)
        (d / "visible").mkdir(parents=True, exist_ok=True)
        Image.fromarray(th[i]).save(d / "lwir" / f"I0{i:04d}.jpg", quality=95)
        Image.fromarray(rgb[i]).save(d / "visible" / f"I0{i:04d}.jpg", quality=95)
        lines.append(f"set06/V00{i % 2}/I0{i:04d}")
    split = tmp_path / "test-all-20.txt"
    split.write_text("\n".join(lines) + "\n")
    out = tmp_path / "out"
    r = K.main(["--dataset_path", str(root), "--split_file", str(split), "--fusion_method", "thermal_only", "--out_folder", str(out),
                "--batch", "3"])
    assert r["frames"] == 4 and r["rows"] > 0
    rows = K.read_kaist_rows(r["txt"], 4)
    var = np.load(out / "KAIST_thermal_only_variance.npz", allow_pickle=True)["vars"].item()
    assert sorted(var) == [1, 2, 3, 4] and [len(v) for v in var.values()] == [len(x) for x in rows]
    first = open(r["txt"]).readline().strip().split(",")
    assert first[0] == "1" and len(first) == 6 and all(float(v) >= 0 for v in first[1:])
    # KAIST_annotation.json as the reference passes it to evalKAIST.evaluation_script.evaluate (demo_LAMR_KAIST.py:145): COCO-style,
    # image ids 0-based; the detections themselves as ground truth.  The rules of the protocol are pinned case by case in
    # tests/test_kaist_eval_cpu.py; here: the driver hands the file it wrote and the annotation file to the evaluator and reports
    # what the evaluator returns (all / day / night, recall), and a set scored against itself has no false positive.
    from proben_amd.evalKAIST.evaluation_script import evaluate
    boxes = [(i, x, y, w, h) for i in range(4) for (x, y, w, h, s) in rows[i]]
    gt = {"images": [{"id": i, "im_name": lines[i]} for i in range(4)],
          "annotations": [{"id": k, "image_id": i, "category_id": 1, "bbox": [x, y, w, h], "height": h, "occlusion": 0, "ignore": 0}
                          for k, (i, x, y, w, h) in enumerate(boxes)],
          "categories": [{"id": 1, "name": "person"}]}
    ann = tmp_path / "KAIST_annotation.json"
    ann.write_text(json.dumps(gt))
    r2 = K.main(["--dataset_path", str(root), "--split_file", str(split), "--fusion_method", "thermal_only", "--out_folder", str(out),
                 "--annotation_json", str(ann)])
    ev = evaluate(str(ann), r2["txt"], "Multispectral")
    assert r2["MR_all"] == r2["log_average_miss_rate"] == ev["all"].summarize(0) and r2["MR_day"] == ev["day"].summarize(0)
    assert r2["MR_night"] == -1.0                                       # four frames: all of them in the day part of the split
    assert all(not e["dtIgnore"][0][~e["dtMatches"][0]].any() and e["dtMatches"].all() for e in ev["all"].evalImgs if e is not None)
    r3 = K.main(["--dataset_path", str(root), "--split_file", str(split), "--fusion_method", "probEn", "--out_folder", str(out),
                 "--annotation_json", str(ann), "--batch", "4"])
    assert r3["frames"] == 4 and r3["rows"] > 0 and (r3["log_average_miss_rate"] == -1.0 or 0.0 <= r3["log_average_miss_rate"] <= 1.0)
