"""Host-side logic on CPU: evaluator vs the reference's vendored COCOeval fixture, config, resize rule,
shards, containers, J1 schema, class whitelist, KAIST rows."""
import json
import os

import numpy as np
import pytest
import torch

import proben_amd  # noqa: F401
from proben_amd import data, evaluation, late_fusion
from proben_amd.cli import demo_LAMR_KAIST
from proben_amd.structures import Boxes, ImageList, Instances


@pytest.mark.parametrize("impl", ["native", "numpy"])
def test_cocoeval_matches_reference_vendored_evaluator(golden_dir, impl):
    """Both evaluators (the multithreaded C++ one in libproben_hip.so and its NumPy restatement) against the
    fixture produced by the reference's vendored COCOeval (tests/golden/gen_cocoeval.py)."""
    z = json.load(open(os.path.join(golden_dir, "cocoeval_case.json")))
    ev = evaluation.COCOevalBBox(z["gt"], z["dets"], impl=impl)
    ev.evaluate()
    ev.accumulate()
    stats = ev.summarize(printer=None)
    np.testing.assert_allclose(stats, z["stats"], rtol=0, atol=1e-12)
    prec = ev.eval["precision"]
    assert float(prec[prec > -1].sum()) == pytest.approx(z["precision_sum"], abs=1e-9)
    rec = ev.eval["recall"]
    assert float(rec[rec > -1].sum()) == pytest.approx(z["recall_sum"], abs=1e-9)
    for k, want in enumerate(z["per_class_ap"]):
        p = prec[:, :, k, 0, -1]
        p = p[p > -1]
        assert float(np.mean(p)) == pytest.approx(want, abs=1e-12)


def _random_eval_case(n_images, seed, crowd):
    rng = np.random.default_rng(seed)
    images = [{"id": 10 + 3 * i, "height": 512, "width": 640} for i in range(n_images)]
    anns, res = [], []
    for im in images:
        for _ in range(rng.integers(0, 12)):
            x, y = rng.uniform(0, 500), rng.uniform(0, 400)
            w, h = rng.uniform(4, 200, 2)
            c = int(rng.integers(1, 4))
            anns.append({"id": len(anns), "image_id": im["id"], "category_id": c, "bbox": [x, y, w, h], "area": w * h,
                         "iscrowd": int(crowd and rng.random() < .15)})   # ids start at 0: id 0 = "unmatched" quirk
            for _ in range(rng.integers(0, 3)):
                res.append({"image_id": im["id"], "category_id": c, "score": float(np.round(rng.uniform(.5, 1), 2)),
                            "bbox": [x + rng.normal(0, 4), y + rng.normal(0, 4), w * rng.uniform(.8, 1.2), h]})
        for _ in range(rng.integers(0, 8)):   # category 4 has no ground truth; 5 is not a dataset category
            x, y = rng.uniform(0, 500), rng.uniform(0, 400)
            w, h = rng.uniform(4, 200, 2)
            res.append({"image_id": im["id"], "category_id": int(rng.integers(1, 6)), "bbox": [x, y, w, h],
                        "score": float(np.round(rng.uniform(.5, 1), 2))})
    return {"images": images, "annotations": anns, "categories": [{"id": c} for c in (1, 2, 3, 4)]}, res


@pytest.mark.parametrize("n_images,seed,crowd,threads", [(1, 0, False, 1), (40, 1, False, 3), (150, 2, True, 0)])
def test_native_cocoeval_bit_identical_to_numpy(n_images, seed, crowd, threads):
    """Tied scores (rounded to 0.01), crowd boxes, >100 detections in one cell, empty cells, a category without
    ground truth, annotation id 0: precision / recall tables must be equal bit for bit."""
    gt, res = _random_eval_case(n_images, seed, crowd)
    if n_images == 40:  # one (image, category) cell above maxDets = 100
        res += [{"image_id": 10, "category_id": 1, "bbox": [5.0 * k, 3.0 * k, 40, 40], "score": 0.5 + (k % 50) / 100}
                for k in range(130)]
    a = evaluation.COCOevalBBox(gt, res, impl="numpy")
    b = evaluation.COCOevalBBox(gt, res, impl="native", num_threads=threads)
    for e in (a, b):
        e.evaluate()
        e.accumulate()
        e.summarize(printer=None)
    assert np.array_equal(a.eval["precision"], b.eval["precision"])
    assert np.array_equal(a.eval["recall"], b.eval["recall"])
    assert np.array_equal(a.stats, b.stats)


def test_native_cocoeval_orders_nan_scores_last_like_numpy():
    """A NaN score is what the reference's own bayesian_fusion_multiclass (demo_probEn.py:32-42) makes of a cluster member whose class
    probabilities sum to 1 + 1 ulp (background = 1 - sum < 0 -> log = NaN).  cocoeval.py sorts with np.argsort(-score, mergesort): NaN
    last, stable.  The native evaluator used a bare `a > b` until round 6 - not an ordering once a NaN is in the list; the fused-mAP
    fixture found it (15 NaN rows of 7 885 moved AP by 6 points).  NaN rows sprinkled over every cell: tables equal bit for bit, and the
    same tables as with the NaN rows moved to the END of the file (they sort last wherever they stand)."""
    gt, res = _random_eval_case(60, 5, True)
    rng = np.random.default_rng(3)
    nan_rows = [dict(r, score=float("nan")) for r in rng.choice(res, 25, replace=False)]
    mixed = list(res)
    for r in nan_rows:
        mixed.insert(int(rng.integers(0, len(mixed))), r)
    out = []
    for rows in (mixed, res + nan_rows):
        a = evaluation.COCOevalBBox(gt, rows, impl="numpy")
        b = evaluation.COCOevalBBox(gt, rows, impl="native", num_threads=2)
        for e in (a, b):
            e.evaluate()
            e.accumulate()
            e.summarize(printer=None)
        assert np.array_equal(a.eval["precision"], b.eval["precision"]) and np.array_equal(a.eval["recall"], b.eval["recall"])
        assert np.array_equal(a.stats, b.stats)
        out.append(b.stats)
    assert np.array_equal(out[0], out[1])


def test_native_cocoeval_empty_and_bad_rows():
    gt = {"images": [{"id": 1}], "annotations": [], "categories": [{"id": 1}]}
    e = evaluation.COCOevalBBox(gt, [])
    e.evaluate()
    e.accumulate()
    assert (e.eval["precision"] == -1).all() and (e.summarize(printer=None) == -1).all()
    with pytest.raises(AssertionError):   # the reference's loadRes assertion
        evaluation.COCOevalBBox(gt, [{"image_id": 7, "category_id": 1, "bbox": [0, 0, 1, 1], "score": .5}])


def test_flir_evaluator_end_to_end(tmp_path, golden_dir):
    z = json.load(open(os.path.join(golden_dir, "cocoeval_case.json")))
    gt_path = tmp_path / "FLIR_thermal_RGBT_pairs_val.json"
    json.dump(z["gt"], open(gt_path, "w"))
    data.register_coco_instances("flir_test", {}, str(gt_path), str(tmp_path))
    data.DatasetCatalog.get("flir_test")
    ev = evaluation.FLIREvaluator("flir_test", proben_amd.get_cfg(), False, output_dir=str(tmp_path))
    # detections arrive as Instances in CONTIGUOUS class ids (0..2); the evaluator maps them back to dataset ids
    by_img = {}
    for d in z["dets"]:
        by_img.setdefault(d["image_id"], []).append(d)
    for iid, ds in by_img.items():
        inst = Instances((512, 640))
        b = torch.tensor([d["bbox"] for d in ds], dtype=torch.float64)
        b[:, 2:] += b[:, :2]
        inst.pred_boxes = Boxes(b.float())
        inst.scores = torch.tensor([d["score"] for d in ds])
        inst.pred_classes = torch.tensor([d["category_id"] - 1 for d in ds])
        ev.process([{"image_id": iid}], [{"instances": inst}])
    res = ev.evaluate()["bbox"]
    assert res["AP50"] == pytest.approx(z["stats"][1] * 100, abs=0.05)   # boxes went through float32
    assert set(res) >= {"AP", "AP50", "AP75", "AP-person", "AP-bicycle", "AP-car"}


def test_instances_to_coco_json_class_whitelist():
    inst = Instances((512, 640))
    inst.pred_boxes = Boxes(torch.tensor([[0, 0, 10, 20.0]] * 6))
    inst.scores = torch.linspace(0.9, 0.4, 6)
    inst.pred_classes = torch.tensor([0, 3, 5, 7, 16, 2])   # 3 (ProbEn "background") is dropped, 5/7 -> 2
    out = evaluation.instances_to_coco_json(inst, 7)
    assert [o["category_id"] for o in out] == [0, 2, 2, 16, 2]
    assert out[0]["bbox"] == [0.0, 0.0, 10.0, 20.0] and out[0]["image_id"] == 7


def test_config_merges_reference_style_yaml(tmp_path):
    (tmp_path / "base.yaml").write_text("MODEL:\n  RESNETS:\n    DEPTH: 50\n  RPN:\n    BBOX_REG_WEIGHTS: (1.0, 1.0, 1.0, 1.0)\n")
    (tmp_path / "child.yaml").write_text('_BASE_: "base.yaml"\nMODEL:\n  RESNETS:\n    DEPTH: 101\n  WEIGHTS: "x.pth"\n')
    cfg = proben_amd.get_cfg()
    cfg.merge_from_file(str(tmp_path / "child.yaml"))
    assert cfg.MODEL.RESNETS.DEPTH == 101 and cfg.MODEL.WEIGHTS == "x.pth"
    assert cfg.MODEL.RPN.BBOX_REG_WEIGHTS == (1.0, 1.0, 1.0, 1.0)
    cfg.MODEL.ROI_BOX_HEAD.DROP_OUT = True  # scripts add new keys on the fly (demo_FLIR_save_predictions.py:53)
    cfg.merge_from_list(["MODEL.ROI_HEADS.NUM_CLASSES", 3])
    assert cfg.clone().MODEL.ROI_HEADS.NUM_CLASSES == 3
    from proben_amd.predictor import detector_config_from_cfg
    assert detector_config_from_cfg(cfg).num_classes == 3


def test_predictor_refuses_cpu_device():
    cfg = proben_amd.get_cfg()
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.WEIGHTS = "synthetic://1"
    with pytest.raises(proben_amd._lib.HipLibraryError):
        proben_amd.DefaultPredictor(cfg)


def test_resize_shortest_edge_rule():
    assert data.resize_shortest_edge_shape(512, 640) == (800, 1000)       # FLIR / KAIST
    assert data.resize_shortest_edge_shape(480, 640) == (800, 1067)
    assert data.resize_shortest_edge_shape(100, 1000) == (133, 1333)      # capped by MAX_SIZE_TEST
    assert data.resize_shortest_edge_shape(640, 512) == (1000, 800)


def test_inference_sampler_shards():
    for n, w in [(10, 4), (8, 8), (3, 8), (0, 2), (1366, 8)]:
        parts = [list(data.InferenceSampler(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n))                           # contiguous, ordered, complete
        shard = (n - 1) // w + 1 if n else 0
        assert all(len(p) <= shard for p in parts)


def test_boxes_and_instances_contract():
    b = Boxes(torch.tensor([[-5.0, 2.0, 700.0, 600.0], [1.0, 1.0, 1.0, 5.0]]))
    b.clip((512, 640))
    assert b.tensor.tolist() == [[0.0, 2.0, 640.0, 512.0], [1.0, 1.0, 1.0, 5.0]]
    assert b.nonempty().tolist() == [True, False] and b.area().tolist() == [640.0 * 510.0, 0.0]
    b.scale(0.5, 2.0)
    assert b.tensor[0].tolist() == [0.0, 4.0, 320.0, 1024.0]
    assert Boxes([]).tensor.shape == (0, 4) and len(Boxes.cat([b, b])) == 4
    inst = Instances((10, 20))
    inst.pred_boxes = b
    inst.scores = torch.tensor([0.9, 0.1])
    with pytest.raises(AssertionError):
        inst.pred_classes = torch.tensor([1])                             # length check (instances.py)
    # tests/test_instances.py:9-21: int indexing keeps a length-1 Instances, out of range raises
    assert len(inst[1]) == 1 and len(inst[-1]) == 1
    with pytest.raises(IndexError):
        inst[2]
    assert len(Instances.cat([inst, inst])) == 4 and inst.has("scores") and not inst.has("vars")
    with pytest.raises(AttributeError):
        inst.nothing
    il = ImageList.from_tensors([torch.ones(3, 5, 7), torch.ones(3, 6, 4)], 32)
    assert il.tensor.shape == (2, 3, 32, 32) and il.image_sizes == [(5, 7), (6, 4)] and float(il.tensor[0, 0, 5:].sum()) == 0


def test_j1_prediction_schema(tmp_path):
    inst = Instances((512, 640))
    inst.pred_boxes = Boxes(torch.tensor([[1.0, 2, 30, 40], [5.0, 6, 70, 80], [9.0, 9, 20, 20]]))
    inst.scores = torch.tensor([0.9, 0.8, 0.7])
    inst.pred_classes = torch.tensor([0, 5, 2])      # class 5 (> 2) is dropped like :148-155
    inst.class_logits = torch.rand(3, 4)
    inst.prob_score = torch.rand(3, 3)
    inst.vars = torch.rand(3, 1)
    pred = late_fusion.predictions_to_j1(["a.jpeg"], [11], [inst])
    assert list(pred) == late_fusion.J1_KEYS
    assert pred["classes"] == [[0, 2]] and len(pred["boxes"][0]) == 2 and len(pred["vars"][0][0]) == 1
    assert len(pred["class_logits"][0][0]) == 4 and len(pred["probs"][0][0]) == 3 and pred["image_id"] == [11]
    p = tmp_path / "val_thermal_only_predictions.json"
    late_fusion.write_j1(str(p), pred)
    assert late_fusion.read_j1(str(p)) == json.loads(json.dumps(pred))

def test_j1_writer_reproduces_the_references_file_byte_for_byte(tmp_path, golden_dir):
    """predictions_to_j1 + write_j1 against tests/golden/j1_case.json = the prediction file the REFERENCE's writer statements
    (demo/FLIR/demo_FLIR_save_predictions.py:133-176, executed on stub predictor outputs by tests/golden/gen_j1.py) produce for the
    same detections: key order, indent, the float repr of float32 values widened by `.tolist()`, classes > 2 dropped, empty images."""
    z = json.load(open(os.path.join(golden_dir, "j1_case.json")))
    K = z["K"]
    f32 = lambda u, *shape: torch.from_numpy(np.asarray(u, dtype=np.uint32).view(np.float32).reshape(*shape).copy())  # noqa: E731
    insts = []
    for fr in z["frames"]:
        n = fr["n"]
        inst = Instances((512, 640))
        inst.pred_boxes = Boxes(f32(fr["boxes_u32"], n, 4))
        inst.scores = f32(fr["scores_u32"], n)
        inst.pred_classes = torch.tensor(fr["classes"], dtype=torch.int64)
        inst.class_logits = f32(fr["logits_u32"], n, K + 1)
        inst.prob_score = f32(fr["probs_u32"], n, K)
        inst.vars = f32(fr["vars_u32"], n, 1)
        insts.append(inst)
    pred = late_fusion.predictions_to_j1(z["files_names"], z["image_ids"], insts)
    p = tmp_path / "val_thermal_only_predictions.json"
    late_fusion.write_j1(str(p), pred)
    assert p.read_text() == z["text"]
    assert any(c > 2 for fr in z["frames"] for c in fr["classes"]) and any(fr["n"] == 0 for fr in z["frames"])   # the fixture covers both


def test_model_zoo_pickle_reader_and_rgb_only_cfg(tmp_path, monkeypatch):
    """checkpoint/detection_checkpoint.py:27-45: a detectron2 model-zoo `.pkl` ({"model": ndarrays, "__author__"}) loads into the same
    state dict a `.pth` gives; a Caffe2 / Detectron1 pickle is refused (its key-matching heuristics are not on the inference path).
    demo_FLIR_save_predictions.py:59-61: `rgb_only` is the 80-class COCO detector with the zoo weights."""
    import pickle
    from proben_amd.cli.save_predictions import COCO_ZOO_WEIGHTS, build_cfg
    from proben_amd.opt import config_parser
    from proben_amd.synthetic import synthetic_state_dict
    from proben_amd.weights import load_state_dict_file
    sd = synthetic_state_dict(50, 80, 3, seed=3)
    sd = {k: v for k, v in sd.items() if "var_pred" not in k}       # the zoo model has no variance head
    zoo = tmp_path / "model_final_f6e8b1.pkl"
    with open(zoo, "wb") as f:
        pickle.dump({"model": {k: v.numpy() for k, v in sd.items()}, "__author__": "Detectron2 Model Zoo"}, f, protocol=2)
    got = load_state_dict_file(str(zoo))
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert got["roi_heads.box_predictor.cls_score.weight"].shape[0] == 81 and got["roi_heads.box_predictor.bbox_pred.weight"].shape[0] == 320
    pth = tmp_path / "m.pth"
    torch.save({"model": sd}, pth)
    again = load_state_dict_file(str(pth))
    assert all(torch.equal(again[k], sd[k]) for k in sd)
    c2 = tmp_path / "R-101.pkl"
    with open(c2, "wb") as f:
        pickle.dump({"blobs": {"conv1_w": np.zeros((64, 3, 7, 7), np.float32)}}, f, protocol=2)
    with pytest.raises(ValueError, match="model-zoo"):
        load_state_dict_file(str(c2))
    # cfg of the variant: 80 classes; the zoo weights when the reference's relative path exists, else --model_path / synthetic
    cfg = build_cfg(config_parser(["--dataset_path", "x", "--fusion_method", "rgb_only"]))
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 80 and cfg.INPUT.FORMAT == "BGR" and list(cfg.MODEL.PIXEL_MEAN) == [103.53, 116.28, 123.675]
    monkeypatch.chdir(tmp_path)
    os.makedirs(os.path.dirname(COCO_ZOO_WEIGHTS))
    os.replace(zoo, COCO_ZOO_WEIGHTS)
    cfg = build_cfg(config_parser(["--dataset_path", "x", "--fusion_method", "rgb_only", "--model_path", "ignored.pth"]))
    assert cfg.MODEL.WEIGHTS == COCO_ZOO_WEIGHTS
    assert build_cfg(config_parser(["--dataset_path", "x", "--fusion_method", "thermal_only"])).MODEL.ROI_HEADS.NUM_CLASSES == 3


def test_kaist_rows_byte_exact_vs_reference_writer(golden_dir):
    """K1: text rows and variance file of the KAIST driver against tests/golden/kaist_rows.json, produced by executing
    the reference's own writer statements (demo/KAIST/demo_LAMR_KAIST.py:128-142) - byte for byte."""
    g = json.load(open(os.path.join(golden_dir, "kaist_rows.json")))
    insts = []
    for fr in g["frames"]:
        n = fr["n"]
        f32 = lambda k, shape: np.asarray(fr[k], dtype=np.uint32).view(np.float32).reshape(shape).copy()  # noqa: E731
        inst = Instances((512, 640))
        inst.pred_boxes = Boxes(torch.from_numpy(f32("boxes_u32", (n, 4))))
        inst.scores = torch.from_numpy(f32("scores_u32", (n,)))
        inst.vars = torch.from_numpy(f32("vars_u32", (n, 1)))
        insts.append(inst)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        n = demo_LAMR_KAIST.write_kaist(os.path.join(d, "r.txt"), os.path.join(d, "v.npz"), insts)
        text = open(os.path.join(d, "r.txt")).read()
        var = np.load(os.path.join(d, "v.npz"), allow_pickle=True)["vars"].item()
        back = demo_LAMR_KAIST.read_kaist_rows(os.path.join(d, "r.txt"), len(insts))
    assert text == g["text"] and n == text.count("\n")
    assert sorted(var) == g["var_keys"] and var[1].shape == (3, 1)            # 1-based keys, every frame present
    assert [len(b) for b in back] == [fr["n"] for fr in g["frames"]]
    assert demo_LAMR_KAIST.kaist_rows(0, insts[0])[0] == "1,10.0,20.0,20.0,40.0,1.0"


def test_registry_and_build_model_contract():
    """fvcore-style Registry, the reference's NAME strings resolve, unknown names fail with the registry's
    KeyError, MODEL.DEVICE=cpu is refused loudly (no CPU fallback in the product)."""
    import proben_amd
    from proben_amd import modeling as M
    r = M.Registry("THINGS")

    @r.register()
    class Foo:
        pass

    def bar():
        return 1
    r.register(bar)
    assert r.get("Foo") is Foo and r.get("bar") is bar and "Foo" in r
    with pytest.raises(KeyError, match="No object named 'nope' found in 'THINGS' registry!"):
        r.get("nope")
    with pytest.raises(AssertionError):
        r.register(bar)
    cfg = proben_amd.get_cfg()
    for reg, name in ((M.META_ARCH_REGISTRY, cfg.MODEL.META_ARCHITECTURE), (M.BACKBONE_REGISTRY, cfg.MODEL.BACKBONE.NAME),
                      (M.PROPOSAL_GENERATOR_REGISTRY, cfg.MODEL.PROPOSAL_GENERATOR.NAME), (M.RPN_HEAD_REGISTRY, cfg.MODEL.RPN.HEAD_NAME),
                      (M.ANCHOR_GENERATOR_REGISTRY, cfg.MODEL.ANCHOR_GENERATOR.NAME), (M.ROI_HEADS_REGISTRY, cfg.MODEL.ROI_HEADS.NAME),
                      (M.ROI_BOX_HEAD_REGISTRY, cfg.MODEL.ROI_BOX_HEAD.NAME)):
        assert callable(reg.get(name))
    bad = cfg.clone()
    bad.MODEL.META_ARCHITECTURE = "RetinaNet"                    # other meta-architectures are out of scope
    with pytest.raises(KeyError):
        M.build_model(bad)
    bad = cfg.clone()
    bad.MODEL.ROI_HEADS.NAME = "CascadeROIHeads"
    bad.MODEL.DEVICE = "cuda"
    with pytest.raises(KeyError, match="ROI_HEADS"):
        M.build_model(bad)
    cpu = cfg.clone()
    cpu.MODEL.DEVICE = "cpu"
    with pytest.raises(proben_amd._lib.HipLibraryError):
        M.build_model(cpu)
    assert proben_amd.build_model is M.build_model and proben_amd.Box2BoxTransform is M.Box2BoxTransform


def test_transform_gen_and_host_resize_rules():
    """DefaultPredictor.transform_gen (engine/defaults.py:170): ResizeShortestEdge size rule incl. MIN_SIZE_TEST = 0
    (NoOpTransform), Pillow path for 3 channels; the OpenCV-rule resizes are 2 x 2 taps without antialiasing."""
    from proben_amd import modeling as M
    from proben_amd.data import cv2_linear_resize_f, cv2_linear_resize_u8, resize_shortest_edge_shape
    assert resize_shortest_edge_shape(512, 640, 800, 1333) == (800, 1000)
    assert resize_shortest_edge_shape(480, 1600, 800, 1333) == (400, 1333)
    assert resize_shortest_edge_shape(512, 640, 0, 1333) == (512, 640)
    tg = M.ResizeShortestEdge([800, 800], 1333)
    img = np.random.default_rng(0).integers(0, 256, (64, 80, 3), dtype=np.uint8)
    t = tg.get_transform(img)
    out = t.apply_image(img)
    from PIL import Image
    assert out.shape == (800, 1000, 3)
    assert np.array_equal(out, np.asarray(Image.fromarray(img).resize((1000, 800), Image.BILINEAR)))
    np.testing.assert_allclose(t.apply_coords(np.array([[8.0, 4.0]])), [[100.0, 50.0]])
    # OpenCV rule: exact 2x down-scale of a uint8 image = rounded mean of each 2 x 2 block (fixed-point, half up)
    a = np.random.default_rng(1).integers(0, 256, (8, 12, 3), dtype=np.uint8)
    half = cv2_linear_resize_u8(a, 4, 6)
    want = (a.reshape(4, 2, 6, 2, 3).astype(np.int64).sum(axis=(1, 3)) + 2) >> 2
    assert np.abs(half.astype(np.int64) - want).max() <= 1 and half.dtype == np.uint8
    assert np.array_equal(cv2_linear_resize_u8(a, 8, 12), a)
    # identity on a constant image, linear ramp stays linear in the interior (float rule)
    ramp = np.tile(np.arange(16, dtype=np.float64)[None, :, None], (4, 1, 4))
    up = cv2_linear_resize_f(ramp, 4, 32)
    np.testing.assert_allclose(up[0, 1:-1, 0], (np.arange(1, 31) + 0.5) * 0.5 - 0.5, rtol=0, atol=1e-6)
    assert t.apply_image(img).dtype == np.uint8 and M.ResizeTransform(4, 16, 4, 32).apply_image(ramp).shape == (4, 32, 4)


def test_bench_board_power_sampler_is_best_effort(tmp_path, monkeypatch):
    """bench.py's `power` object: a helper process polls rocm-smi during the timed region.  With a rocm-smi that answers, samples
    inside the window are summarised (median board power, cap, shader clock, joules per unit) and the helper's process group is
    gone afterwards; without rocm-smi the bench line only says so - the measurement never depends on it."""
    import importlib.util
    import stat
    import subprocess
    import time
    spec = importlib.util.spec_from_file_location("bench_for_power_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fake = tmp_path / "rocm-smi"
    fake.write_text('#!/bin/bash\nif [[ "$*" == *showmaxpower* ]]; then echo \'{"card0": {"Max Graphics Package Power (W)": "1400.0"}}\'; '
                    'else echo "WARNING: not json"; echo \'{"card0": {"Current Socket Graphics Package Power (W)": "1377.0", '
                    '"sclk clock speed:": "(1916Mhz)", "sclk clock level:": "1"}}\'; sleep 0.05; fi\n')
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    p = bench.BoardPower(0)
    p.start()
    assert p.proc is not None and p.cap == 1400.0
    time.sleep(0.2)
    w0 = time.time()
    time.sleep(0.8)
    w1 = time.time()
    r = p.result(w0, w1, units=64)
    assert r["available"] and r["board_w_median"] == 1377.0 and r["cap_w"] == 1400.0 and r["sclk_mhz_median"] == 1916.0 and r["samples"] >= 2
    assert abs(r["joules_per_unit"] - 1377.0 * (w1 - w0) / 64) < 1e-2
    p.proc.wait(timeout=5)                               # the helper loop (its own session) was terminated
    for _ in range(40):                                  # grandchildren (the fake tool's `sleep`) die with the group, a moment later
        if subprocess.run(["pgrep", "-g", str(p.proc.pid)], capture_output=True).returncode != 0:
            break
        time.sleep(0.05)
    else:
        raise AssertionError("the power helper's process group is still alive")
    assert not os.path.exists(p.path)
    # no rocm-smi at all
    monkeypatch.setenv("PATH", str(tmp_path / "empty"))
    q = bench.BoardPower(0)
    q.start()
    assert q.proc is None and q.result(w0, w1, 1) == {"available": False, "cap_w": None}
