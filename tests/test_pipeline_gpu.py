"""Predictor, late-fusion driver (P5 case split) and the device-to-device stage fusion on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _j1_from_synth(per_image, d):
    out = {k: [] for k in ["image", "boxes", "scores", "classes", "image_id", "class_logits", "probs", "vars"]}
    for i, infos in enumerate(per_image):
        x = infos[d]
        out["image"].append(f"im{i}.jpeg")
        out["boxes"].append(np.asarray(x["bbox"]).tolist())
        out["scores"].append(np.asarray(x["score"]).tolist())
        out["classes"].append(np.asarray(x["class"]).tolist())
        out["image_id"].append(1000 + i)
        out["class_logits"].append(np.asarray(x["prob"]).tolist())
        out["probs"].append(np.asarray(x["prob"]).tolist())
        out["vars"].append(np.asarray(x["vars"]).tolist())
    return out


@pytest.mark.parametrize("kdet", [2, 3])
@pytest.mark.parametrize("method", [("probEn", "v-avg"), ("avg", "s-avg"), ("max", "argmax")])
def test_late_fusion_case_split_matches_oracle(kdet, method):
    """demo_probEn.py:237-267: 0 detectors -> skip, 1 -> passthrough, 2/3 -> fusion of the non-empty lists."""
    import proben_amd  # noqa: F401
    from oracle import proben as O
    from proben_amd import late_fusion as LF
    from proben_amd.synthetic import synth_detections
    per_image = synth_detections(24, seed=40 + kdet, kdet=kdet, nmax=30)
    empty = {"bbox": np.zeros((0, 4)), "score": np.zeros(0), "class": np.zeros(0, int), "prob": np.zeros((0, 3)), "vars": np.zeros((0, 1))}
    per_image[0] = [empty] * kdet                                       # nobody fired
    per_image[1] = [per_image[1][0]] + [empty] * (kdet - 1)              # only detector 1
    per_image[2] = [empty] * (kdet - 1) + [per_image[2][-1]]             # only the last detector
    if kdet == 3:
        per_image[3] = [empty, per_image[3][1], per_image[3][2]]        # two of three
    dets = [_j1_from_synth(per_image, d) for d in range(kdet)]
    got = LF.late_fusion(dets, list(method))
    for i, infos in enumerate(per_image):
        live = [x for x in infos if len(x["score"])]
        if not live:
            assert got[i] is None
            continue
        b, s, c = got[i]
        if len(live) == 1:
            np.testing.assert_array_equal(np.asarray(b), live[0]["bbox"])
            np.testing.assert_array_equal(s.numpy(), live[0]["score"].astype(np.float32))
            continue
        wb, ws, wc = O.fusion(list(method), *live)
        assert len(b) == len(wb), (i, method)
        np.testing.assert_array_equal(c.numpy(), wc)
        np.testing.assert_allclose(s.numpy(), ws, rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(np.asarray(b, dtype=np.float64), np.asarray(wb, dtype=np.float64), rtol=1e-6, atol=1e-6, equal_nan=True)


def test_predictor_and_device_stage_fusion():
    """DefaultPredictor contract + fuse_detections (device-to-device) == JSON-style host route."""
    import proben_amd
    from proben_amd import fusion as F
    from proben_amd import late_fusion as LF
    from proben_amd.synthetic import synthetic_images
    preds = []
    for seed in (1, 2):
        cfg = proben_amd.get_cfg()
        cfg.MODEL.RESNETS.DEPTH = 50
        cfg.MODEL.ROI_HEADS.NUM_CLASSES = 3
        cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.5
        cfg.MODEL.ROI_BOX_HEAD.OUTPUT_LOGITS = True
        cfg.MODEL.ROI_HEADS.ENABLE_GAUSSIANNLLOSS = True
        cfg.MODEL.WEIGHTS = f"synthetic://{seed}"
        preds.append(proben_amd.DefaultPredictor(cfg))
    frames = synthetic_images(3, height=256, width=320, seed=5)
    one = preds[0](frames[0])["instances"]
    assert one.image_size == (256, 320) and one.has("vars") and one.has("prob_score") and one.has("class_logits")
    dets, j1 = [], []
    for p in preds:
        d = p.model.forward_batch([torch.from_numpy(f).cuda() for f in frames], out_sizes=[(256, 320)] * 3, resize_to=(800, 1000))
        dets.append(d)
        insts = [o["instances"] for o in p.model.to_instances(d)]
        j1.append(LF.predictions_to_j1([f"im{i}.jpeg" for i in range(3)], list(range(3)), insts))
    batch0 = [o["instances"] for o in preds[0].predict_batch(list(frames))]
    assert torch.equal(batch0[0].pred_boxes.tensor, one.pred_boxes.tensor)     # batch == single image
    fused = F.fuse_detections(dets, "probEn", "v-avg")
    host = LF.late_fusion(j1, ["probEn", "v-avg"])
    cnt, off = fused["counts"].cpu().numpy(), fused["offsets"].cpu().numpy()
    for i in range(3):
        if host[i] is None:
            assert cnt[i] == 0
            continue
        b, s, c = host[i]
        sl = slice(off[i], off[i] + cnt[i])
        assert cnt[i] == len(s)
        np.testing.assert_array_equal(fused["classes"][sl].cpu().numpy(), c.numpy())
        np.testing.assert_allclose(fused["scores"][sl].cpu().numpy(), s.numpy(), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(fused["boxes"][sl].cpu().numpy(), np.asarray(b), rtol=1e-9, atol=1e-9, equal_nan=True)
    # (max, argmax) = nms_1 route (demo_probEn.py:44-71,189-196); image 1 is made a one-detector image: passthrough
    dets[1]["counts"][1] = 0
    j1[1] = LF.predictions_to_j1([f"im{i}.jpeg" for i in range(3)], list(range(3)),
                                 [o["instances"] for o in preds[1].model.to_instances(dets[1])])
    fused = F.fuse_detections(dets, "max", "argmax")
    host = LF.late_fusion(j1, ["max", "argmax"])
    assert fused["nms_route"]
    kc, off = fused["counts"].cpu().numpy(), fused["offsets"].cpu().numpy()
    for i in range(3):
        b, sc, c = host[i]
        rows = slice(off[i], off[i] + kc[i])     # every route: image i's fused rows are [offsets[i], +counts[i])
        assert kc[i] == len(sc), i
        np.testing.assert_allclose(fused["scores"][rows].cpu().numpy(), sc.numpy(), rtol=1e-6)
        np.testing.assert_array_equal(fused["classes"][rows].cpu().numpy(), np.asarray(c, dtype=np.float32))
        np.testing.assert_allclose(fused["boxes"][rows].cpu().numpy(), np.asarray(b, dtype=np.float64), rtol=1e-6, atol=1e-4)
    assert kc[1] == int(dets[0]["counts"][1])   # untouched list of the only detector that fired
    # the evaluation-row form (what crosses ranks) needs no special case for this route
    rows = LF.fused_rows_device(fused, [10, 11, 12]).cpu().numpy()
    want = []
    for i in range(3):
        b, sc, c = host[i]
        for bb, ss, cc in zip(np.asarray(b, dtype=np.float64), sc.numpy(), np.asarray(c)):
            if int(cc) in (0, 1, 2):
                want.append([10 + i, bb[0], bb[1], bb[2] - bb[0], bb[3] - bb[1], ss, cc])
    np.testing.assert_allclose(rows, np.asarray(want, dtype=np.float64).reshape(-1, 7), rtol=1e-6, atol=1e-4)


def test_frame_pair_pipeline_concurrent_equals_serial():
    """Detectors on separate HIP streams (the bench configuration), and the staggered throughput mode with several
    batches in flight, give bit-identical results to the serial run."""
    import proben_amd  # noqa: F401
    from proben_amd.pipeline import FramePairPipeline
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_images, synthetic_state_dict
    models = [GeneralizedRCNN(DetectorConfig(), synthetic_state_dict(50, 3, 3, seed=s)) for s in (1, 2)]
    ft = torch.from_numpy(synthetic_images(4, 256, 320, seed=7)).cuda()
    fr = torch.from_numpy(synthetic_images(4, 256, 320, seed=8)).cuda()
    outs = []
    for concurrent, staggered in ((False, False), (True, False), (True, False), (True, True)):
        pipe = FramePairPipeline(models, "probEn", "v-avg", concurrent=concurrent, staggered=staggered)
        if staggered:   # throughput mode: several batches in flight, results valid after wait()
            for _ in range(3):
                dets, fused = pipe([ft, fr], [(256, 320)] * 4, (800, 1000))
            pipe.wait()
        else:
            dets, fused = pipe([ft, fr], [(256, 320)] * 4, (800, 1000))
        torch.cuda.synchronize()
        outs.append((dets, fused))
    (d0, f0) = outs[0]
    for d1, f1 in outs[1:]:
        for a, b in zip(d0, d1):
            assert torch.equal(a["counts"], b["counts"])
            for i, c in enumerate(a["counts"].tolist()):
                assert torch.equal(a["boxes"][i, :c], b["boxes"][i, :c]) and torch.equal(a["scores"][i, :c], b["scores"][i, :c])
        assert torch.equal(f0["counts"], f1["counts"])
        off, cnt = f0["offsets"].tolist(), f0["counts"].tolist()
        for o, c in zip(off, cnt):
            assert torch.equal(f0["boxes"][o:o + c], f1["boxes"][o:o + c]) and torch.equal(f0["scores"][o:o + c], f1["scores"][o:o + c])


def test_cli_end_to_end_on_synthetic_dataset(tmp_path):
    """save_predictions (x2 methods) -> val_<method>_predictions.json -> demo_probEn -> AP table, on a synthetic
    FLIR-shaped dataset written as real JPEG files (the reference's directory layout and file names)."""
    import json
    from PIL import Image
    from proben_amd.cli import demo_probEn, save_predictions
    from proben_amd.synthetic import synthetic_images
    root = tmp_path / "val"
    (root / "thermal_8_bit").mkdir(parents=True)
    (root / "RGB").mkdir()
    n, H, W = 6, 256, 320
    th, rgb = synthetic_images(n, H, W, seed=21), synthetic_images(n, H + 40, W + 60, seed=22)
    images, anns = [], []
    for i in range(n):
        Image.fromarray(th[i]).save(root / "thermal_8_bit" / f"FLIR_{i:05d}.jpeg", quality=95)
        Image.fromarray(rgb[i]).save(root / "RGB" / f"FLIR_{i:05d}.jpg", quality=95)
        images.append({"id": i, "file_name": f"thermal_8_bit/FLIR_{i:05d}.jpeg", "height": H, "width": W})
        anns.append({"id": i + 1, "image_id": i, "category_id": 1 + i % 3, "bbox": [20, 30, 60, 80], "area": 4800, "iscrowd": 0})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"},
                                                                      {"id": 3, "name": "car"}]},
              open(root / "FLIR_thermal_RGBT_pairs_val.json", "w"))
    pred = tmp_path / "pred"
    for method in ("thermal_only", "early_fusion"):
        save_predictions.main(["--dataset_path", str(root), "--prediction_path", str(pred), "--fusion_method", method,
                               "--batch", "4", "--outfolder", str(tmp_path / "out")])
        d = json.load(open(pred / f"val_{method}_predictions.json"))
        assert list(d) == ["image", "boxes", "scores", "classes", "image_id", "class_logits", "probs", "vars"] and len(d["image"]) == n
    res = demo_probEn.main(["--dataset_path", str(root), "--prediction_path", str(pred), "--outfolder", str(tmp_path / "out"),
                            "--detectors", "thermal_only,early_fusion", "--score_fusion", "probEn", "--box_fusion", "v-avg",
                            "--dataset_name", "flir_cli_test"])
    assert "bbox" in res and "AP50" in res["bbox"]


def test_kaist_single_class_pipeline():
    """Config 5 shape: K = 1 detectors (R50-FPN) + the binary ProbEn form (demo_probEn.py:24-30)."""
    import proben_amd  # noqa: F401
    from proben_amd.pipeline import FramePairPipeline
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_images, synthetic_state_dict
    models = [GeneralizedRCNN(DetectorConfig(num_classes=1), synthetic_state_dict(50, 1, 3, seed=s)) for s in (3, 4)]
    fr = [torch.from_numpy(synthetic_images(2, 256, 320, seed=9 + i)).cuda() for i in range(2)]
    dets, fused = FramePairPipeline(models, "probEn_binary", "v-avg", max_class=0)(fr, [(256, 320)] * 2, (800, 1000))
    assert dets[0]["prob_score"].shape[2] == 1 and dets[0]["class_logits"].shape[2] == 2
    c = fused["counts"].tolist()
    assert all(x >= 0 for x in c)
    for o, n in zip(fused["offsets"].tolist(), c):
        s = fused["scores"][o:o + n]
        assert bool(((s >= 0) & (s <= 1)).all()) and bool((fused["classes"][o:o + n] == 0).all())


def test_device_rows_match_host_evaluator_rows():
    """fused_rows_device == what FLIREvaluator.process builds from Instances (class whitelist, 5/7 -> 2, xywh)."""
    import proben_amd  # noqa: F401
    from proben_amd import evaluation, late_fusion as LF
    from proben_amd.structures import Boxes, Instances
    g = torch.Generator().manual_seed(3)
    B, S = 5, 12
    counts = torch.tensor([12, 0, 7, 3, 9], dtype=torch.int32)
    boxes = torch.rand(B * S, 4, generator=g, dtype=torch.float64) * 300
    boxes[:, 2:] += boxes[:, :2] + 1
    fused = {"boxes": boxes.cuda(), "scores": torch.rand(B * S, generator=g).cuda(),
             "classes": torch.randint(0, 9, (B * S,), generator=g).float().cuda(), "counts": counts.cuda(),
             "offsets": (torch.arange(B) * S).int().cuda(), "stride": S}
    ids = [100, 101, 102, 103, 104]
    rows = LF.fused_rows_device(fused, ids).cpu().numpy()
    want = []
    for b in range(B):
        c = int(counts[b])
        inst = Instances((512, 640))
        inst.pred_boxes = Boxes(boxes[b * S:b * S + c].float())
        inst.scores = fused["scores"][b * S:b * S + c].cpu()
        inst.pred_classes = fused["classes"][b * S:b * S + c].cpu()
        want += evaluation.instances_to_coco_json(inst, ids[b])
    assert len(rows) == len(want)
    for r, w in zip(rows, want):
        assert int(r[0]) == w["image_id"] and int(r[6]) == w["category_id"]
        np.testing.assert_allclose(r[1:5], w["bbox"], rtol=1e-5, atol=1e-4)   # the host path goes through float32 boxes
        assert r[5] == pytest.approx(w["score"], rel=1e-6)


def test_full_size_pipeline_deterministic_and_batch_invariant():
    """BASELINE configs[2] at its real size (two R101-FPN detectors, 640x512 frames -> 800x1000, ProbEn probEn / v-avg):
    two runs of the configuration's own batch of 32 pairs are bit-identical, and pair 5's fused rows equal those of a
    batch holding only that pair (size-independent properties; the oracle comparison at this size is test_r101_full_size_matches_oracle)."""
    import proben_amd  # noqa: F401
    from proben_amd.pipeline import FramePairPipeline
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_images, synthetic_state_dict
    models = [GeneralizedRCNN(DetectorConfig(), synthetic_state_dict(101, 3, 3, seed=s)) for s in (1, 2)]
    ft = torch.from_numpy(synthetic_images(32, seed=21)).cuda()
    fr = torch.from_numpy(synthetic_images(32, seed=22)).cuda()
    pipe = FramePairPipeline(models, "probEn", "v-avg")
    runs = []
    for _ in range(2):
        dets, fused = pipe([ft, fr], [(512, 640)] * 32, (800, 1000))
        torch.cuda.synchronize()
        runs.append((dets, fused))
    (d0, f0), (d1, f1) = runs
    for a, b in zip(d0, d1):
        for k in ("boxes", "scores", "classes", "counts", "vars", "prob_score"):
            assert torch.equal(a[k], b[k]), k
    assert torch.equal(f0["counts"], f1["counts"]) and torch.equal(f0["offsets"], f1["offsets"])
    for o, c in zip(f0["offsets"].tolist(), f0["counts"].tolist()):      # rows beyond counts are uninitialised padding
        for k in ("boxes", "scores", "classes"):
            assert torch.equal(f0[k][o:o + c], f1[k][o:o + c]), k
    assert int(f0["counts"].sum()) > 0
    _, one = pipe([ft[5:6], fr[5:6]], [(512, 640)], (800, 1000))
    torch.cuda.synchronize()
    o, c = int(f0["offsets"][5]), int(f0["counts"][5])
    assert c == int(one["counts"][0])
    o1 = int(one["offsets"][0])
    for k in ("boxes", "scores", "classes"):
        assert torch.equal(f0[k][o:o + c], one[k][o1:o1 + c]), k


def test_staggered_pipeline_with_reallocated_inputs():
    """A loader that creates and frees its batch tensors every step (the caching allocator may recycle that memory for the
    next batch while the side streams still read the old one unless the pipeline records the streams): results must equal
    the serial run for every step."""
    import proben_amd  # noqa: F401
    from proben_amd.pipeline import FramePairPipeline, HostFeeder
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_images, synthetic_state_dict
    models = [GeneralizedRCNN(DetectorConfig(), synthetic_state_dict(50, 3, 3, seed=s)) for s in (1, 2)]
    host = [(synthetic_images(2, 256, 320, seed=20 + i), synthetic_images(2, 256, 320, seed=40 + i)) for i in range(4)]
    serial = FramePairPipeline(models, concurrent=False)
    want = []
    for a, b in host:
        d, f = serial([torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()], [(256, 320)] * 2, (800, 1000))
        torch.cuda.synchronize()
        want.append((d[0], f))

    def same(d, f, wd, wf):     # padded layouts: compare the valid rows only
        assert torch.equal(d[0]["counts"], wd["counts"]) and torch.equal(f["counts"], wf["counts"])
        for i, c in enumerate(wd["counts"].tolist()):
            assert torch.equal(d[0]["boxes"][i, :c], wd["boxes"][i, :c])
        for o, c in zip(wf["offsets"].tolist(), wf["counts"].tolist()):
            assert torch.equal(f["scores"][o:o + c], wf["scores"][o:o + c]) and torch.equal(f["boxes"][o:o + c], wf["boxes"][o:o + c])
    pipe = FramePairPipeline(models, concurrent=True, staggered=True)
    got = []
    for a, b in host:
        fa, fb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()     # fresh allocations every step ...
        got.append(pipe([fa, fb], [(256, 320)] * 2, (800, 1000)))
        del fa, fb                                                           # ... freed while the side streams still read them
        junk = torch.full((2, 256, 320, 3), 255, dtype=torch.uint8, device="cuda")   # tries to reuse the freed blocks
        del junk
    pipe.wait(got[-1])
    torch.cuda.synchronize()
    for (d, f), (wd, wf) in zip(got, want):
        same(d, f, wd, wf)
    # the pinned-memory double-buffered uploader feeds the same pipeline
    feeder = HostFeeder(lambda it=iter(host * 2): [torch.from_numpy(x).pin_memory() for x in next(it)], "cuda")
    for i in range(4):
        batch = feeder.next()
        d, f = pipe(batch, [(256, 320)] * 2, (800, 1000))
        feeder.mark_consumed(*pipe.streams)
        pipe.wait((d, f))
        torch.cuda.synchronize()
        same(d, f, *want[i])


def test_config3_three_detector_pipeline_full_size():
    """BASELINE configs[3] at its real size: thermal (3-ch) + early fusion (4-ch stem) + middle fusion (6-ch: two backbone
    passes, 512-channel RPN / box head) R101-FPN detectors on 640x512 frames -> 800x1000, 3-way ProbEn (probEn / v-avg):
    the device-to-device pipeline equals the reference-style route (per-detector prediction lists -> late_fusion driver,
    demo_probEn.py:198-298), and the evaluation rows a rank would contribute are well formed."""
    import proben_amd  # noqa: F401
    from proben_amd import late_fusion as LF
    from proben_amd.pipeline import FramePairPipeline
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_images, synthetic_state_dict
    B = 4
    models, frames = [], []
    for i, (ch, fmt) in enumerate(((3, "BGR"), (4, "BGRT"), (6, "BGRTTT"))):
        mean = (103.53, 116.28, 123.675) + (135.438,) * (ch - 3)
        models.append(GeneralizedRCNN(DetectorConfig(input_format=fmt, pixel_mean=mean, pixel_std=(1.0,) * ch),
                                      synthetic_state_dict(101, 3, ch, seed=i + 1)))
        frames.append(torch.from_numpy(synthetic_images(B, channels=ch, seed=70 + i)).cuda())
    assert models[2].w.middle_fusion and models[2].w.rpn_channels == 512
    pipe = FramePairPipeline(models, "probEn", "v-avg")
    dets, fused = pipe(frames, [(512, 640)] * B, (800, 1000))
    torch.cuda.synchronize()
    assert fused["stride"] == 300 and all(int(d["counts"].min()) > 0 for d in dets)
    j1 = [LF.predictions_to_j1([f"im{i}.jpeg" for i in range(B)], list(range(B)), [o["instances"] for o in m.to_instances(d)])
          for m, d in zip(models, dets)]
    host = LF.late_fusion(j1, ["probEn", "v-avg"])
    cnt, off = fused["counts"].cpu().numpy(), fused["offsets"].cpu().numpy()
    for i in range(B):
        b, s, c = host[i]
        sl = slice(off[i], off[i] + cnt[i])
        assert cnt[i] == len(s) and cnt[i] > 0
        np.testing.assert_array_equal(fused["classes"][sl].cpu().numpy(), c.numpy())
        np.testing.assert_allclose(fused["scores"][sl].cpu().numpy(), s.numpy(), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(fused["boxes"][sl].cpu().numpy(), np.asarray(b), rtol=1e-9, atol=1e-9, equal_nan=True)
    rows = LF.fused_rows_device(fused, list(range(100, 100 + B))).cpu().numpy()
    assert rows.shape[1] == 7 and len(rows) <= int(cnt.sum()) and set(rows[:, 6].astype(int)) <= {0, 1, 2}
    assert np.all(rows[:, 3] > 0) and np.all(rows[:, 4] > 0) and np.all((rows[:, 5] > 0) & (rows[:, 5] <= 1))


def test_late_fusion_driver_reproduces_the_references_records(golden_dir):
    """proben_amd.late_fusion.apply_late_fusion_and_evaluate (same call as demo_probEn.py:198; the fused images go through ONE batched
    launch of the ProbEn kernel) against tests/golden/p5_cases.json - the records the REFERENCE's function handed to its evaluator when
    tests/golden/gen_p5.py ran it on the same three prediction dicts: which images are skipped / passed through / fused (2 and 3
    detectors, every firing pattern), file_name, image_id (detector 2's), height / width, float32 boxes / scores / classes in the
    reference's row order, for the 11 (score, box) combinations."""
    import json
    import os
    import proben_amd  # noqa: F401
    from proben_amd import late_fusion as LF

    class Recorder:
        def reset(self):
            self.rows = []

        def process(self, inputs, outputs):
            for i, o in zip(inputs, outputs):
                inst = o["instances"]
                assert inst.pred_boxes.tensor.dtype == torch.float32 and inst.scores.dtype == torch.float32 and inst.pred_classes.dtype == torch.float32
                self.rows.append(dict(i, boxes=inst.pred_boxes.tensor.cpu().numpy(), scores=inst.scores.cpu().numpy(), classes=inst.pred_classes.cpu().numpy(),
                                      image_size=tuple(inst.image_size)))

        def evaluate(self):
            return {"recorded": len(self.rows)}

    z = json.load(open(os.path.join(golden_dir, "p5_cases.json")))
    assert len(z["runs"]) == 22
    for key, want in z["runs"].items():
        sm, bm, kdet = key.rsplit("_", 2)
        rec = Recorder()
        res = LF.apply_late_fusion_and_evaluate(None, rec, z["det_1"], z["det_2"], [sm, bm], det_3=z["det_3"] if kdet == "3" else "")
        assert res == {"recorded": len(want)}, key
        for g, w in zip(rec.rows, want):
            assert (g["file_name"], g["image_id"], g["height"], g["width"]) == (w["file_name"], w["image_id"], w["height"], w["width"]), key
            assert g["image_size"] == (w["height"], w["width"])
            np.testing.assert_array_equal(g["classes"], np.asarray(w["classes"], np.float32))
            np.testing.assert_allclose(g["scores"], np.asarray(w["scores"], np.float32), rtol=1e-6, atol=0)
            np.testing.assert_allclose(g["boxes"], np.asarray(w["boxes"], np.float32).reshape(-1, 4), rtol=1e-6, atol=1e-5)
