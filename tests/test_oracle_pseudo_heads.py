"""The pseudo-trained-head fixture (tests/golden/pseudo_heads_r101.npz, generator gen_pseudo_heads.py) is consistent with the oracle
in THIS container: the committed detections of its first evaluation frame are what oracle/detector.py computes from the committed
head tensors, and the committed AP table is what the evaluator derives from the committed rows.  (The -m gpu counterpart compares
the HIP detector with these rows; this one keeps the fixture itself honest on every CPU run.)"""
import os

import numpy as np
import torch


def test_fixture_rows_and_table_reproduce(golden_dir):
    import proben_amd  # noqa: F401
    from PIL import Image
    from oracle import detector as D
    from proben_amd import evaluation
    from proben_amd.data import resize_shortest_edge_shape
    from proben_amd.synthetic import labelled_frames, synthetic_state_dict
    z = np.load(os.path.join(golden_dir, "pseudo_heads_r101.npz"))
    sd = synthetic_state_dict(int(z["depth"]), 3, 3, seed=int(z["seed"]))
    fitted = [k for k in z.files if "/" in k]
    assert len(fitted) == 8
    for k in fitted:
        name = k.replace("/", ".")
        assert tuple(sd[name].shape) == tuple(z[k].shape), name
        sd[name] = torch.from_numpy(z[k])
    n_eval = int(z["n_eval"])
    frames, gts = labelled_frames(n_eval, seed=int(z["eval_seed"]))
    rows = z["oracle_rows"]
    assert rows.shape[1] == 7 and int(rows[:, 0].max()) == n_eval - 1
    # the committed table from the committed rows (numpy evaluator: no GPU library needed for the host C++ either way)
    images = [{"id": i, "height": 512, "width": 640, "file_name": f"{i}.jpeg"} for i in range(n_eval)]
    cats = [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"}, {"id": 3, "name": "car"}]
    anns, aid = [], 1
    for i, (b, c) in enumerate(gts):
        for bb, cc in zip(b, c):
            w, h = float(bb[2] - bb[0]), float(bb[3] - bb[1])
            anns.append({"id": aid, "image_id": i, "category_id": int(cc) + 1, "bbox": [float(bb[0]), float(bb[1]), w, h], "area": w * h, "iscrowd": 0})
            aid += 1
    dets = [{"image_id": int(r[0]), "category_id": int(r[6]) + 1, "bbox": [float(r[1]), float(r[2]), float(r[3] - r[1]), float(r[4] - r[2])], "score": float(r[5])}
            for r in rows]
    ev = evaluation.COCOevalBBox({"images": images, "annotations": anns, "categories": cats}, dets, impl="native")
    ev.evaluate()
    ev.accumulate()
    np.testing.assert_allclose(np.asarray(ev.summarize(printer=None)), z["oracle_stats"], rtol=0, atol=1e-12)
    assert z["oracle_stats"][1] > 0.8          # a detector worth comparing against: AP50 > 80 on the known objects
    # the first frame through the oracle
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    r = np.array(Image.fromarray(frames[0]).resize((new_hw[1], new_hw[0]), Image.BILINEAR))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    o = D.forward([torch.from_numpy(r).permute(2, 0, 1).float().contiguous()], sd, D.DetectorSpec(depth=int(z["depth"])), out_sizes=[(512, 640)])[0]
    want = rows[rows[:, 0] == 0]
    assert len(o["scores"]) == len(want) > 5
    np.testing.assert_allclose(o["boxes"].numpy(), want[:, 1:5], rtol=0, atol=2e-2)
    np.testing.assert_allclose(o["scores"].numpy(), want[:, 5], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(o["classes"].numpy(), want[:, 6].astype(np.int64))


def test_fused_map_fixture_is_consistent(golden_dir):
    """tests/golden/fused_map_sets.npz (gen_fused_map.py: the oracle's rows of BOTH pseudo-trained detectors, with class probabilities and
    variances, on the disjoint evaluation sets) against what is already pinned: the thermal detector's rows are the committed single-detector
    rows (same weights, same frames: bit for bit on the fixture's own set); a row's score is its class's probability; the RGB detector's first frame re-derives from
    the committed RGB heads; and the oracle route (oracle.proben on the two lists) gives a fused AP table with the NaN scores the
    reference's own `1 - sum(p)` background makes (demo_probEn.py:32-42) ordered last by both evaluators alike."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import proben_amd  # noqa: F401
    from PIL import Image
    from oracle import detector as D
    from parity_map import FUSED_METHODS, coco_stats, load_fused_fixture, oracle_fused_rows
    from proben_amd.data import resize_shortest_edge_shape
    e = np.load(os.path.join(golden_dir, "fused_map_sets.npz"))
    single = np.load(os.path.join(golden_dir, "pseudo_heads_r101.npz"))
    more = np.load(os.path.join(golden_dir, "pseudo_heads_r101_sets.npz"))
    assert np.array_equal(e["t_7002"][:, :7], single["oracle_rows"])
    from proben_amd.synthetic import labelled_frames
    for k in e.files:
        if k.startswith("t_") and "rows_" + k[2:] in more.files:      # generated in other rounds with another thread count: the fp32 convolution
            a, b = e[k][:, :7], more["rows_" + k[2:]]                 # sums differ in their last bits, so a couple of detections sit on the other
            assert abs(len(a) - len(b)) <= 0.001 * len(b), k          # side of the 0.5 threshold (most sets: bit for bit) - the same detector:
            if not np.array_equal(a, b):                              # ... or two near-tied rows swap places: the same AP table to 0.05 point
                _, gts_k = labelled_frames(int(e["n_frames"]), seed=int(k[2:]))
                np.testing.assert_allclose(coco_stats(gts_k, a)[:3], coco_stats(gts_k, b)[:3], rtol=0, atol=5e-4, err_msg=k)
        if k[:2] in ("t_", "r_"):
            r = e[k]
            assert r.shape[1] == 11 and np.array_equal(r[:, 5], r[np.arange(len(r)), 7 + r[:, 6].astype(int)]), k      # score = prob[class]
            assert (r[:, 10] > 0).all(), k                                                                              # variance = exp(.)
    assert sum(1 for k in e.files if k.startswith("t_") and "r_" + k[2:] in e.files) >= 4, "at least four evaluation sets with both detectors' rows"
    sds, sets = load_fused_fixture(golden_dir, max_sets=1)
    name, ft, fr, gts, ot, orr = sets[0]
    # the RGB detector's first frame through the oracle (the thermal one is covered by the test above)
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    img = np.array(Image.fromarray(fr[0]).resize((new_hw[1], new_hw[0]), Image.BILINEAR))
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    o = D.forward([torch.from_numpy(img).permute(2, 0, 1).float().contiguous()], sds[1], D.DetectorSpec(depth=101), out_sizes=[(512, 640)])[0]
    want = orr[orr[:, 0] == 0]
    assert len(o["scores"]) == len(want) > 5
    np.testing.assert_allclose(o["boxes"].numpy(), want[:, 1:5], rtol=0, atol=2e-2)
    np.testing.assert_allclose(o["prob_score"].numpy(), want[:, 7:10], rtol=0, atol=2e-4)
    np.testing.assert_allclose(o["vars"].numpy().reshape(-1), want[:, 10], rtol=1e-3, atol=0)
    # the oracle route on one set: a detector pair worth fusing, and NaN rows that do not disturb the order of the others
    for method in FUSED_METHODS:
        rows = oracle_fused_rows(ot, orr, len(ft), method)
        stats = coco_stats(gts, rows)
        assert stats[1] > 0.7, (method, stats[:3])
        nan = np.isnan(rows[:, 5])
        if method[0] == "probEn":
            assert 0 < nan.sum() < 0.01 * len(rows)          # a handful of members with 1 - sum(p) < 0
            moved = np.concatenate([rows[~nan], rows[nan]])   # NaN rows at the end of the file instead of in place: they sort last either way
            assert np.array_equal(coco_stats(gts, moved), stats)
