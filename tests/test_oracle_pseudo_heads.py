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
