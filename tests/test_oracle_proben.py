"""Oracle (oracle/proben.py) vs golden vectors produced by the reference's own
demo_probEn.fusion (tests/golden/gen_proben.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import proben as O

SCORE = ["probEn", "avg", "max"]
BOX = ["v-avg", "s-avg", "avg", "argmax"]


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(os.path.join(golden_dir, "proben_cases.npz"))


def load_case(z, ci):
    dets = []
    for di in range(int(z[f"c{ci}_ndet"])):
        dets.append({"img_name": "x", "bbox": z[f"c{ci}_d{di}_bbox"], "score": z[f"c{ci}_d{di}_score"],
                     "class": z[f"c{ci}_d{di}_class"], "prob": z[f"c{ci}_d{di}_prob"],
                     "vars": z[f"c{ci}_d{di}_vars"]})
    return dets


def canon(b, s, c):
    """Order-insensitive view for the tie case (reference tie order depends on the NumPy build)."""
    key = np.lexsort((b[:, 3], b[:, 2], b[:, 1], b[:, 0]))
    return b[key], s[key], c[key]


@pytest.mark.parametrize("sm", SCORE)
@pytest.mark.parametrize("bm", BOX)
def test_oracle_matches_reference_fusion(cases, sm, bm):
    if sm == "max" and bm == "argmax":
        pytest.skip("nms_1 route: torchvision absent when goldens were generated (parity unpinned)")
    n = int(cases["num_cases"])
    for ci in range(n):
        dets = load_case(cases, ci)
        b, s, c = O.fusion([sm, bm], *dets)
        rb, rs, rc = cases[f"c{ci}_{sm}_{bm}_boxes"], cases[f"c{ci}_{sm}_{bm}_scores"], cases[f"c{ci}_{sm}_{bm}_classes"]
        assert b.shape == rb.shape, (ci, sm, bm)
        if ci == n - 1:  # tie case
            b, s, c = canon(b, s, c)
            rb, rs, rc = canon(rb, rs, rc)
        np.testing.assert_array_equal(c, rc)
        np.testing.assert_allclose(s, rs, rtol=1e-6, atol=0, equal_nan=True)
        np.testing.assert_allclose(b, rb, rtol=1e-12, atol=1e-9, equal_nan=True)


def test_binary_bayesian_fusion(cases):
    vals = cases["binary_in"]
    off = 0
    for m, want in zip(cases["binary_len"], cases["binary_out"]):
        v = vals[off:off + m]
        off += m
        got, _ = O.fuse_score("probEn_binary", np.zeros((m, 1)), v, 0)
        assert got == pytest.approx(want, rel=1e-14)


def test_oracle_driver_reproduces_the_references_records(golden_dir):
    """oracle.proben.late_fusion_rows against tests/golden/p5_cases.json = what the REFERENCE's apply_late_fusion_and_evaluate
    (demo_probEn.py:198-298, executed by tests/golden/gen_p5.py with a recording evaluator) hands to `evaluator.process`: which images are
    skipped, passed through or fused (2 and 3 detectors, every firing pattern), file_name from detector 1's name with '.jpeg', image_id
    from detector 2, float32 boxes / scores / classes - for the 11 (score, box) combinations the reference runs without torchvision."""
    import json
    from oracle import proben as O
    z = json.load(open(os.path.join(golden_dir, "p5_cases.json")))
    assert len(z["runs"]) == 22
    for key, want in z["runs"].items():
        sm, bm, kdet = key.rsplit("_", 2)
        got = O.late_fusion_rows(z["det_1"], z["det_2"], [sm, bm], det_3=z["det_3"] if kdet == "3" else "")
        assert [r["file_name"] for r in got] == [r["file_name"] for r in want], key
        assert [r["image_id"] for r in got] == [r["image_id"] for r in want] and all(r["height"] == 512 and r["width"] == 640 for r in want)
        for g, w in zip(got, want):
            assert w["boxes_dtype"] == "torch.float32" and w["scores_dtype"] == "torch.float32" and w["classes_dtype"] == "torch.float32"
            np.testing.assert_array_equal(g["classes"], np.asarray(w["classes"], np.float32))
            np.testing.assert_allclose(g["scores"], np.asarray(w["scores"], np.float32), rtol=1e-6, atol=0)
            np.testing.assert_allclose(g["boxes"], np.asarray(w["boxes"], np.float32).reshape(-1, 4), rtol=1e-6, atol=1e-5)
    # the firing patterns of the fixture cover every branch of the driver's case split
    assert {tuple(f) for f in z["fire"]} == {(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)}
