"""ProbEn HIP kernel (through the C-ABI) vs golden vectors and vs the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SCORE = ["probEn", "avg", "max"]
BOX = ["v-avg", "s-avg", "avg", "argmax"]
COMBOS = [(s, b) for s in SCORE for b in BOX if not (s == "max" and b == "argmax")]


@pytest.fixture(scope="module")
def cases(golden_dir):
    return np.load(os.path.join(golden_dir, "proben_cases.npz"))


def load_case(z, ci):
    return [{"img_name": "x", "bbox": z[f"c{ci}_d{di}_bbox"], "score": z[f"c{ci}_d{di}_score"],
             "class": z[f"c{ci}_d{di}_class"], "prob": z[f"c{ci}_d{di}_prob"], "vars": z[f"c{ci}_d{di}_vars"]}
            for di in range(int(z[f"c{ci}_ndet"]))]


def canon(b, s, c):
    key = np.lexsort((b[:, 3], b[:, 2], b[:, 1], b[:, 0]))
    return b[key], s[key], c[key]


@pytest.mark.parametrize("sm,bm", COMBOS)
def test_hip_matches_reference_goldens(cases, sm, bm):
    """Bit-exact keep set / classes; scores within 1e-6 rel (device log/exp vs glibc),
    boxes within 1e-9 (float64 sums in the same order)."""
    from proben_amd import fusion as F
    n = int(cases["num_cases"])
    for ci in range(n):
        b, s, c = F.fusion([sm, bm], *load_case(cases, ci))
        b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
        s, c = s.numpy(), c.numpy()
        rb, rs, rc = cases[f"c{ci}_{sm}_{bm}_boxes"], cases[f"c{ci}_{sm}_{bm}_scores"], cases[f"c{ci}_{sm}_{bm}_classes"]
        assert b.shape == rb.shape, (ci, sm, bm)
        if ci == n - 1:
            b, s, c = canon(b, s, c)
            rb, rs, rc = canon(rb, rs, rc)
        np.testing.assert_array_equal(c, rc)
        np.testing.assert_allclose(s, rs, rtol=1e-6, atol=0, equal_nan=True)
        np.testing.assert_allclose(b, rb, rtol=1e-9, atol=1e-9, equal_nan=True)


def synth_batch(B, seed, kdet=2, nmax=100, K=3):
    from proben_amd.synthetic import synth_detections
    return synth_detections(B, seed, kdet=kdet, nmax=nmax, K=K)


@pytest.mark.parametrize("sm,bm", [("probEn", "v-avg"), ("avg", "s-avg"), ("max", "avg"), ("probEn", "argmax")])
@pytest.mark.parametrize("kdet", [2, 3])
def test_hip_matches_oracle_batched(sm, bm, kdet):
    from oracle import proben as O
    from proben_amd import fusion as F
    per_image = synth_batch(48, seed=11 + kdet, kdet=kdet)
    per_image[3] = [dict(per_image[3][0], bbox=np.zeros((0, 4)), score=np.zeros(0), **{"class": np.zeros(0, int)},
                         prob=np.zeros((0, 3)), vars=np.zeros((0, 1)))] * kdet  # an empty image
    b, s, p, v, c, offs = F.pack_infos(per_image)
    out = F.fuse_batch(b, s, p, v, c, offs, sm, bm)
    counts = out["counts"].cpu().numpy()
    offs_h = offs.cpu().numpy()
    ob, os_, oc, ok = out["boxes"].cpu().numpy(), out["scores"].cpu().numpy(), out["classes"].cpu().numpy(), out["keep"].cpu().numpy()
    for i, infos in enumerate(per_image):
        n = offs_h[i + 1] - offs_h[i]
        if n == 0:
            assert counts[i] == 0
            continue
        rb, rs, rcl, rp, rv = O.concat_infos(infos)
        keep, es, eb, ec = O.nms_bayesian(rb, rs, rcl, rp, rv, 0.5, sm, bm)
        m = counts[i]
        assert m == len(keep), i
        sl = slice(offs_h[i], offs_h[i] + m)
        np.testing.assert_array_equal(ok[sl], keep)
        np.testing.assert_array_equal(oc[sl], ec.astype(np.float32))
        np.testing.assert_allclose(os_[sl], es.astype(np.float32), rtol=1e-6, equal_nan=True)
        np.testing.assert_allclose(ob[sl], eb, rtol=1e-9, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("sm,bm", [("probEn", "v-avg"), ("avg", "s-avg"), ("max", "avg"), ("probEn", "argmax")])
def test_image_bound_moves_no_bit(sm, bm):
    """The per-image row bound sizes the LDS carve and picks the clustering form of csrc/proben.hip: with the detectors' bound (the
    longest image here) the pair tests go into two bit matrices and wave 0 walks those; at 1100 rows (147 KB) the matrices do not
    fit and the walk computes the IoUs on the way, one pivot after the other.  The same outputs; a bound that does not fit 160 KiB
    at all is refused, not truncated."""
    from proben_amd import _lib, fusion as F
    per_image = synth_batch(40, seed=23, kdet=3)
    b, s, p, v, c, offs = F.pack_infos(per_image)
    tight = F.fuse_batch(b, s, p, v, c, offs, sm, bm)
    wide = F.fuse_batch(b, s, p, v, c, offs, sm, bm, max_rows=1100)
    assert torch.equal(tight["counts"], wide["counts"]) and int(tight["counts"].sum()) > 0
    offs_h, cnt = offs.cpu().numpy(), tight["counts"].cpu().numpy()
    for i in range(len(per_image)):
        sl = slice(offs_h[i], offs_h[i] + cnt[i])
        assert torch.equal(tight["keep"][sl], wide["keep"][sl])
        assert torch.equal(tight["boxes"][sl].view(torch.int64), wide["boxes"][sl].view(torch.int64))
        assert torch.equal(tight["scores"][sl].view(torch.int32), wide["scores"][sl].view(torch.int32))
        assert torch.equal(tight["classes"][sl], wide["classes"][sl])
    with pytest.raises(_lib.HipLibraryError, match="LDS"):
        F.fuse_batch(b, s, p, v, c, offs, sm, bm, max_rows=2000)


def test_binary_mode_k1():
    """K = 1 (KAIST, config 5): pe score mode PROBEN_BINARY == demo_probEn.py:24-30."""
    from oracle import proben as O
    from proben_amd import fusion as F
    per_image = synth_batch(16, seed=5, kdet=2, K=1)
    for infos in per_image:
        for d in infos:
            d["class"] = np.zeros(len(d["score"]), int)
    b, s, p, v, c, offs = F.pack_infos(per_image)
    out = F.fuse_batch(b, s, p, v, c, offs, "probEn_binary", "v-avg")
    counts, offs_h = out["counts"].cpu().numpy(), offs.cpu().numpy()
    for i, infos in enumerate(per_image):
        if offs_h[i + 1] == offs_h[i]:
            continue
        keep, es, eb, ec = O.nms_bayesian(*O.concat_infos(infos), 0.5, "probEn_binary", "v-avg")
        sl = slice(offs_h[i], offs_h[i] + counts[i])
        assert counts[i] == len(keep)
        np.testing.assert_allclose(out["scores"].cpu().numpy()[sl], es.astype(np.float32), rtol=1e-6)
        np.testing.assert_allclose(out["boxes"].cpu().numpy()[sl], eb, rtol=1e-9, atol=1e-9)


def test_full_size_properties():
    """BASELINE size (B = 4096 images, <= 300 rows): size-independent properties.
    (1) idempotence: fusing an already-fused list with 'avg'/'avg' and nothing overlapping changes nothing;
    (2) every output class-band row set is mutually non-overlapping at IoU 0.5 w.r.t. the pivots' boxes;
    (3) permutation of images leaves per-image results unchanged."""
    from proben_amd import fusion as F
    per_image = synth_batch(4096, seed=2, kdet=3)
    b, s, p, v, c, offs = F.pack_infos(per_image)
    out = F.fuse_batch(b, s, p, v, c, offs, "probEn", "v-avg")
    counts = out["counts"]
    n_in = offs[1:] - offs[:-1]
    assert bool((counts >= 0).all()) and bool((counts <= n_in).all())
    assert bool(((counts > 0) == (n_in > 0)).all())
    # (3) reverse the image order
    rev = per_image[::-1]
    b2, s2, p2, v2, c2, offs2 = F.pack_infos(rev)
    out2 = F.fuse_batch(b2, s2, p2, v2, c2, offs2, "probEn", "v-avg")
    assert torch.equal(out2["counts"].flip(0), counts)
    oh, oh2 = offs.cpu().numpy(), offs2.cpu().numpy()
    ch = counts.cpu().numpy()
    B = len(per_image)
    for i in (0, 1, 777, 4095):
        j = B - 1 - i
        a = out["boxes"][oh[i]:oh[i] + ch[i]]
        bb = out2["boxes"][oh2[j]:oh2[j] + ch[i]]
        assert torch.equal(a, bb)
    # (1) idempotence of the keep set under passthrough modes: pivots' own boxes ('avg' score, 'argmax' box
    # keeps the pivot's or a member's box) - rerun on the kept pivots only -> no further merging beyond IoU rule
    outk = F.fuse_batch(b, s, p, v, c, offs, "avg", "argmax")
    assert torch.equal(outk["counts"], counts)  # clustering is independent of the fusion formulas
    for i in (0, 1, 777, 4095):
        assert torch.equal(outk["keep"][oh[i]:oh[i] + ch[i]], out["keep"][oh[i]:oh[i] + ch[i]])
