"""The Pillow bilinear resize of the reference's 3-channel path (transform.py:92-97): oracle restatement and the
product's host tap tables against Pillow itself (when importable) and against the committed Pillow outputs."""
import os

import numpy as np
import pytest

import proben_amd  # noqa: F401
from oracle import resize as R
from proben_amd.data import pil_bilinear_tables, resize_shortest_edge_shape


def test_oracle_matches_committed_pillow_outputs(golden_dir):
    z = np.load(os.path.join(golden_dir, "pil_resize.npz"))
    n = len([k for k in z.files if k.startswith("in")])
    assert n >= 6
    for i in range(n):
        nh, nw = z[f"size{i}"]
        got = R.pil_bilinear_resize_u8(z[f"in{i}"], int(nh), int(nw))
        assert np.array_equal(got, z[f"out{i}"]), i


@pytest.mark.parametrize("case", [(512, 640, 800, 1000), (480, 640, 800, 1067), (720, 1280, 750, 1333), (100, 37, 271, 100)])
def test_oracle_matches_installed_pillow(case):
    Image = pytest.importorskip("PIL.Image")
    h, w, nh, nw = case
    if (h, w) != (100, 37):
        assert resize_shortest_edge_shape(h, w, 800, 1333) == (nh, nw)   # the sizes the FLIR / KAIST frames really get
    img = np.random.default_rng(h).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    assert np.array_equal(R.pil_bilinear_resize_u8(img, nh, nw), ref)


@pytest.mark.parametrize("sizes", [(640, 1000), (512, 800), (1280, 1333), (53, 91), (64, 31), (100, 100), (1, 5), (7, 1)])
def test_product_tap_tables_equal_oracle(sizes):
    t = pil_bilinear_tables(*sizes)
    bounds, kk = R.pil_bilinear_coeffs(*sizes)
    assert np.array_equal(t[:, :2], bounds) and np.array_equal(t[:, 2:], kk)
