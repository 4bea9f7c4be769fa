"""Pins the oracle's ROIAlign backward (oracle/csrc/roi_align.c, restating ROIAlign_cpu.cpp:221-394) to the oracle forward,
which is itself bit-exact against the reference's compiled kernel and its test tables: the forward is linear in the
features, so its matrix can be probed with unit inputs, and the backward must be that matrix transposed."""
import numpy as np
import pytest
import torch

from oracle import roi_align as O


@pytest.mark.parametrize("aligned,ratio", [(True, 0), (True, 2), (False, 0), (False, 3)])
def test_backward_is_the_transpose_of_the_forward(aligned, ratio):
    N, C, H, W, ph, pw = 2, 1, 6, 7, 3, 2
    rois = torch.tensor([[0, 0.3, 0.2, 5.7, 4.9], [1, 2.0, 1.0, 6.9, 5.8], [0, -1.5, -0.5, 2.0, 9.0], [1, 3.0, 3.0, 3.0, 3.0]], dtype=torch.float32)
    scale = 0.75
    K = rois.shape[0]
    A = np.zeros((K * C * ph * pw, N * C * H * W), dtype=np.float64)      # forward matrix, one unit input per column
    for j in range(N * C * H * W):
        x = torch.zeros(N * C * H * W)
        x[j] = 1.0
        A[:, j] = O.roi_align_forward(x.view(N, C, H, W), rois, scale, ph, pw, ratio, aligned).reshape(-1).double().numpy()
    g = torch.randn(K, C, ph, pw, generator=torch.Generator().manual_seed(3))
    got = O.roi_align_backward(g, rois, scale, ph, pw, N, C, H, W, ratio, aligned).reshape(-1).double().numpy()
    want = A.T @ g.reshape(-1).double().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)


def test_backward_shapes_and_empty():
    out = O.roi_align_backward(torch.zeros(0, 3, 7, 7), torch.zeros(0, 5), 0.25, 7, 7, 2, 3, 10, 12, 0, True)
    assert out.shape == (2, 3, 10, 12) and float(out.abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        O.roi_align_backward(torch.ones(1, 1, 2, 2), torch.tensor([[0, 5.0, 5.0, 1.0, 1.0]]), 1.0, 2, 2, 1, 1, 8, 8, 0, True)
