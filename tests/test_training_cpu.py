"""Training-side pieces: the box-head losses against independent NumPy formulas."""
import numpy as np
import torch

import proben_amd  # noqa: F401
from proben_amd.modeling import Box2BoxTransform
from proben_amd.training import FastRCNNLosses, smooth_l1_loss


def _batch(R=40, K=3, seed=5):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(R, 2, generator=g) * 300
    prop = torch.cat([xy, xy + 20 + torch.rand(R, 2, generator=g) * 100], 1)
    gt = prop + torch.randn(R, 4, generator=g) * 4
    gt[:, 2:] = torch.maximum(gt[:, 2:], gt[:, :2] + 1)
    cls = torch.randint(0, K + 1, (R,), generator=g)
    logits = torch.randn(R, K + 1, generator=g, requires_grad=True)
    deltas = torch.randn(R, 4 * K, generator=g, requires_grad=True)
    var = (torch.rand(R, 1, generator=g) + 0.2).requires_grad_()
    return prop, gt, cls, logits, deltas, var


def test_losses_match_numpy_formulas_and_backpropagate():
    prop, gt, cls, logits, deltas, var = _batch()
    t = Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    L = FastRCNNLosses(t, logits, deltas, var, prop, gt, cls, smooth_l1_beta=0.5).losses()
    lg = logits.detach().double().numpy()
    lse = np.log(np.exp(lg - lg.max(1, keepdims=True)).sum(1)) + lg.max(1)
    np.testing.assert_allclose(float(L["loss_cls"]), float(np.mean(lse - lg[np.arange(len(cls)), cls.numpy()])), rtol=1e-5)
    fg = np.nonzero(cls.numpy() < 3)[0]
    tgt = t.get_deltas(prop, gt).double().numpy()[fg]
    pred = np.stack([deltas.detach().double().numpy()[i, 4 * cls[i]: 4 * cls[i] + 4] for i in fg])
    n = np.abs(pred - tgt)
    sl1 = np.where(n < 0.5, 0.5 * n ** 2 / 0.5, n - 0.25).sum() / len(cls)
    np.testing.assert_allclose(float(L["loss_box_reg"]), sl1, rtol=1e-5)
    v = np.maximum(var.detach().double().numpy()[fg], 1e-6)
    nll = (0.5 * (np.log(v) + (pred - tgt) ** 2 / v)).mean()
    np.testing.assert_allclose(float(L["gaussian_loss"]), nll, rtol=1e-5)
    sum(L.values()).backward()
    assert logits.grad is not None and deltas.grad is not None and var.grad is not None
    assert float(var.grad[np.nonzero(cls.numpy() == 3)[0]].abs().sum()) == 0.0      # background rows get no variance gradient


def test_empty_batch_and_l1_limit():
    t = Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    z = FastRCNNLosses(t, torch.zeros(0, 4, requires_grad=True), torch.zeros(0, 12, requires_grad=True), torch.zeros(0, 1), torch.zeros(0, 4),
                       torch.zeros(0, 4), torch.zeros(0, dtype=torch.long))
    assert float(z.softmax_cross_entropy_loss()) == 0.0 and float(z.smooth_l1_loss()) == 0.0
    a, b = torch.tensor([0.0, 2.0, -3.0]), torch.tensor([1.0, 0.0, 0.0])
    assert torch.equal(smooth_l1_loss(a, b, 0.0), torch.tensor([1.0, 2.0, 3.0]))
    np.testing.assert_allclose(smooth_l1_loss(a, b, 4.0).numpy(), [0.125, 0.5, 1.125])
