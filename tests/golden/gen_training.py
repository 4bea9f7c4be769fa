"""Golden vectors for the training half, GENERATED from the reference's own code in the build container
(/root/reference/detectron2 through tests/golden/ref_harness.py): `FastRCNNOutputs`' cross entropy and this repository's Gaussian NLL
(modeling/roi_heads/fast_rcnn.py:237-283; the smooth-L1 box loss lives in fvcore, which is absent), `Matcher` + `pairwise_iou` labelling of proposals (modeling/matcher.py,
structures/boxes.py:266-300) and the counts `subsample_labels` draws (modeling/sampling.py:7-50).
    python tests/golden/gen_training.py        ->  tests/golden/training_cases.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness as H  # noqa: E402

H.install_detectron2_standins()
from detectron2.modeling.box_regression import Box2BoxTransform  # noqa: E402
from detectron2.modeling.matcher import Matcher  # noqa: E402
from detectron2.modeling.roi_heads.fast_rcnn import FastRCNNOutputs  # noqa: E402
from detectron2.modeling.sampling import subsample_labels  # noqa: E402
from detectron2.structures import Boxes, Instances, pairwise_iou  # noqa: E402

# losses() first LOGS accuracy figures into fvcore's event storage (absent here: a placeholder); logging is not arithmetic
FastRCNNOutputs._log_accuracy = lambda self: None



def boxes(g, n, size=400.0):
    xy = torch.rand(n, 2, generator=g) * size
    return torch.cat([xy, xy + 8 + torch.rand(n, 2, generator=g) * 150], 1)


def main():
    out = {}
    g = torch.Generator().manual_seed(20)
    t = Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    # ---- losses: three cases (class-specific deltas with variance, class-agnostic without variance, smooth-L1 beta > 0)
    for case, (R, K, agnostic, with_var, beta) in enumerate([(96, 3, False, True, 0.0), (64, 80, True, False, 0.0), (50, 3, False, True, 0.5)]):
        prop = boxes(g, R)
        gt = prop + torch.randn(R, 4, generator=g) * 6
        gt[:, 2:] = torch.maximum(gt[:, 2:], gt[:, :2] + 2)
        cls = torch.randint(0, K + 1, (R,), generator=g)
        logits = torch.randn(R, K + 1, generator=g)
        deltas = torch.randn(R, 4 if agnostic else 4 * K, generator=g) * 0.5
        var = torch.exp(torch.randn(R, 1, generator=g) * 0.3) if with_var else torch.Tensor([])
        inst = Instances((512, 640))
        inst.proposal_boxes, inst.gt_boxes, inst.gt_classes = Boxes(prop), Boxes(gt), cls
        o = FastRCNNOutputs(t, logits, deltas, [inst], smooth_l1_beta=beta, variance=var)
        # loss_box_reg delegates to fvcore.nn.smooth_l1_loss - a third-party package that is absent here (restated from its published
        # definition in proben_amd/training.py, checked against NumPy in tests/test_training_cpu.py): not generated
        losses = {"loss_cls": o.softmax_cross_entropy_loss()}
        if with_var:
            losses["gaussian_loss"] = o.bbox_gaussian_loss()
        out[f"loss{case}_gt_deltas"] = t.get_deltas(prop, gt).numpy()
        for k, v in dict(prop=prop, gt=gt, cls=cls, logits=logits, deltas=deltas, var=var).items():
            out[f"loss{case}_{k}"] = v.numpy()
        out[f"loss{case}_beta"] = np.float64(beta)
        for k, v in losses.items():
            out[f"loss{case}_out_{k}"] = np.float64(float(v))
    # ---- proposal labelling: Matcher([0.5], [0, 1], allow_low_quality_matches=False) over pairwise_iou(gt, proposals)
    m = Matcher([0.5], [0, 1], allow_low_quality_matches=False)
    for case, (P, G) in enumerate([(300, 5), (40, 1), (120, 9)]):
        gt = boxes(g, G)
        prop = torch.cat([boxes(g, P - 2 * G), gt + torch.randn(G, 4, generator=g) * 4, gt], 0)      # some near-hits and the gt itself
        iou = pairwise_iou(Boxes(gt), Boxes(prop))
        idx, lab = m(iou)
        gcls = torch.randint(0, 3, (G,), generator=g)
        cls = gcls[idx].clone()
        cls[lab == 0] = 3
        pos, neg = subsample_labels(cls, 64, 0.25, 3)
        out[f"match{case}_gt"], out[f"match{case}_prop"], out[f"match{case}_gt_classes"] = gt.numpy(), prop.numpy(), gcls.numpy()
        out[f"match{case}_iou"], out[f"match{case}_idx"], out[f"match{case}_classes"] = iou.numpy(), idx.numpy(), cls.numpy()
        out[f"match{case}_npos"], out[f"match{case}_nneg"] = np.int64(len(pos)), np.int64(len(neg))
    np.savez_compressed(os.path.join(HERE, "training_cases.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and v.shape else float(v)) for k, v in out.items() if "out_" in k or "npos" in k or "nneg" in k})


if __name__ == "__main__":
    main()
