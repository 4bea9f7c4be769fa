"""Training half of the box head (ranked last): the losses of `FastRCNNOutputs`
(modeling/roi_heads/fast_rcnn.py:149-388) including this repository's Gaussian negative-log-likelihood loss on the predicted
box variance (fast_rcnn.py:237-263), plus the differentiable ROIAlign (`layers.ROIAlign`, forward and backward on the gfx950
kernels).  The detector's convolution backward passes, an optimiser and DDP are NOT part of this build: inference is the hot
path; these pieces exist so that the variance head can be fine-tuned on frozen features with stock PyTorch autograd.

Plain tensor math on whatever device the tensors live on, like the reference's Python."""
import torch
import torch.nn.functional as F


def smooth_l1_loss(input, target, beta, reduction="none"):
    """fvcore.nn.smooth_l1_loss (fvcore 0.1.x, a dependency that is not vendored in the reference tree): L1 for beta < 1e-5,
    else 0.5 x^2 / beta below beta and |x| - 0.5 beta above."""
    n = torch.abs(input - target)
    loss = n if beta < 1e-5 else torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    return loss.sum() if reduction == "sum" else loss


class FastRCNNLosses:
    """`FastRCNNOutputs(...).losses()` for a batch that is already concatenated over images:
    pred_class_logits [R, K+1] (background last), pred_proposal_deltas [R, 4K] (or [R, 4] class-agnostic), variance [R, 1] or
    [R, 4] or empty, proposals / gt_boxes [R, 4] XYXY tensors, gt_classes [R] in [0, K] (K = background)."""

    def __init__(self, box2box_transform, pred_class_logits, pred_proposal_deltas, variance, proposals, gt_boxes, gt_classes, smooth_l1_beta=0.0):
        assert not proposals.requires_grad, "Proposals should not require gradients!"   # fast_rcnn.py:201-203
        self.box2box_transform = box2box_transform
        self.pred_class_logits = pred_class_logits
        self.pred_proposal_deltas = pred_proposal_deltas
        self.variance = variance
        self.proposals, self.gt_boxes, self.gt_classes = proposals, gt_boxes, gt_classes
        self.smooth_l1_beta = smooth_l1_beta
        self._no_instances = proposals.shape[0] == 0

    def _fg(self):
        bg = self.pred_class_logits.shape[1] - 1
        fg_inds = torch.nonzero((self.gt_classes >= 0) & (self.gt_classes < bg)).squeeze(1)
        gt_deltas = self.box2box_transform.get_deltas(self.proposals, self.gt_boxes)
        box_dim = gt_deltas.size(1)
        dev = self.pred_proposal_deltas.device
        if self.pred_proposal_deltas.size(1) == box_dim:          # class-agnostic regression
            cols = torch.arange(box_dim, device=dev)
        else:                                                      # columns [4k, 4k + 4) of the gt class k
            cols = box_dim * self.gt_classes[fg_inds][:, None] + torch.arange(box_dim, device=dev)
        return fg_inds, cols, gt_deltas

    def softmax_cross_entropy_loss(self):                         # fast_rcnn.py:265-283
        if self._no_instances:
            return 0.0 * self.pred_class_logits.sum()
        return F.cross_entropy(self.pred_class_logits, self.gt_classes, reduction="mean")

    def smooth_l1_loss(self):                                     # fast_rcnn.py:285-343: summed over foreground, divided by ALL regions
        if self._no_instances:
            return 0.0 * self.pred_proposal_deltas.sum()
        fg_inds, cols, gt_deltas = self._fg()
        loss = smooth_l1_loss(self.pred_proposal_deltas[fg_inds[:, None], cols], gt_deltas[fg_inds], self.smooth_l1_beta, reduction="sum")
        return loss / self.gt_classes.numel()

    def bbox_gaussian_loss(self):                                 # fast_rcnn.py:237-263: nn.GaussianNLLLoss()(pred, target, var), mean, eps 1e-6
        fg_inds, cols, gt_deltas = self._fg()
        return F.gaussian_nll_loss(self.pred_proposal_deltas[fg_inds[:, None], cols], gt_deltas[fg_inds], self.variance[fg_inds])

    def losses(self):                                             # fast_rcnn.py:367-385
        out = {"loss_cls": self.softmax_cross_entropy_loss(), "loss_box_reg": self.smooth_l1_loss()}
        if len(self.variance) > 0:
            out["gaussian_loss"] = self.bbox_gaussian_loss()
        return out
