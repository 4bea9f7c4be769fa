"""Box-head fine-tuning on frozen features (the training path, ranked last) - the part of the reference's
`demo/FLIR/demo_train_FLIR.py` (DefaultTrainer: engine/defaults.py:229-330) that the Gaussian-NLL variance head needs, one process per
GPU: the frozen detector (HIP inference path) supplies pyramid features and RPN proposals, proposals are labelled and sampled like
`ROIHeads.label_and_sample_proposals` (roi_heads/roi_heads.py:130-285, sampling.py:7-50, matcher.py), ROIAlign + the box head run
forward AND backward on the gfx950 kernels (`training.BoxHead`), gradients are averaged over ranks by `BucketedGradAllReduce` (RCCL)
and applied by the fused SGD kernel.  The convolutional layers are not trained (no convolution backward in this build).

Boxes here are in the detector's input frame (the resized image), like the reference's training targets."""
import math

import torch

from . import layers as L
from .training import BoxHead, BucketedGradAllReduce, FusedSGD, box_head_train_step


def pairwise_iou(a, b):
    """structures/boxes.py:266-300 (`pairwise_iou`): [A, 4] x [B, 4] XYXY -> [A, B]; 0 where the union is empty."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[None, :, 2:]) - torch.max(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None, :] - inter
    return torch.where(inter > 0, inter / union, torch.zeros_like(inter))


@torch.no_grad()
def label_and_sample_proposals(proposals, counts, gt_boxes, gt_classes, num_classes, batch_size_per_image=512, positive_fraction=0.25,
                               iou_threshold=0.5, append_gt=True, generator=None):
    """proposals [N, P, 4] with counts [N] live rows (the RPN's output), per-image ground truth lists -> fixed-shape training batch:
    boxes [N, S, 4], live [N] (rows in use), classes [N, S] in [0, K] (K = background), matched_gt [N, S, 4].
    Rule by rule the reference's: ground-truth boxes are appended to the proposals (PROPOSAL_APPEND_GT), every proposal is matched
    to its highest-IoU ground truth, below `iou_threshold` it is background; at most int(S x positive_fraction) foreground rows are
    drawn at random, the rest is filled with random background rows; an image without ground truth yields background only."""
    N, S, K = proposals.shape[0], batch_size_per_image, num_classes
    dev = proposals.device
    boxes = torch.zeros((N, S, 4), device=dev)
    classes = torch.full((N, S), K, dtype=torch.int64, device=dev)
    matched = torch.zeros((N, S, 4), device=dev)
    live = torch.zeros((N,), dtype=torch.int32, device=dev)
    cnt = counts.tolist()
    for n in range(N):
        gb = gt_boxes[n].to(dev).float().reshape(-1, 4)
        gc = gt_classes[n].to(dev).long().reshape(-1)
        p = proposals[n, :cnt[n]]
        if append_gt and len(gb):
            p = torch.cat([p, gb], 0)
        if len(gb):
            iou = pairwise_iou(gb, p)                       # [G, P']
            best, idx = iou.max(0)
            lab = torch.where(best >= iou_threshold, gc[idx], torch.full_like(idx, K))
            mg = gb[idx]
        else:
            lab = torch.full((len(p),), K, dtype=torch.int64, device=dev)
            mg = torch.zeros((len(p), 4), device=dev)
        pos = torch.nonzero(lab != K).squeeze(1)
        neg = torch.nonzero(lab == K).squeeze(1)
        num_pos = min(pos.numel(), int(S * positive_fraction))
        num_neg = min(neg.numel(), S - num_pos)
        perm_p = torch.randperm(pos.numel(), generator=generator)[:num_pos].to(dev)
        perm_n = torch.randperm(neg.numel(), generator=generator)[:num_neg].to(dev)
        sel = torch.cat([pos[perm_p], neg[perm_n]], 0)
        m = sel.numel()
        boxes[n, :m], classes[n, :m], matched[n, :m], live[n] = p[sel], lab[sel], mg[sel], m
    return boxes, live, classes, matched


class BoxHeadFineTuner:
    """model: a GeneralizedRCNN (frozen; its box-head weights seed the trainable copy when `init_from_model`).  `step(frames, targets)`
    = one SGD step on this rank's batch; `export()` writes the trained weights back into the model's inference head."""

    def __init__(self, model, lr=0.005, momentum=0.9, weight_decay=1e-4, batch_size_per_image=512, positive_fraction=0.25,
                 loss_scale=1024.0, seed=0, init_from_model=True, bucket_bytes=64 << 20, clip_grad_norm=0.0):
        from .modeling import Box2BoxTransform
        self.model, self.cfg = model, model.cfg
        K = self.cfg.num_classes
        self.S, self.pf, self.loss_scale, self.clip = batch_size_per_image, positive_fraction, loss_scale, clip_grad_norm
        self.head = BoxHead(49 * model.w.rpn_channels, K, model.device, seed=seed)      # 49 x 256, 49 x 512 for a middle-fusion model
        if init_from_model:
            self.load_from_model()
        self.opt = FusedSGD(self.head.flat, lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.reducer = BucketedGradAllReduce(self.head.flat, bucket_bytes=bucket_bytes)      # broadcasts rank 0's head when world > 1
        self.transform = Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))                   # ROI_BOX_HEAD.BBOX_REG_WEIGHTS
        self.gen = torch.Generator().manual_seed(seed)

    def load_from_model(self):
        w, f, n = self.model.w, self.head.flat, self.head.cols
        with torch.no_grad():
            f["fc1.weight"].copy_(w.fc1[0].float()); f["fc1.bias"].copy_(w.fc1[1])
            f["fc2.weight"].copy_(w.fc2[0].float()); f["fc2.bias"].copy_(w.fc2[1])
            f["predictor.weight"].zero_(); f["predictor.bias"].zero_()
            f["predictor.weight"][:n].copy_(w.predictor[0][:n].float()); f["predictor.bias"][:n].copy_(w.predictor[1][:n])
        f.refresh_shadow()

    def export(self):
        """Trained head -> the detector's inference weights (fc1 is already in the ROIAlign kernel's (ph, pw, c) column order)."""
        w, f, n = self.model.w, self.head.flat, self.head.cols
        with torch.no_grad():
            w.fc1 = (f.half("fc1.weight").clone(), f["fc1.bias"].detach().clone())
            w.fc2 = (f.half("fc2.weight").clone(), f["fc2.bias"].detach().clone())
            w.predictor = (f.half("predictor.weight")[:n].clone().contiguous(), f["predictor.bias"].detach()[:n].clone())
            w.has_var = True

    def reference_state_dict(self):
        """The trained head under the REFERENCE's checkpoint keys (what DetectionCheckpointer saves and `weights.load_state_dict_file`
        / `GeneralizedRCNN` load): `roi_heads.box_head.fc{1,2}.{weight,bias}` with fc1's columns back in the reference's (c, ph, pw)
        order (the inverse of weights.py's permutation to the ROIAlign kernel's (ph, pw, c)), and the fused predictor split back into
        `roi_heads.box_predictor.{cls_score,bbox_pred,var_pred}.{weight,bias}` without its zero padding."""
        f, K, C = self.head.flat, self.head.K, self.model.w.rpn_channels
        w1 = f["fc1.weight"].detach().float().cpu()
        w1 = w1.view(w1.shape[0], 7, 7, C).permute(0, 3, 1, 2).reshape(w1.shape[0], 49 * C).contiguous()
        pw, pb = f["predictor.weight"].detach().float().cpu(), f["predictor.bias"].detach().float().cpu()
        q, h = "roi_heads.box_predictor.", "roi_heads.box_head."
        return {h + "fc1.weight": w1, h + "fc1.bias": f["fc1.bias"].detach().float().cpu().clone(),
                h + "fc2.weight": f["fc2.weight"].detach().float().cpu().clone(), h + "fc2.bias": f["fc2.bias"].detach().float().cpu().clone(),
                q + "cls_score.weight": pw[:K + 1].clone(), q + "cls_score.bias": pb[:K + 1].clone(),
                q + "bbox_pred.weight": pw[K + 1:5 * K + 1].clone(), q + "bbox_pred.bias": pb[K + 1:5 * K + 1].clone(),
                q + "var_pred.weight": pw[5 * K + 1:5 * K + 2].clone(), q + "var_pred.bias": pb[5 * K + 1:5 * K + 2].clone()}

    def state_dict(self):
        """What a resumed run needs (the reference's DetectionCheckpointer saves model + optimizer + scheduler, engine/defaults.py:264-275):
        master weights, momentum, the optimiser's step count and the sampler's generator state."""
        f = self.head.flat
        return {"names": list(f.names), "shapes": dict(f.shapes), "master": f.master.detach().cpu().clone(), "momentum": f.momentum.cpu().clone(),
                "steps": self.opt.steps, "lr": self.opt.lr, "generator": self.gen.get_state()}

    def load_state_dict(self, sd):
        f = self.head.flat
        if list(sd["names"]) != list(f.names) or {k: tuple(v) for k, v in sd["shapes"].items()} != f.shapes:
            raise ValueError("checkpoint of a different box head (tensor names / shapes differ)")
        with torch.no_grad():
            f.master.copy_(sd["master"].to(f.master.device))
            f.momentum.copy_(sd["momentum"].to(f.momentum.device))
        f.refresh_shadow()
        self.opt.steps, self.opt.lr = int(sd["steps"]), float(sd["lr"])
        self.gen.set_state(sd["generator"])

    @torch.no_grad()
    def _features(self, frames, resize_to):
        det = self.model.forward_batch(frames, resize_to=resize_to, keep_intermediates=True)
        return det["_feats"], det["proposals"], det["proposal_counts"], det["image_sizes"]

    def step(self, frames, gt_boxes, gt_classes, resize_to=None, lr=None):
        """frames: what forward_batch takes; gt_boxes / gt_classes: per-image tensors in the ORIGINAL frame's pixels (scaled here to the
        detector's input size like the reference's dataset mapper does with its transforms)."""
        feats, props, pcnt, sizes = self._features(frames, resize_to)
        N = props.shape[0]
        gtb = []
        for n in range(N):      # every image by its own frame size and its own resized size
            h0, w0 = (frames.shape[1], frames.shape[2]) if isinstance(frames, torch.Tensor) else (frames[n].shape[0], frames[n].shape[1])
            sy, sx = sizes[n][0] / h0, sizes[n][1] / w0
            gtb.append(gt_boxes[n].float().cpu().reshape(-1, 4) * torch.tensor([sx, sy, sx, sy]))
        boxes, live, classes, matched = label_and_sample_proposals(props, pcnt, gtb, gt_classes, self.cfg.num_classes, self.S, self.pf, generator=self.gen)
        pooled = L.roi_align_nhwc(feats[:4], boxes, scales=[1 / 4, 1 / 8, 1 / 16, 1 / 32], pooled=(7, 7), sampling_ratio=0, aligned=True,
                                  counts=live, per_image=self.S, num_rois=N * self.S)
        keep = (torch.arange(self.S, device=boxes.device)[None, :] < live[:, None]).reshape(-1)
        rows = torch.nonzero(keep).squeeze(1)
        if lr is not None:
            self.opt.lr = lr
        losses = box_head_train_step(self.head, self.opt, self.reducer, pooled.view(N * self.S, -1)[rows], boxes.view(-1, 4)[rows],
                                     matched.view(-1, 4)[rows], classes.view(-1)[rows], self.transform, loss_scale=self.loss_scale,
                                     clip_grad_norm=self.clip)
        with torch.no_grad():
            fg = classes.view(-1)[rows] < self.cfg.num_classes
            losses["foreground_fraction"] = float(fg.float().mean())
        return losses


def multistep_lr(base_lr, step, milestones=(), gamma=0.1, warmup_iters=1000, warmup_factor=0.001, warmup_method="linear"):
    """WarmupMultiStepLR (solver/lr_scheduler.py:16-51, 86-115; SOLVER.STEPS / GAMMA / WARMUP_ITERS / WARMUP_FACTOR / WARMUP_METHOD):
    lr = base x warm-up factor x gamma ** (number of milestones <= step)."""
    if step >= warmup_iters:
        warm = 1.0
    elif warmup_method == "constant":
        warm = warmup_factor
    elif warmup_method == "linear":
        alpha = step / warmup_iters
        warm = warmup_factor * (1 - alpha) + alpha
    else:
        raise ValueError(f"Unknown warmup method: {warmup_method}")
    return base_lr * warm * gamma ** sum(1 for m in milestones if m <= step)


def warmup_lr(base_lr, step, warmup_iters=100, warmup_factor=0.001):
    """The linear warm-up alone (short fine-tuning runs)."""
    return multistep_lr(base_lr, step, (), 0.1, warmup_iters, warmup_factor)
