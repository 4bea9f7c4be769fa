"""Training half of the box head (ranked last): the losses of `FastRCNNOutputs`
(modeling/roi_heads/fast_rcnn.py:149-388) including this repository's Gaussian negative-log-likelihood loss on the predicted
box variance (fast_rcnn.py:237-263), plus the differentiable ROIAlign (`layers.ROIAlign`, forward and backward on the gfx950
kernels).  Round 4 adds what fine-tuning the box head on frozen features needs on MI355X, one process per GPU:
  * `FlatParams`: every trainable tensor is a view into ONE flat fp32 buffer (+ flat gradient, momentum and fp16-shadow buffers);
  * `HipLinear`: the FC layers' forward AND backward on the gfx950 MFMA GEMM (dX = dY W, dW = dY^T X as two more launches of the
    same kernel; fp16 operands, fp32 accumulation, fp32 weight gradients);
  * `BucketedGradAllReduce`: DistributedDataParallel's job (engine/defaults.py:257-262) for flat gradients - contiguous buckets
    all-reduced over RCCL (gloo in the CPU tests) as soon as backward has produced them, overlapping the rest of backward;
  * `FusedSGD`: torch.optim.SGD as solver/build.py:93-133 configures it, as ONE kernel per parameter group
    (csrc/optim.hip: momentum, weight decay, the 1 / world_size of DDP, the inverse loss scale and the fp16 shadow refresh fused);
  * `BoxHead` + `box_head_train_step`: FastRCNNConvFCHead + FastRCNNOutputLayers with this repository's variance head
    (roi_heads/box_head.py:20-80, fast_rcnn.py:454-545) trained with the losses below.
The detector's CONVOLUTION backward passes are not part of this build (the backbone stays frozen): inference is the hot path.

The losses are plain tensor math on whatever device the tensors live on, like the reference's Python."""
import math

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


# ------------------------------------------------------------------------------------------------------------------------------
# Flat parameters, fused SGD, bucketed gradient all-reduce, HIP linear layers (round 4)
# ------------------------------------------------------------------------------------------------------------------------------
class FlatParams:
    """`shapes`: ordered {name: shape}.  One flat fp32 master buffer, one flat gradient buffer, one flat momentum buffer and one
    flat fp16 shadow; `self[name]` is an nn.Parameter VIEW of the master whose `.grad` is the matching view of the gradient buffer
    (autograd accumulates into it in place), `self.half(name)` the fp16 view the GEMMs read.  Every tensor starts at a multiple of
    4 elements (16-byte aligned: vector loads in csrc/optim.hip, whole all-reduce buckets)."""

    def __init__(self, shapes, device, groups=None):
        self.names, self.offsets, self.shapes = [], {}, {}
        off = 0
        for name, shape in shapes.items():
            n = int(math.prod(shape))
            self.names.append(name)
            self.offsets[name] = (off, n)
            self.shapes[name] = tuple(shape)
            off += (n + 3) // 4 * 4
        self.numel = off
        self.master = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.momentum = torch.zeros(off, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(off, dtype=torch.float16, device=device)
        self.params = {}
        for name in self.names:
            o, n = self.offsets[name]
            p = torch.nn.Parameter(self.master[o:o + n].view(self.shapes[name]))
            p.grad = self.grad[o:o + n].view(self.shapes[name])
            self.params[name] = p
        # parameter groups of solver/build.py:93-133: ("weights" | "bias") -> contiguous runs of names with the same hyper-parameters
        self.group_of = groups or (lambda name: "bias" if name.endswith("bias") else "weights")

    def __getitem__(self, name):
        return self.params[name]

    def half(self, name):
        o, n = self.offsets[name]
        return self.shadow[o:o + n].view(self.shapes[name])

    def refresh_shadow(self):
        self.shadow.copy_(self.master)

    def zero_grad(self):
        self.grad.zero_()

    def runs(self):
        """[(group, first element, element count)]: maximal contiguous runs of tensors of one group (padding included)."""
        out = []
        for name in self.names:
            o, n = self.offsets[name]
            n4 = (n + 3) // 4 * 4
            g = self.group_of(name)
            if out and out[-1][0] == g and out[-1][1] + out[-1][2] == o:
                out[-1] = (g, out[-1][1], out[-1][2] + n4)
            else:
                out.append((g, o, n4))
        return out


class FusedSGD:
    """torch.optim.SGD(params, lr, momentum) with the per-group lr / weight decay of solver/build.py:93-133
    (BASE_LR x BIAS_LR_FACTOR, WEIGHT_DECAY / WEIGHT_DECAY_BIAS), one launch of pe_sgd_momentum_f32 per contiguous group run."""

    def __init__(self, flat, lr=0.001, momentum=0.9, weight_decay=0.0001, bias_lr_factor=1.0, weight_decay_bias=None):
        self.flat, self.lr, self.mu = flat, lr, momentum
        self.hyper = {"weights": (1.0, weight_decay), "bias": (bias_lr_factor, weight_decay if weight_decay_bias is None else weight_decay_bias)}
        self.steps = 0

    def step(self, grad_scale=1.0, lr=None):
        from . import _lib
        f = self.flat
        _lib.require_cuda(f.master)
        base = self.lr if lr is None else lr
        for group, o, n in f.runs():
            factor, wd = self.hyper[group]
            st = _lib.lib().pe_sgd_momentum_f32(_lib.ptr(f.master[o:]), _lib.ptr(f.grad[o:]), _lib.ptr(f.momentum[o:]), _lib.ptr(f.shadow[o:]),
                                                n, base * factor, self.mu, wd, grad_scale, int(self.steps == 0), _lib.stream())
            _lib.check(st, "pe_sgd_momentum_f32")
        self.steps += 1


class BucketedGradAllReduce:
    """What torch's DistributedDataParallel does for the reference's trainer (engine/defaults.py:257-262), for FlatParams: the flat
    gradient buffer is cut - at tensor boundaries, walking the tensors in REVERSE registration order, i.e. roughly in the order
    backward produces them - into contiguous buckets of ~`bucket_bytes`; a hook on every parameter counts its bucket down and the
    bucket's SUM all-reduce is launched (async, RCCL on the device / gloo in the CPU tests) the moment it is complete, so that it
    overlaps the rest of backward.  `finish()` launches whatever did not fire (a parameter without gradient this step), waits, and
    returns 1 / world_size - the factor FusedSGD folds into its kernel instead of a separate averaging pass.
    Bucket size: xGMI is point to point (7 links x ~153 GB/s per GPU), ring all-reduce is per-link bound and pays its latency per
    collective: 64 MiB buckets keep a ResNet-101 detector at ~4 collectives per step instead of DDP's default 25 MiB."""

    def __init__(self, flat, bucket_bytes=64 << 20, group=None, broadcast=True):
        import torch.distributed as dist
        self.dist, self.flat, self.group = dist, flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # flags that must agree over ranks travel on the device with RCCL, on the host with gloo
        self.backend_is_device = bool(dist.is_initialized() and dist.get_backend(group) != "gloo")
        self.collective = dist.is_initialized()      # a ONE-rank group (PROBEN_FORCE_DIST=1) still runs its collectives: the RCCL path on a one-GPU box
        self.buckets, self.bucket_of = [], {}
        hi, cur = flat.numel, []
        for name in reversed(flat.names):
            o, _ = flat.offsets[name]
            cur.append(name)
            if (hi - o) * 4 >= bucket_bytes or name == flat.names[0]:
                for nm in cur:
                    self.bucket_of[nm] = len(self.buckets)
                self.buckets.append((o, hi, len(cur)))
                hi, cur = o, []
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.handles, self.order = [], []
        for name in flat.names:
            flat[name].register_post_accumulate_grad_hook(lambda p, nm=name: self._ready(nm))
        if broadcast and self.collective:                     # DDP's initial parameter broadcast from rank 0
            src = dist.get_global_rank(group, 0) if group is not None else 0
            if flat.master.is_cuda and dist.get_backend(group) == "gloo":
                host = flat.master.cpu()
                dist.broadcast(host, src=src, group=group)
                flat.master.copy_(host)
            else:
                dist.broadcast(flat.master, src=src, group=group)
            flat.refresh_shadow()

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self.launched[b] = True
        self.order.append(b)
        if self.collective:
            view = self.flat.grad[lo:hi]
            if view.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # ranks sharing one device for testing (PROBEN_DIST_BACKEND=gloo): stage through the host, synchronously
                host = view.cpu()
                self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group)
                view.copy_(host)
            else:
                self.handles.append(self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _ready(self, name):
        b = self.bucket_of[name]
        self.pending[b] -= 1
        if self.pending[b] == 0 and not self.launched[b]:
            self._launch(b)

    def finish(self):
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for h in self.handles:
            h.wait()
        order = self.order
        self.handles, self.order = [], []
        self.pending = [b[2] for b in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.last_order = order
        return 1.0 / self.world


def _pad_rows(t, multiple):
    r = t.shape[0] % multiple
    return t if r == 0 else torch.cat([t, t.new_zeros((multiple - r,) + tuple(t.shape[1:]))], 0)


class HipLinear(torch.autograd.Function):
    """y = x W^T + b (+ ReLU) on the gfx950 GEMM (csrc/conv_igemm2.hip through layers.linear_f16), forward and backward.
    x [M, K] fp16; `weight` / `bias` are the fp32 master parameters (they receive the gradients), `weight16` their fp16 shadow (what
    the MFMAs read).  Backward = two more launches of the same kernel: dX = dY W (fp16 out) and dW = dY^T X (fp32 out; the GEMM's
    reduction dimension is the row count M, zero-padded to a multiple of 64), db = column sums.  K and the output width must be
    multiples of 64 (dX's reduction dimension is the output width): pad the last layer's columns."""

    @staticmethod
    def forward(ctx, x, weight, bias, weight16, relu, out_f32):
        from . import layers as L
        y = L.linear_f16(x, weight16, bias, relu=relu, out_f32=out_f32)
        ctx.save_for_backward(x, weight16, y if relu else None)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import layers as L
        x, w16, y = ctx.saved_tensors
        dy = dy.to(torch.float16)
        if ctx.relu:
            dy = dy * (y > 0)
        dy = dy.contiguous()
        dx = L.linear_f16(dy, w16.t().contiguous(), None) if ctx.needs_input_grad[0] else None          # [M, K] = dY [M, N] x W [N, K]
        dyt, xt = _pad_rows(dy, 64).t().contiguous(), _pad_rows(x, 64).t().contiguous()               # [N, M'], [K, M']
        dw = L.linear_f16(dyt, xt, None, out_f32=True)                                                 # [N, K] = dY^T x X
        db = dy.float().sum(0)
        return dx, dw, db, None, None, None


class BoxHead:
    """FastRCNNConvFCHead (two FC layers, roi_heads/box_head.py:23-80) + FastRCNNOutputLayers with the variance predictor of this
    repository (fast_rcnn.py:508-543: cls_score [K + 1], bbox_pred [4K], var_pred [1] -> variance = exp(.)) as ONE predictor GEMM whose
    columns are [scores | deltas | log-variance | zero padding to a multiple of 64] - the column layout of the inference head
    (rcnn.py) -; parameters in a FlatParams."""

    def __init__(self, in_features, num_classes, device, fc_dim=1024, seed=0):
        K = num_classes
        self.K, self.cols = K, (K + 1) + 4 * K + 1
        self.cols_padded = (self.cols + 63) // 64 * 64
        # weights first, biases last: two parameter-group runs = two optimizer launches
        self.flat = FlatParams({"fc1.weight": (fc_dim, in_features), "fc2.weight": (fc_dim, fc_dim), "predictor.weight": (self.cols_padded, fc_dim),
                                "fc1.bias": (fc_dim,), "fc2.bias": (fc_dim,), "predictor.bias": (self.cols_padded,)}, device)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():      # c2_xavier_fill for the FCs, normal(0.01 / 0.001 / 0.01) for cls_score / bbox_pred / var_pred (box_head.py:60-63, fast_rcnn.py:493-512: cls_score 0.01, bbox_pred 0.001, var_pred 0.01)
            for nm, fan in (("fc1.weight", in_features), ("fc2.weight", fc_dim)):
                bound = math.sqrt(3.0 / fan)
                self.flat[nm].copy_(((torch.rand(self.flat.shapes[nm], generator=g) * 2 - 1) * bound).to(device))
            w = torch.zeros(self.cols_padded, fc_dim)
            w[:K + 1] = torch.randn(K + 1, fc_dim, generator=g) * 0.01
            w[K + 1:K + 1 + 4 * K] = torch.randn(4 * K, fc_dim, generator=g) * 0.001
            w[K + 1 + 4 * K:self.cols] = torch.randn(1, fc_dim, generator=g) * 0.01
            self.flat["predictor.weight"].copy_(w.to(device))
        self.flat.refresh_shadow()

    def forward(self, pooled):
        """pooled [R, ...] fp16 (ROIAlign output, any trailing shape) -> (scores [R, K + 1], deltas [R, 4K], variance [R, 1]) fp32;
        variance = exp(var_pred(x)) (fast_rcnn.py:541-543)."""
        f = self.flat
        x = pooled.reshape(pooled.shape[0], -1)
        x = HipLinear.apply(x, f["fc1.weight"], f["fc1.bias"], f.half("fc1.weight"), True, False)
        x = HipLinear.apply(x, f["fc2.weight"], f["fc2.bias"], f.half("fc2.weight"), True, False)
        h = HipLinear.apply(x, f["predictor.weight"], f["predictor.bias"], f.half("predictor.weight"), False, True)
        K = self.K
        return h[:, :K + 1], h[:, K + 1:K + 1 + 4 * K], torch.exp(h[:, K + 1 + 4 * K:self.cols])


def box_head_train_step(head, optimizer, reducer, pooled, proposals, gt_boxes, gt_classes, box2box_transform, loss_scale=1024.0,
                        smooth_l1_beta=0.0, clip_grad_norm=0.0):
    """One SGD step of the box head on frozen features: forward (HIP GEMMs) -> FastRCNNLosses -> backward (HIP GEMMs; the loss is
    scaled so that fp16 activation gradients do not underflow) -> bucketed all-reduce (overlapped) -> fused SGD with
    grad_scale = 1 / (world_size x loss_scale).  `clip_grad_norm` > 0 = SOLVER.CLIP_GRADIENTS with CLIP_TYPE "norm", NORM_TYPE 2:
    like the reference's `optimizer_wgc_step` (solver/build.py:19-36, 66-90) every parameter TENSOR is clipped by its own norm.
    Returns the unscaled losses (+ "skipped": 1.0 when the step was not applied).

    Non-finite values are handled COLLECTIVELY, so that no rank leaves its peers waiting in an all-reduce: a non-finite loss on any
    rank is agreed on over the process group BEFORE backward and raises FloatingPointError on every rank; a non-finite gradient
    after the all-reduce (an fp16 overflow of the scaled activation gradients - every rank sees the same reduced buffer) skips the
    step on every rank and leaves weights, momentum and the fp16 shadow untouched."""
    head.flat.zero_grad()
    scores, deltas, variance = head.forward(pooled)
    losses = FastRCNNLosses(box2box_transform, scores, deltas, variance, proposals, gt_boxes, gt_classes, smooth_l1_beta).losses()
    total = sum(losses.values())
    out = {k: float(v.detach()) for k, v in losses.items()}
    bad = 0.0 if all(math.isfinite(v) for v in out.values()) else 1.0
    if reducer is not None and reducer.world > 1:
        flag = torch.tensor([bad], device=head.flat.grad.device if reducer.backend_is_device else "cpu")
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=reducer.group)
        bad = float(flag.item())
    if bad:
        raise FloatingPointError(f"box-head training diverged on at least one rank (this rank's losses {out}): lower the learning rate or clip gradients")
    (total * loss_scale).backward()
    scale = (reducer.finish() if reducer is not None else 1.0) / loss_scale
    flat = head.flat
    if not bool(torch.isfinite(flat.grad).all()):
        out["skipped"] = 1.0
        return out
    if clip_grad_norm > 0:
        for name in flat.names:
            o, n = flat.offsets[name]
            g = flat.grad[o:o + n]
            coef = clip_grad_norm / (float(g.norm()) * scale + 1e-6)
            if coef < 1.0:
                g.mul_(coef)
    optimizer.step(grad_scale=scale)
    return out
