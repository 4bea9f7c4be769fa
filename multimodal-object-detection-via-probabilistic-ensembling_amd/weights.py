"""Turn a reference-format state dict (SURVEY A.3 key names; what `torch.save(model.state_dict())`
of the reference's GeneralizedRCNN holds, demo/FLIR/demo_train_FLIR.py:113) into the packed device
tensors the HIP kernels consume:

  * FrozenBatchNorm2d folded into the preceding conv: scale = gamma * rsqrt(var + 1e-5),
    shift = beta - mean * scale (layers/batch_norm.py:45-65) -> fp16 weight * scale, fp32 bias = shift;
  * conv weights [Cout,Cin,KH,KW] -> [Cout,KH,KW,Cin] fp16 (K contiguous for the MFMA B operand);
  * 7x7 stem -> [64, 8, 8, 4] (zero row / column / channel padding, see csrc/conv_igemm.hip);
  * RPN objectness (3) + anchor_deltas (12) fused into one [15, C] head;
  * fc1 columns permuted from (c, ph, pw) to the ROIAlign kernel's (ph, pw, c) order;
  * cls_score (K+1) + bbox_pred (4K) + var_pred (1) fused into one [5K+2, 1024] predictor.
"""
import torch

BN_EPS = 1e-5
STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}  # backbone/resnet.py:515-519


def load_state_dict_file(path):
    """A checkpoint file -> {reference key: tensor} (checkpoint/detection_checkpoint.py:27-45):
      * `.pth` written by torch.save: a raw state dict or {"model": state_dict};
      * `.pkl` in the detectron2 model-zoo format: {"model": {key: ndarray}, "__author__": ...} with the state dict's own key names
        (trained_models/Detectron2_pretrained/model_final_f6e8b1.pkl of demo_FLIR_save_predictions.py:59-61, the COCO R101-FPN
        detector behind `--fusion_method rgb_only`).  A Caffe2 / Detectron1 pickle ("blobs", or no "__author__") needs the
        reference's name-matching heuristics (c2_model_loading.py), which are not part of the inference path: refused."""
    if str(path).endswith(".pkl"):
        import pickle
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        if not (isinstance(data, dict) and "model" in data and "__author__" in data):
            raise ValueError(f"{path}: not a detectron2 model-zoo pickle (Caffe2 / Detectron1 weights need the reference's "
                             "key-matching heuristics; convert them with the reference and save a .pth)")
        obj = data["model"]
    else:
        obj = torch.load(path, map_location="cpu")
        if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
            obj = obj["model"]
    return {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in obj.items() if not k.endswith("_momentum")}


def infer_depth(sd, prefix="backbone.bottom_up"):
    n = len({k.split(".")[3] for k in sd if k.startswith(prefix + ".res4.")})
    for d, blocks in STAGE_BLOCKS.items():
        if blocks[2] == n:
            return d
    raise ValueError(f"unsupported ResNet depth: res4 has {n} blocks")


def _fold(sd, name):
    w = sd[name + ".weight"].float()
    scale = sd[name + ".norm.weight"].float() * torch.rsqrt(sd[name + ".norm.running_var"].float() + BN_EPS)
    shift = sd[name + ".norm.bias"].float() - sd[name + ".norm.running_mean"].float() * scale
    return w * scale.view(-1, 1, 1, 1), shift


def _pack_conv(w):
    return w.permute(0, 2, 3, 1).contiguous().half()


def _pack_stem(w):
    cout, cin = w.shape[0], w.shape[1]
    p = torch.zeros((cout, 8, 8, 4), dtype=torch.float32)
    p[:, :7, :7, :cin] = w.permute(0, 2, 3, 1)
    return p.half()


def pack_stem_fused(w):
    """[64, cin<=4, 7, 7] -> [64, 7, 8, 4] for csrc/stem.hip: tap t reads input column 2*c - 4 + t, so the real
    taps sit at t = 1..7 and t = 0 is a zero weight (keeps every MFMA fragment a 16-byte aligned pixel pair)."""
    cout, cin = w.shape[0], w.shape[1]
    p = torch.zeros((cout, 7, 8, 4), dtype=torch.float32)
    p[:, :, 1:, :cin] = w.permute(0, 2, 3, 1)
    return p.half()


class PackedDetector:
    """Device-resident packed weights of one detector."""

    def __init__(self, sd, device="cuda", num_classes=None):
        self.device = torch.device(device)
        self.depth = infer_depth(sd)
        self.has_backbone_2 = any(k.startswith("backbone_2.") for k in sd)
        self.stem_in = sd["backbone.bottom_up.stem.conv1.weight"].shape[1]
        self.rpn_channels = sd["proposal_generator.rpn_head.conv.weight"].shape[0]
        self.middle_fusion = self.rpn_channels == 512
        ncls = sd["roi_heads.box_predictor.cls_score.weight"].shape[0] - 1
        self.num_classes = num_classes if num_classes is not None else ncls
        assert self.num_classes == ncls, (self.num_classes, ncls)
        self.convs = {}
        self.wd = {}   # name -> fragment-ordered copy of a 3x3 weight for the weights-direct kernel (csrc/conv_wd.h)
        self.tails = {}            # block -> packed conv3 of a fused bottleneck tail      } filled by _pack_wd on a GPU; a host-packed
        self.chains64 = {}         # stage -> fused 64-wide chain (res2)                  } object keeps them empty, so rcnn raises its
        self.rpn_head_fused = None  # packed 15-row RPN head for the fused launch         } 'no CPU fallback' error, not AttributeError
        self._pack_backbone(sd, "backbone")
        if self.has_backbone_2:
            self._pack_backbone(sd, "backbone_2")
        dev = self.device
        p = "proposal_generator.rpn_head"
        self.convs["rpn.conv"] = (_pack_conv(sd[p + ".conv.weight"].float()).to(dev), sd[p + ".conv.bias"].float().to(dev))
        hw = torch.cat([sd[p + ".objectness_logits.weight"].float(), sd[p + ".anchor_deltas.weight"].float()], 0)
        hb = torch.cat([sd[p + ".objectness_logits.bias"].float(), sd[p + ".anchor_deltas.bias"].float()], 0)
        assert hw.shape[0] == 15, "expected 3 anchors per cell"
        self.convs["rpn.head"] = (_pack_conv(hw).to(dev), hb.to(dev))
        C = self.rpn_channels
        w1 = sd["roi_heads.box_head.fc1.weight"].float()
        w1 = w1.view(w1.shape[0], C, 7, 7).permute(0, 2, 3, 1).reshape(w1.shape[0], 49 * C)
        self.fc1 = (w1.contiguous().half().to(dev), sd["roi_heads.box_head.fc1.bias"].float().to(dev))
        self.fc2 = (sd["roi_heads.box_head.fc2.weight"].half().contiguous().to(dev), sd["roi_heads.box_head.fc2.bias"].float().to(dev))
        q = "roi_heads.box_predictor"
        has_var = (q + ".var_pred.weight") in sd
        pw = [sd[q + ".cls_score.weight"].float(), sd[q + ".bbox_pred.weight"].float()]
        pb = [sd[q + ".cls_score.bias"].float(), sd[q + ".bbox_pred.bias"].float()]
        if has_var:
            pw.append(sd[q + ".var_pred.weight"].float())
            pb.append(sd[q + ".var_pred.bias"].float())
        else:  # ENABLE_GAUSSIANNLLOSS off: log-variance 0 -> variance 1
            pw.append(torch.zeros(1, pw[0].shape[1]))
            pb.append(torch.zeros(1))
        self.has_var = has_var
        self.predictor = (torch.cat(pw, 0).half().contiguous().to(dev), torch.cat(pb, 0).to(dev))
        self.head_cols = 5 * self.num_classes + 2
        self.head_stride = (self.head_cols + 7) // 8 * 8
        self._pack_wd()

    def _pack_wd(self):
        """Second, fragment-ordered copy of every 3x3 weight the weights-direct kernel can take (Cin % 64 == 0,
        Cout % 256 == 0); whether a launch uses it depends on the feature-map width (layers.conv_wd_supported)."""
        if self.device.type != "cuda":
            return
        from . import layers as L
        for name, (w, b) in self.convs.items():
            if w.dim() == 4 and w.shape[1] == 3 and w.shape[2] == 3 and w.shape[3] % 64 == 0 and w.shape[0] % 256 == 0 and b is not None:
                self.wd[name] = L.conv_wd_pack(w)
        # fused bottleneck tails (conv2 3x3 + ReLU + conv3 1x1 + shortcut + ReLU in one launch) where conv2 is 256 wide (res4)
        self.tails = {}
        for name in list(self.wd):
            if name.endswith(".conv2") and self.convs[name][0].shape[0] == 256:
                w3, b3 = self.convs[name[:-1] + "3"]
                if w3.shape[3] == 256 and w3.shape[0] % 256 == 0:
                    self.tails[name[: -len(".conv2")]] = (L.conv_wd_pack_tail(w3.reshape(w3.shape[0], 256).contiguous()), b3)
        # 64-wide stride-1 stages (res2): every block from its 3x3 on, fused with the next block's conv1 (csrc/bneck64.hip)
        self.chains64 = {}
        for name in self.convs:
            if not name.endswith(".0.conv2") or tuple(self.convs[name][0].shape) != (64, 3, 3, 64):
                continue
            stage = name[: -len(".0.conv2")]
            blocks, bi = [], 0
            while f"{stage}.{bi}.conv2" in self.convs:
                blocks.append(f"{stage}.{bi}")
                bi += 1
            ok = all(tuple(self.convs[b + ".conv2"][0].shape) == (64, 3, 3, 64) and tuple(self.convs[b + ".conv3"][0].shape) == (256, 1, 1, 64) and
                     tuple(self.convs[b + ".conv1"][0].shape)[:3] == (64, 1, 1) for b in blocks)
            ok = ok and all(self.convs[b + k][1] is not None for b in blocks for k in (".conv1", ".conv2", ".conv3"))
            ok = ok and all(tuple(self.convs[b + ".conv1"][0].shape) == (64, 1, 1, 256) for b in blocks[1:])
            ok = ok and all((b + ".shortcut") not in self.convs for b in blocks[1:])
            sc0 = self.convs.get(blocks[0] + ".shortcut")
            ok = ok and sc0 is not None and tuple(sc0[0].shape) == (256, 1, 1, 64) and sc0[1] is not None
            if not ok:
                continue
            chain = []
            for i, b in enumerate(blocks):
                nxt = self.convs[blocks[i + 1] + ".conv1"] if i + 1 < len(blocks) else None
                sc = sc0 if i == 0 else None
                packed = L.bneck64_pack(self.convs[b + ".conv2"][0], self.convs[b + ".conv3"][0].reshape(256, 64),
                                        sc[0].reshape(256, 64) if sc else None, nxt[0].reshape(64, 256) if nxt else None)
                chain.append({"packed": packed, "b2": self.convs[b + ".conv2"][1], "b3": self.convs[b + ".conv3"][1],
                              "bsc": sc[1] if sc else None, "b1n": nxt[1] if nxt else None})
            self.chains64[stage] = chain
        # fused StandardRPNHead (3x3 + ReLU + 1x1 in one launch) when the RPN is 256 wide
        self.rpn_head_fused = None
        hw, hb = self.convs["rpn.head"]
        if "rpn.conv" in self.wd and self.convs["rpn.conv"][0].shape[0] == 256 and hw.shape[0] <= 16:
            b16 = torch.zeros(16, dtype=torch.float32, device=self.device)
            b16[: hb.shape[0]] = hb
            self.rpn_head_fused = (L.conv_wd_pack_head(hw.reshape(hw.shape[0], -1).contiguous()), b16)

    def _pack_backbone(self, sd, prefix):
        dev = self.device
        bu = prefix + ".bottom_up"
        w, b = _fold(sd, bu + ".stem.conv1")
        self.convs[bu + ".stem.conv1"] = (_pack_stem(w).to(dev), b.to(dev))
        if w.shape[0] == 64 and w.shape[1] <= 4:  # every reference config (STEM_OUT_CHANNELS 64)
            self.convs[bu + ".stem.fused"] = (pack_stem_fused(w).to(dev), b.to(dev))
        for si, nb in enumerate(STAGE_BLOCKS[self.depth]):
            for bi in range(nb):
                for c in ("conv1", "conv2", "conv3", "shortcut"):
                    name = f"{bu}.res{si + 2}.{bi}.{c}"
                    if name + ".weight" in sd:
                        w, b = _fold(sd, name)
                        self.convs[name] = (_pack_conv(w).to(dev), b.to(dev))
        for i in (2, 3, 4, 5):
            for kind in ("fpn_lateral", "fpn_output"):
                name = f"{prefix}.{kind}{i}"
                self.convs[name] = (_pack_conv(sd[name + ".weight"].float()).to(dev), sd[name + ".bias"].float().to(dev))

    def nbytes(self):
        tot = sum(w.numel() * w.element_size() + b.numel() * 4 for w, b in self.convs.values())
        for w, b in (self.fc1, self.fc2, self.predictor):
            tot += w.numel() * 2 + b.numel() * 4
        return tot
