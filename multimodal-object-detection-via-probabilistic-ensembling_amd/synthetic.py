"""Seeded synthetic workloads (no datasets or checkpoints exist offline).

synth_detections: BASELINE config-3 style per-image detection lists: per detector
n ~ U{0..nmax}, boxes inside 640x512, 30 % cross-detector near-duplicates (jitter sigma 3 px),
probs ~ Dirichlet(1,1,1,0.3)[:K] filtered to max > 0.5, vars ~ U(0.5,3); values round-tripped
through float32 like the JSON written from float32 tensors.
"""
import numpy as np


def _probs(rng, n, K):
    out = np.zeros((0, K))
    while len(out) < n:
        alpha = [1.0] * K + [0.3]
        p = rng.dirichlet(alpha, max(4 * (n - len(out)), 16))[:, :K]
        out = np.concatenate([out, p[p.max(1) > 0.5]])
    return out[:n]


def synth_detections(num_images, seed, kdet=2, nmax=100, K=3, dup_frac=0.3, jitter=3.0, frame=(640.0, 512.0)):
    rng = np.random.default_rng(seed)
    W, H = frame
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)  # noqa: E731
    per_image = []
    for _ in range(num_images):
        infos, base = [], None
        for _d in range(kdet):
            n = int(rng.integers(0, nmax + 1))
            x1 = rng.uniform(0, W - 80, n)
            y1 = rng.uniform(0, H - 72, n)
            bx = np.stack([x1, y1, np.minimum(x1 + rng.uniform(10, 160, n), W),
                           np.minimum(y1 + rng.uniform(10, 160, n), H)], 1).reshape(n, 4)
            if base is not None and len(base) and n:
                dup = rng.random(n) < dup_frac
                src = base[rng.integers(0, len(base), n)]
                bx[dup] = np.clip(src[dup] + rng.normal(0, jitter, (int(dup.sum()), 4)), 0, [W, H, W, H])
            p = _probs(rng, n, K) if n else np.zeros((0, K))
            infos.append({"bbox": f32(bx), "score": f32(p.max(1)) if n else np.zeros(0),
                          "class": p.argmax(1).astype(np.int64) if n else np.zeros(0, np.int64),
                          "prob": f32(p).reshape(n, K), "vars": f32(rng.uniform(0.5, 3.0, (n, 1)))})
            base = bx if base is None else np.concatenate([base, bx])
        per_image.append(infos)
    return per_image


# ------------------------------------------------------------------------------------------------
# Random-init detector weights (no checkpoints exist offline).  Key names follow the reference's
# state dict (SURVEY A.3 / detectron2 module tree) so a real `.pth` drops in unchanged.
# ------------------------------------------------------------------------------------------------
STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}  # backbone/resnet.py:515-519


def synthetic_state_dict(depth=101, num_classes=3, in_channels=3, seed=1, cls_std=0.1, obj_std=0.1, two_backbones=False):
    """Seeded Faster R-CNN R{50,101}-FPN weights: He-normal convs, FrozenBN statistics
    gamma~U(.5,1.5) (x0.6/sqrt(#blocks) on each block's last BN so the residual sums stay O(1)), beta~N(0,.1),
    mean~N(0,.1), var~U(.5,1.5); heads scaled so that objectness / class scores spread enough to
    give ~1000 proposals and tens of detections above 0.5 on random images.
    in_channels 6 (middle fusion): 3-channel backbone(s), 512-channel RPN / box head."""
    import torch
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def randn(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    def uni(n, lo, hi):
        return torch.rand(n, generator=g) * (hi - lo) + lo

    def conv_bn(name, cin, cout, k, gamma_scale=1.0):
        sd[name + ".weight"] = randn(cout, cin, k, k, std=(2.0 / (k * k * cout)) ** 0.5)
        sd[name + ".norm.weight"] = uni(cout, 0.5, 1.5) * gamma_scale
        sd[name + ".norm.bias"] = randn(cout, std=0.1)
        sd[name + ".norm.running_mean"] = randn(cout, std=0.1)
        sd[name + ".norm.running_var"] = uni(cout, 0.5, 1.5)

    def bottom_up(prefix, cin):
        conv_bn(prefix + ".stem.conv1", cin, 64, 7, gamma_scale=0.05)  # inputs are 0..255 minus mean
        c_in = 64
        for si, nb in enumerate(STAGE_BLOCKS[depth]):
            mid, c_out = 64 * 2 ** si, 256 * 2 ** si
            for bi in range(nb):
                p = f"{prefix}.res{si + 2}.{bi}"
                if c_in != c_out:
                    conv_bn(p + ".shortcut", c_in, c_out, 1, gamma_scale=0.7)
                conv_bn(p + ".conv1", c_in, mid, 1)
                conv_bn(p + ".conv2", mid, mid, 3)
                conv_bn(p + ".conv3", mid, c_out, 1, gamma_scale=0.6 / nb ** 0.5)
                c_in = c_out

    def backbone(prefix, cin):
        bottom_up(prefix + ".bottom_up", cin)
        for i, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
            sd[f"{prefix}.fpn_lateral{i}.weight"] = randn(256, c, 1, 1, std=(1.0 / c) ** 0.5)
            sd[f"{prefix}.fpn_lateral{i}.bias"] = randn(256, std=0.01)
            sd[f"{prefix}.fpn_output{i}.weight"] = randn(256, 256, 3, 3, std=(1.0 / (256 * 9)) ** 0.5)
            sd[f"{prefix}.fpn_output{i}.bias"] = randn(256, std=0.01)

    middle = in_channels == 6
    backbone("backbone", 3 if middle else in_channels)
    if middle and two_backbones:
        backbone("backbone_2", 3)
    ch = 512 if middle else 256
    p = "proposal_generator.rpn_head"
    sd[p + ".conv.weight"] = randn(ch, ch, 3, 3, std=(2.0 / (ch * 9)) ** 0.5)
    sd[p + ".conv.bias"] = randn(ch, std=0.01)
    # zero-sum rows: the post-ReLU features have a large common mode that would otherwise make
    # every anchor / ROI score the same; removing it lets scores vary across space, levels, classes
    w = randn(3, ch, 1, 1, std=obj_std)
    sd[p + ".objectness_logits.weight"] = w - w.mean(dim=1, keepdim=True)
    sd[p + ".objectness_logits.bias"] = randn(3, std=0.01)
    w = randn(12, ch, 1, 1, std=0.06)
    sd[p + ".anchor_deltas.weight"] = w - w.mean(dim=1, keepdim=True)
    sd[p + ".anchor_deltas.bias"] = randn(12, std=0.01)
    fin = ch * 49
    sd["roi_heads.box_head.fc1.weight"] = randn(1024, fin, std=(2.0 / fin) ** 0.5)
    sd["roi_heads.box_head.fc1.bias"] = randn(1024, std=0.01)
    sd["roi_heads.box_head.fc2.weight"] = randn(1024, 1024, std=(2.0 / 1024) ** 0.5)
    sd["roi_heads.box_head.fc2.bias"] = randn(1024, std=0.01)
    K = num_classes
    w = randn(K + 1, 1024, std=cls_std)
    sd["roi_heads.box_predictor.cls_score.weight"] = w - w.mean(dim=1, keepdim=True)
    sd["roi_heads.box_predictor.cls_score.bias"] = randn(K + 1, std=0.01)
    w = randn(4 * K, 1024, std=0.02)
    sd["roi_heads.box_predictor.bbox_pred.weight"] = w - w.mean(dim=1, keepdim=True)
    sd["roi_heads.box_predictor.bbox_pred.bias"] = randn(4 * K, std=0.01)
    sd["roi_heads.box_predictor.var_pred.weight"] = randn(1, 1024, std=0.01)
    sd["roi_heads.box_predictor.var_pred.bias"] = randn(1, std=0.01)
    return sd


def synthetic_images(n, height=512, width=640, channels=3, seed=0, structured=True):
    """uint8 HWC frames: uniform noise, optionally with blurred rectangles so detections exist."""
    rng = np.random.default_rng(seed)
    imgs = rng.integers(0, 256, size=(n, height, width, channels), dtype=np.uint8)
    if structured:
        for i in range(n):
            for _ in range(6):
                x0, y0 = int(rng.integers(0, width - 60)), int(rng.integers(0, height - 60))
                w, h = int(rng.integers(30, 200)), int(rng.integers(30, 200))
                imgs[i, y0:y0 + h, x0:x0 + w] = (imgs[i, y0:y0 + h, x0:x0 + w] // 4 + int(rng.integers(0, 192))).astype(np.uint8)
    return imgs


def labelled_frames(n, height=512, width=640, seed=0, max_objects=6):
    """uint8 BGR-order frames with KNOWN objects (for the pseudo-trained-head parity harness, tests/golden/gen_pseudo_heads.py):
    mid-grey noise background and up to `max_objects` non-overlapping rectangles whose appearance encodes their class -
    0 'person': bright and smooth, 1 'bicycle': dark and smooth, 2 'car': 8-pixel checkerboard of both - so that even a
    random-feature box head can be fitted to tell them apart.  Returns (frames [n,H,W,3], list of (boxes [m,4] xyxy float32,
    classes [m] int64))."""
    rng = np.random.default_rng(seed)
    frames = rng.normal(128.0, 30.0, size=(n, height, width, 3)).clip(0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    checker = (((yy // 8) + (xx // 8)) % 2).astype(bool)
    gts = []
    for i in range(n):
        boxes, classes = [], []
        for _ in range(40):
            if len(boxes) >= max_objects:
                break
            w, h = int(rng.integers(48, 220)), int(rng.integers(48, 200))
            x0, y0 = int(rng.integers(4, width - w - 4)), int(rng.integers(4, height - h - 4))
            if any(x0 < b[2] + 12 and b[0] < x0 + w + 12 and y0 < b[3] + 12 and b[1] < y0 + h + 12 for b in boxes):
                continue
            c = int(rng.integers(0, 3))
            patch = frames[i, y0:y0 + h, x0:x0 + w]
            noise = rng.normal(0.0, 6.0, size=patch.shape)
            if c == 0:
                val = 225.0 + noise
            elif c == 1:
                val = 25.0 + noise
            else:
                val = np.where(checker[y0:y0 + h, x0:x0 + w, None], 235.0, 15.0) + noise
            frames[i, y0:y0 + h, x0:x0 + w] = val.clip(0, 255).astype(np.uint8)
            boxes.append([x0, y0, x0 + w, y0 + h])
            classes.append(c)
        gts.append((np.asarray(boxes, dtype=np.float32).reshape(-1, 4), np.asarray(classes, dtype=np.int64)))
    return frames, gts


def labelled_frames_rgb(n, height=512, width=640, seed=0, max_objects=6, hidden=0.2):
    """The 'RGB camera' of the scenes `labelled_frames(n, seed=seed)` shows to the thermal one (fused-mAP harness,
    tests/golden/gen_fused_map.py): the SAME objects (boxes and classes are `labelled_frames`' own), another appearance - tinted
    noise background, class 0 red-ish, class 1 blue-ish, class 2 a yellow / navy checkerboard - and a fraction `hidden` of the
    objects not drawn at all (what a night scene does to the RGB detector), so that the two detectors' lists differ the way
    ProbEn's inputs do: common objects from both, some from one only.  Returns (frames [n,H,W,3] uint8 BGR order, the same ground truth)."""
    _, gts = labelled_frames(n, height, width, seed, max_objects)
    rng = np.random.default_rng(seed + 100003)
    tint = np.array([95.0, 110.0, 120.0])
    frames = (rng.normal(0.0, 35.0, size=(n, height, width, 3)) + tint).clip(0, 255).astype(np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    checker = (((yy // 8) + (xx // 8)) % 2).astype(bool)
    colours = {0: np.array([60.0, 90.0, 225.0]), 1: np.array([215.0, 120.0, 40.0])}
    for i in range(n):
        for b, c in zip(gts[i][0].astype(np.int64), gts[i][1]):
            drawn = rng.random() >= hidden
            noise = rng.normal(0.0, 8.0, size=(b[3] - b[1], b[2] - b[0], 3))
            if not drawn:
                continue
            if int(c) == 2:
                val = np.where(checker[b[1]:b[3], b[0]:b[2], None], np.array([40.0, 230.0, 230.0]), np.array([120.0, 20.0, 20.0])) + noise
            else:
                val = colours[int(c)] + noise
            frames[i, b[1]:b[3], b[0]:b[2]] = val.clip(0, 255).astype(np.uint8)
    return frames, gts
