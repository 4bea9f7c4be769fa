"""Build-container generator: "pseudo-trained" box-predictor weights + the ORACLE's detections with them.

Why: north_star's contract is "demo_mAP_FLIR reproduces the reference mAP within 1e-3"; neither FLIR nor its checkpoints exist
offline, and with random-init weights hundreds of candidates sit on the 0.5 threshold, so HIP-vs-oracle agreement could only be
quoted as a matching rate.  Here the scores are made to separate the way a trained model's do:
  * backbone, FPN, RPN, fc1, fc2 stay the seeded random-init weights of proben_amd.synthetic.synthetic_state_dict (R101, seed 1);
  * stage A, on TRAIN frames with known objects (synthetic.labelled_frames): the RPN head's two 1x1 layers are fitted on the
    oracle's 256-channel RPN feature `relu(conv3x3(p_l))` - objectness by logistic regression per anchor shape (positive: anchor IoU
    with an object >= 0.6 or the best anchor of an object, negative: < 0.3), anchor deltas by ridge regression on the positives -
    because a random RPN puts ~1 of its 1000 proposals on an object and nothing could be fitted behind it;
  * stage B: with that RPN the oracle (= restatement of the reference's CPU path) is run up to the fc2 features of its 1000
    proposals per frame; `cls_score` is fitted by multinomial logistic regression (label = class of the ground-truth box a
    proposal overlaps with IoU >= 0.5, else background) and `bbox_pred` by ridge regression on the reference's box deltas
    (weights 10, 10, 5, 5) of the foreground proposals;
  * on held-out EVAL frames the oracle runs end to end with the fitted heads; its detections are the fixture.
Outputs (tests/golden/pseudo_heads_r101.npz): the eight fitted tensors, the oracle's eval detections (frame, box, score, class) and
its AP table against the known objects.  tests/test_parity_map_gpu.py runs the HIP detector with the same weights on the same
frames and compares the two AP figures against the SAME ground truth.

    python tests/golden/gen_pseudo_heads.py            (~10 minutes on 8 cores)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import proben_amd  # noqa: E402,F401
from oracle import detector as D  # noqa: E402
from proben_amd import evaluation  # noqa: E402
from proben_amd.data import resize_shortest_edge_shape  # noqa: E402
from proben_amd.synthetic import labelled_frames, synthetic_state_dict  # noqa: E402

DEPTH, SEED, N_TRAIN, N_EVAL = 101, 1, 48, 256
TRAIN_SEED, EVAL_SEED = 7001, 7002
OUT = os.path.join(ROOT, "tests", "golden", "pseudo_heads_r101.npz")


def to_oracle_input(frame, new_hw):
    from PIL import Image
    r = np.array(Image.fromarray(frame).resize((new_hw[1], new_hw[0]), Image.BILINEAR))     # the reference's 3-channel resize
    return torch.from_numpy(r).permute(2, 0, 1).float().contiguous()


def box_iou(a, b):
    x1 = torch.max(a[:, None, 0], b[None, :, 0]); y1 = torch.max(a[:, None, 1], b[None, :, 1])
    x2 = torch.min(a[:, None, 2], b[None, :, 2]); y2 = torch.min(a[:, None, 3], b[None, :, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter)


def get_deltas(src, dst, w=(10.0, 10.0, 5.0, 5.0)):
    """Box2BoxTransform.get_deltas (modeling/box_regression.py:41-71)."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    dw, dh = dst[:, 2] - dst[:, 0], dst[:, 3] - dst[:, 1]
    dx, dy = dst[:, 0] + 0.5 * dw, dst[:, 1] + 0.5 * dh
    return torch.stack([w[0] * (dx - sx) / sw, w[1] * (dy - sy) / sh, w[2] * torch.log(dw / sw), w[3] * torch.log(dh / sh)], dim=1)


def coco_tables(frames_gt, dets, hw=(512, 640)):
    images = [{"id": i, "height": hw[0], "width": hw[1], "file_name": f"{i}.jpeg"} for i in range(len(frames_gt))]
    cats = [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"}, {"id": 3, "name": "car"}]
    anns, aid = [], 1
    for i, (b, c) in enumerate(frames_gt):
        for bb, cc in zip(b, c):
            w, h = float(bb[2] - bb[0]), float(bb[3] - bb[1])
            anns.append({"id": aid, "image_id": i, "category_id": int(cc) + 1, "bbox": [float(bb[0]), float(bb[1]), w, h], "area": w * h, "iscrowd": 0})
            aid += 1
    rows = [{"image_id": int(r[0]), "category_id": int(r[6]) + 1, "bbox": [float(r[1]), float(r[2]), float(r[3] - r[1]), float(r[4] - r[2])], "score": float(r[5])}
            for r in dets]
    ev = evaluation.COCOevalBBox({"images": images, "annotations": anns, "categories": cats}, rows, impl="native")
    ev.evaluate()
    ev.accumulate()
    return np.asarray(ev.summarize(printer=None), dtype=np.float64)


def main(out=OUT, seed=SEED, frames_fn=labelled_frames, n_eval=N_EVAL):
    """Defaults = the committed thermal-like fixture.  tests/golden/gen_fused_map.py calls it with another weight seed and the RGB-like
    rendering of the same scenes (`frames_fn`) for the second detector of the fused-mAP harness (n_eval = 0: heads only)."""
    torch.set_num_threads(os.cpu_count() or 8)
    sd = synthetic_state_dict(DEPTH, 3, 3, seed=seed)
    spec = D.DetectorSpec(depth=DEPTH)
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    sx, sy = new_hw[1] / 640.0, new_hw[0] / 512.0
    frames, gts = frames_fn(N_TRAIN, seed=TRAIN_SEED)
    scale4 = torch.tensor([sx, sy, sx, sy])
    # ---------------- stage A: RPN objectness / anchor deltas on the oracle's RPN features ----------------
    pre = "proposal_generator.rpn_head"
    rng = np.random.default_rng(11)
    FA, LA, TA, AA = [], [], [], []          # feature [256], label {0,1}, delta target [4], anchor shape id
    t0 = time.time()
    for i in range(N_TRAIN):
        batch, sizes = D.preprocess([to_oracle_input(frames[i], new_hw)], spec)
        feats = D.backbone_features(batch, sd, spec)
        gb = torch.from_numpy(gts[i][0]) * scale4
        for lvl, (k, stride) in enumerate(zip(["p2", "p3", "p4", "p5", "p6"], [4, 8, 16, 32, 64])):
            t = torch.relu(torch.nn.functional.conv2d(feats[k], sd[pre + ".conv.weight"], sd[pre + ".conv.bias"], padding=1))[0]   # [256,H,W]
            H, W = t.shape[1:]
            anchors = D.grid_anchors((H, W), stride, D.cell_anchors(spec.anchor_sizes[lvl], spec.aspect_ratios))                     # [(H*W*3), 4], cell-major
            tf = t.permute(1, 2, 0).reshape(H * W, 256)
            if len(gb) == 0:
                continue
            iou = box_iou(anchors, gb)
            best, arg = iou.max(dim=1)
            pos = best >= 0.6
            top_per_gt = iou.max(dim=0).values
            pos |= ((iou == top_per_gt[None, :]) & (top_per_gt[None, :] >= 0.3)).any(dim=1)
            neg = torch.nonzero((best < 0.3) & ~pos).squeeze(1)
            neg = neg[torch.from_numpy(rng.permutation(len(neg))[:600])]
            pidx = torch.nonzero(pos).squeeze(1)
            for idx, lab in ((pidx, 1), (neg, 0)):
                if len(idx) == 0:
                    continue
                FA.append(tf[idx // 3]); AA.append(idx % 3)
                LA.append(torch.full((len(idx),), lab, dtype=torch.int64))
                TA.append(get_deltas(anchors[idx], gb[arg[idx]], (1.0, 1.0, 1.0, 1.0)))
        if i % 8 == 0:
            print(f"stage A frame {i}: {sum(int(l.sum()) for l in LA)} positive anchors so far, {time.time() - t0:.0f} s", flush=True)
    FA, LA, TA, AA = torch.cat(FA), torch.cat(LA), torch.cat(TA), torch.cat(AA)
    mu_a, sig_a = FA.mean(0), FA.std(0) + 1e-3
    Fn = ((FA - mu_a) / sig_a).double()
    w_obj, b_obj = torch.zeros((3, 256)), torch.zeros((3,))
    w_dlt, b_dlt = torch.zeros((12, 256)), torch.zeros((12,))
    for a in range(3):
        m = AA == a
        Xa, ya = Fn[m], LA[m].double()
        w = torch.zeros((256,), dtype=torch.float64, requires_grad=True)
        b0 = torch.zeros((1,), dtype=torch.float64, requires_grad=True)
        pw = (1 - ya).sum() / ya.sum().clamp(min=1)
        opt = torch.optim.LBFGS([w, b0], lr=1.0, max_iter=200, history_size=20, line_search_fn="strong_wolfe")

        def closure():
            opt.zero_grad()
            loss = torch.nn.functional.binary_cross_entropy_with_logits(Xa @ w + b0, ya, pos_weight=pw) + 1e-3 * (w * w).sum()
            loss.backward()
            return loss
        opt.step(closure)
        with torch.no_grad():
            acc = (((Xa @ w + b0) > 0).double() == ya).double().mean()
            w_obj[a] = (w / sig_a.double()).float()
            b_obj[a] = (b0 - (w * (mu_a / sig_a).double()).sum()).float()
            mp = m & (LA == 1)
            A = torch.cat([Fn[mp], torch.ones((int(mp.sum()), 1), dtype=torch.float64)], dim=1)
            lam = 20.0 * torch.eye(257, dtype=torch.float64); lam[-1, -1] = 0.0
            sol = torch.linalg.solve(A.t() @ A + lam, A.t() @ TA[mp].double())
            w_dlt[4 * a:4 * a + 4] = (sol[:-1].t() / sig_a.double()).float()
            b_dlt[4 * a:4 * a + 4] = (sol[-1] - (sol[:-1].t() * (mu_a / sig_a).double()).sum(1)).float()
        print(f"RPN anchor shape {a}: {int(ya.sum())} positives / {len(ya)} samples, objectness train accuracy {float(acc):.4f}")
    rpn = {pre + ".objectness_logits.weight": w_obj.view(3, 256, 1, 1).contiguous(), pre + ".objectness_logits.bias": b_obj.contiguous(),
           pre + ".anchor_deltas.weight": w_dlt.view(12, 256, 1, 1).contiguous(), pre + ".anchor_deltas.bias": b_dlt.contiguous()}
    sd.update(rpn)
    # ---------------- stage B: features of the oracle's proposals on the training frames ----------------
    X, Y, T = [], [], []
    t0 = time.time()
    for i in range(N_TRAIN):
        _, inter = D.forward([to_oracle_input(frames[i], new_hw)], sd, spec, out_sizes=[(512, 640)], return_intermediates=True)
        props = inter["proposals"][0][0]                                   # [P,4] in the resized frame
        f = torch.flatten(inter["pooled"], start_dim=1)
        f = torch.relu(torch.nn.functional.linear(f, sd["roi_heads.box_head.fc1.weight"], sd["roi_heads.box_head.fc1.bias"]))
        f = torch.relu(torch.nn.functional.linear(f, sd["roi_heads.box_head.fc2.weight"], sd["roi_heads.box_head.fc2.bias"]))
        gb = torch.from_numpy(gts[i][0]) * scale4
        gc = torch.from_numpy(gts[i][1])
        if len(gb):
            iou = box_iou(props, gb)
            best, arg = iou.max(dim=1)
            lab = torch.where(best >= 0.5, gc[arg], torch.full_like(arg, 3))
            lab = torch.where((best < 0.5) & (best >= 0.35), torch.full_like(lab, -1), lab)      # ambiguous band: not used
            tgt = get_deltas(props, gb[arg])
        else:
            lab, tgt = torch.full((len(props),), 3, dtype=torch.int64), torch.zeros((len(props), 4))
        X.append(f); Y.append(lab); T.append(tgt)
        if i % 8 == 0:
            print(f"stage B frame {i}: {int((lab < 3).sum() - (lab < 0).sum())} foreground of {len(props)} proposals, {time.time() - t0:.0f} s", flush=True)
    X, Y, T = torch.cat(X), torch.cat(Y), torch.cat(T)
    use = Y >= 0
    X, Y, T = X[use], Y[use], T[use]
    mu, sig = X.mean(0), X.std(0) + 1e-3
    Xn = ((X - mu) / sig).double()
    # ---------------- cls_score: multinomial logistic regression (class-balanced, L2) ----------------
    K1 = 4
    W = torch.zeros((K1, Xn.shape[1]), dtype=torch.float64, requires_grad=True)
    b = torch.zeros((K1,), dtype=torch.float64, requires_grad=True)
    cw = torch.tensor([1.0 / max(int((Y == k).sum()), 1) for k in range(K1)], dtype=torch.float64)
    cw = cw / cw.sum() * K1
    opt = torch.optim.LBFGS([W, b], lr=1.0, max_iter=300, history_size=20, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(Xn @ W.t() + b, Y, weight=cw) + 2e-4 * (W * W).sum()
        loss.backward()
        return loss
    opt.step(closure)
    with torch.no_grad():
        acc = ((Xn @ W.t() + b).argmax(1) == Y).double().mean()
        Wc = (W / sig.double()).float()                     # fold the standardisation into the layer
        bc = (b - (W * (mu / sig).double()).sum(1)).float()
        Wc = Wc * 1.0
    print("cls_score fitted: train accuracy %.4f, class counts %s" % (float(acc), [int((Y == k).sum()) for k in range(K1)]))
    # ---------------- bbox_pred: ridge regression of the deltas, per class ----------------
    Wb = torch.zeros((12, X.shape[1])); bb = torch.zeros((12,))
    for k in range(3):
        m = Y == k
        A = torch.cat([Xn[m], torch.ones((int(m.sum()), 1), dtype=torch.float64)], dim=1)
        lam = 30.0 * torch.eye(A.shape[1], dtype=torch.float64); lam[-1, -1] = 0.0
        sol = torch.linalg.solve(A.t() @ A + lam, A.t() @ T[m].double())          # [1025, 4]
        w_k = sol[:-1].t() / sig.double()
        Wb[4 * k:4 * k + 4] = w_k.float()
        bb[4 * k:4 * k + 4] = (sol[-1] - (sol[:-1].t() * (mu / sig).double()).sum(1)).float()
        res = (A @ sol - T[m].double()).abs().mean()
        print(f"bbox_pred class {k}: {int(m.sum())} samples, mean |delta residual| {float(res):.3f}")
    heads = {"roi_heads.box_predictor.cls_score.weight": Wc.contiguous(), "roi_heads.box_predictor.cls_score.bias": bc.contiguous(),
             "roi_heads.box_predictor.bbox_pred.weight": Wb.contiguous(), "roi_heads.box_predictor.bbox_pred.bias": bb.contiguous()}
    sd.update(heads)
    heads.update(rpn)
    # ---------------- the oracle end to end on the held-out frames ----------------
    if n_eval == 0:
        np.savez_compressed(out, depth=DEPTH, seed=seed, **{k.replace(".", "/"): v.numpy() for k, v in heads.items()})
        print("wrote", out, os.path.getsize(out), "bytes")
        return
    eframes, egts = frames_fn(n_eval, seed=EVAL_SEED)
    rows = []
    t0 = time.time()
    for i in range(n_eval):
        o = D.forward([to_oracle_input(eframes[i], new_hw)], sd, spec, out_sizes=[(512, 640)])[0]
        for bx, s, c in zip(o["boxes"].numpy(), o["scores"].numpy(), o["classes"].numpy()):
            rows.append([i, bx[0], bx[1], bx[2], bx[3], s, c])
        if i % 8 == 0:
            print(f"eval frame {i}: {len(o['scores'])} detections, {time.time() - t0:.0f} s", flush=True)
    rows = np.asarray(rows, dtype=np.float32).reshape(-1, 7)
    stats = coco_tables(egts, rows)
    print("oracle vs known objects: AP %.3f AP50 %.3f AP75 %.3f over %d detections" % (stats[0] * 100, stats[1] * 100, stats[2] * 100, len(rows)))
    np.savez_compressed(out, depth=DEPTH, seed=seed, eval_seed=EVAL_SEED, n_eval=n_eval, oracle_rows=rows, oracle_stats=stats,
                        **{k.replace(".", "/"): v.numpy() for k, v in heads.items()})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
