"""COCO-style bbox evaluation for FLIR (host side).

* `bbox_iou_xywh`   - pycocotools 2.0.4 `_mask.iou` -> `bbIou` for boxes (third-party C, not vendored in the
                      reference; call site detectron2/pycocotools/cocoeval.py:190).  PARITY UNPINNED there;
                      restated from its public definition: inter / (a_d + a_g - inter), crowd GT: inter / a_d.
* `COCOevalBBox`    - the bbox path of the reference's vendored COCOeval
                      (detectron2/pycocotools/cocoeval.py:85-191,236-495, Params :500-536) restated with NumPy:
                      stable sort by -score, <= 100 dets, greedy GT matching per IoU threshold, ignore / crowd /
                      area-range rules, 101-point interpolated precision, the 12 summary stats.
* `instances_to_coco_json`, `FLIREvaluator` - detectron2/evaluation/FLIR_evaluation.py:32-382,496-563:
                      class whitelist {0,1,2,5,7,16} with 5,7 -> 2, XYXY -> XYWH, AP50 = stats[1] * 100,
                      per-class AP.  `evaluate()` gathers over ranks with comm.gather like the reference
                      (:125-131) when `distributed=True`; results are written as JSON instead of pickles.
* `inference_on_dataset` - detectron2/evaluation/evaluator.py:84-168 (5 warm-up iterations, total and
                      pure-compute seconds / image).
"""
import copy
import itertools
import json
import logging
import os
import time
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from . import comm
from .data import MetadataCatalog


def bbox_iou_xywh(dt, gt, iscrowd):
    """dt [D,4], gt [G,4] as (x, y, w, h) float64; iscrowd [G] -> IoU [D,G]."""
    dt = np.asarray(dt, dtype=np.float64).reshape(-1, 4)
    gt = np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    da = dt[:, 2] * dt[:, 3]
    ga = gt[:, 2] * gt[:, 3]
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.where((w <= 0) | (h <= 0), 0.0, w * h)
    union = np.where(np.asarray(iscrowd, dtype=bool)[None, :], da[:, None], da[:, None] + ga[None, :] - inter)
    return inter / union


class COCOevalBBox:
    """gt_json: a COCO dataset dict ("images", "annotations", "categories"); results: list of
    {"image_id", "category_id", "bbox" [x,y,w,h], "score"}."""

    def __init__(self, gt_json, results, impl="native", num_threads=0):
        """impl "native": evaluate + accumulate run in libproben_hip.so's multithreaded host evaluator
        (csrc/cocoeval.cpp, pe_cocoeval_bbox); "numpy": the NumPy restatement below (bit-identical results,
        ~1000x slower; it is the form pinned to the reference by tests/golden/cocoeval_case.json)."""
        assert impl in ("native", "numpy"), impl
        self.impl, self.num_threads = impl, num_threads
        self._gt_json, self._results = gt_json, results
        self.iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.max_dets = [1, 10, 100]
        self.area_rng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.img_ids = sorted({im["id"] for im in gt_json["images"]})
        self.cat_ids = sorted({c["id"] for c in gt_json["categories"]})
        self._gts = defaultdict(list)
        self._dts = defaultdict(list)
        for a in gt_json.get("annotations", []):
            self._gts[a["image_id"], a["category_id"]].append(
                {"bbox": a["bbox"], "area": a.get("area", a["bbox"][2] * a["bbox"][3]), "iscrowd": int(a.get("iscrowd", 0)),
                 "id": a["id"]})
        valid_imgs = set(self.img_ids)
        for i, r in enumerate(results):
            assert r["image_id"] in valid_imgs, "Results do not correspond to current coco set"
            bb = r["bbox"]
            self._dts[r["image_id"], r["category_id"]].append({"bbox": bb, "area": bb[2] * bb[3], "score": r["score"], "id": i + 1})
        self.eval = None
        self.stats = None

    def _eval_img(self, img, cat):
        """Matches of one (image, category) for every area range at maxDet = 100."""
        gt, dt = self._gts.get((img, cat), []), self._dts.get((img, cat), [])
        if not gt and not dt:
            return None
        order = np.argsort([-d["score"] for d in dt], kind="mergesort")[: self.max_dets[-1]]
        dt = [dt[i] for i in order]
        ious_all = bbox_iou_xywh([d["bbox"] for d in dt], [g["bbox"] for g in gt], [g["iscrowd"] for g in gt])
        out = []
        T = len(self.iou_thrs)
        for lo, hi in self.area_rng:
            g_ignore = np.array([bool(g["iscrowd"]) or g["area"] < lo or g["area"] > hi for g in gt], dtype=bool)
            gtind = np.argsort(g_ignore.astype(np.int64), kind="mergesort")
            g_ig = g_ignore[gtind]
            g_crowd = np.array([gt[i]["iscrowd"] for i in gtind], dtype=bool)
            ious = ious_all[:, gtind] if len(gt) and len(dt) else ious_all
            gtm = np.zeros((T, len(gt)), dtype=np.int64)
            dtm = np.zeros((T, len(dt)), dtype=np.int64)
            dt_ig = np.zeros((T, len(dt)), dtype=bool)
            if len(gt) and len(dt):
                for ti, thr in enumerate(self.iou_thrs):
                    for di in range(len(dt)):
                        best = min(thr, 1 - 1e-10)
                        m = -1
                        for gi in range(len(gt)):
                            if gtm[ti, gi] > 0 and not g_crowd[gi]:
                                continue
                            if m > -1 and not g_ig[m] and g_ig[gi]:
                                break
                            if ious[di, gi] < best:
                                continue
                            best = ious[di, gi]
                            m = gi
                        if m == -1:
                            continue
                        dt_ig[ti, di] = g_ig[m]
                        dtm[ti, di] = gt[gtind[m]]["id"]
                        gtm[ti, m] = dt[di]["id"]
            d_out = np.array([d["area"] < lo or d["area"] > hi for d in dt], dtype=bool).reshape(1, -1)
            dt_ig = dt_ig | ((dtm == 0) & np.repeat(d_out, T, 0))
            out.append({"dt_scores": np.array([d["score"] for d in dt]), "dtm": dtm, "dt_ig": dt_ig, "gt_ig": g_ig})
        return out

    def evaluate(self):
        if self.impl == "native":
            self._per = None  # matching happens inside pe_cocoeval_bbox (accumulate)
            return
        self._per = {(i, c): self._eval_img(i, c) for c in self.cat_ids for i in self.img_ids}

    def _accumulate_native(self):
        import ctypes
        from . import _lib
        T, R, K, A, M = len(self.iou_thrs), len(self.rec_thrs), len(self.cat_ids), len(self.area_rng), len(self.max_dets)
        img_ix = {v: i for i, v in enumerate(self.img_ids)}
        cat_ix = {v: i for i, v in enumerate(self.cat_ids)}
        anns = [a for a in self._gt_json.get("annotations", []) if a["category_id"] in cat_ix and a["image_id"] in img_ix]
        res = [r for r in self._results if r["category_id"] in cat_ix]  # other categories are never looked up
        gt_img = np.array([img_ix[a["image_id"]] for a in anns], dtype=np.int32)
        gt_cat = np.array([cat_ix[a["category_id"]] for a in anns], dtype=np.int32)
        gt_box = np.array([a["bbox"] for a in anns], dtype=np.float64).reshape(-1, 4)
        gt_area = np.array([a.get("area", a["bbox"][2] * a["bbox"][3]) for a in anns], dtype=np.float64)
        gt_crowd = np.array([int(a.get("iscrowd", 0)) != 0 for a in anns], dtype=np.uint8)
        gt_id = np.array([a["id"] for a in anns], dtype=np.int64)
        dt_img = np.array([img_ix[r["image_id"]] for r in res], dtype=np.int32)
        dt_cat = np.array([cat_ix[r["category_id"]] for r in res], dtype=np.int32)
        dt_box = np.array([r["bbox"] for r in res], dtype=np.float64).reshape(-1, 4)
        dt_score = np.array([r["score"] for r in res], dtype=np.float64)
        precision = np.empty((T, R, K, A, M), dtype=np.float64)
        recall = np.empty((T, K, A, M), dtype=np.float64)
        iou = np.ascontiguousarray(self.iou_thrs, dtype=np.float64)
        rec = np.ascontiguousarray(self.rec_thrs, dtype=np.float64)
        md = np.array(self.max_dets, dtype=np.int32)
        ar = np.array(self.area_rng, dtype=np.float64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a.size else None
        st = _lib.lib().pe_cocoeval_bbox(p(gt_img), p(gt_cat), p(gt_box), p(gt_area), p(gt_crowd), p(gt_id), len(anns),
                                         p(dt_img), p(dt_cat), p(dt_box), p(dt_score), len(res), len(self.img_ids), K,
                                         p(iou), T, p(rec), R, p(md), M, p(ar), A, int(self.num_threads), p(precision),
                                         p(recall))
        _lib.check(st, "pe_cocoeval_bbox")
        self.eval = {"precision": precision, "recall": recall, "counts": [T, R, K, A, M]}

    def accumulate(self):
        if self.impl == "native":
            return self._accumulate_native()
        T, R, K, A, M = len(self.iou_thrs), len(self.rec_thrs), len(self.cat_ids), len(self.area_rng), len(self.max_dets)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        for k, cat in enumerate(self.cat_ids):
            per = [self._per[i, cat] for i in self.img_ids]
            per = [e for e in per if e is not None]
            if not per:
                continue
            for a in range(A):
                for m, max_det in enumerate(self.max_dets):
                    scores = np.concatenate([e[a]["dt_scores"][:max_det] for e in per])
                    inds = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e[a]["dtm"][:, :max_det] for e in per], axis=1)[:, inds]
                    dt_ig = np.concatenate([e[a]["dt_ig"][:, :max_det] for e in per], axis=1)[:, inds]
                    gt_ig = np.concatenate([e[a]["gt_ig"] for e in per])
                    npig = np.count_nonzero(~gt_ig)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=np.float64)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=np.float64)
                    for t in range(T):
                        tp, fp = tp_sum[t], fp_sum[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        q = np.zeros((R,))
                        idx = np.searchsorted(rc, self.rec_thrs, side="left")
                        for ri, pi in enumerate(idx):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
        self.eval = {"precision": precision, "recall": recall, "counts": [T, R, K, A, M]}

    def _summ(self, ap, iou=None, area=0, max_det=2):
        s = self.eval["precision"] if ap else self.eval["recall"]
        if iou is not None:
            t = np.where(np.isclose(self.iou_thrs, iou))[0]
            s = s[t]
        s = s[:, :, :, area, max_det] if ap else s[:, :, area, max_det]
        s = s[s > -1]
        return -1.0 if s.size == 0 else float(np.mean(s))

    def summarize(self, printer=print):
        st = [self._summ(1), self._summ(1, .5), self._summ(1, .75), self._summ(1, area=1), self._summ(1, area=2),
              self._summ(1, area=3), self._summ(0, max_det=0), self._summ(0, max_det=1), self._summ(0, max_det=2),
              self._summ(0, area=1), self._summ(0, area=2), self._summ(0, area=3)]
        self.stats = np.array(st)
        if printer:
            rows = [("Precision", "AP", "0.50:0.95", "all", 100), ("Precision", "AP", "0.50", "all", 100),
                    ("Precision", "AP", "0.75", "all", 100), ("Precision", "AP", "0.50:0.95", "small", 100),
                    ("Precision", "AP", "0.50:0.95", "medium", 100), ("Precision", "AP", "0.50:0.95", "large", 100),
                    ("Recall", "AR", "0.50:0.95", "all", 1), ("Recall", "AR", "0.50:0.95", "all", 10),
                    ("Recall", "AR", "0.50:0.95", "all", 100), ("Recall", "AR", "0.50:0.95", "small", 100),
                    ("Recall", "AR", "0.50:0.95", "medium", 100), ("Recall", "AR", "0.50:0.95", "large", 100)]
            for (title, typ, iou, area, md), v in zip(rows, st):
                printer(" Average {:<9} ({}) @[ IoU={:<9} | area={:>6s} | maxDets={:>3d} ] = {:0.3f}".format(title, typ, iou, area, md, v))
        return self.stats


VALID_CLASSES = (0, 1, 2, 5, 7, 16)


def instances_to_coco_json(instances, img_id):
    """FLIR_evaluation.py:313-382 (bbox fields only)."""
    n = len(instances)
    if n == 0:
        return []
    b = instances.pred_boxes.tensor.detach().cpu().numpy().copy()
    b[:, 2] -= b[:, 0]
    b[:, 3] -= b[:, 1]
    boxes = b.tolist()
    scores = instances.scores.tolist()
    classes = [int(c) for c in instances.pred_classes.tolist()]
    out = []
    for k in range(n):
        if classes[k] in VALID_CLASSES:
            out.append({"image_id": img_id, "category_id": 2 if classes[k] in (5, 7) else classes[k],
                        "bbox": boxes[k], "score": scores[k]})
    return out


class FLIREvaluator:
    """Same constructor / reset / process / evaluate contract as the reference's FLIREvaluator."""

    def __init__(self, dataset_name, cfg, distributed, output_dir=None, out_pr_name=None, save_eval=False, out_eval_path=None):
        self._distributed = distributed
        self._output_dir = output_dir
        self._save_eval = save_eval
        self._out_eval_path = out_eval_path
        self._logger = logging.getLogger(__name__)
        self._metadata = MetadataCatalog.get(dataset_name)
        with open(self._metadata.json_file) as f:
            self._gt = json.load(f)
        self._do_evaluation = "annotations" in self._gt
        self.reset()

    def reset(self):
        self._predictions = []
        self._coco_results = []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            pred = {"image_id": inp["image_id"]}
            if "instances" in out:
                pred["instances"] = instances_to_coco_json(out["instances"].to("cpu"), inp["image_id"])
            self._predictions.append(pred)

    def process_rows(self, rows):
        """Fast path: rows [n,7] = (image_id, x, y, w, h, score, category_id) already filtered / remapped
        (the tensor form that travels through comm.all_gather_rows)."""
        for r in np.asarray(rows, dtype=np.float64):
            self._predictions.append({"image_id": int(r[0]), "instances": [
                {"image_id": int(r[0]), "category_id": int(r[6]), "bbox": [r[1], r[2], r[3], r[4]], "score": float(r[5])}]})

    def evaluate(self, out_eval_path=""):
        if self._distributed and comm.is_distributed():
            # The reference gathers pickled per-image dict lists to rank 0 (FLIR_evaluation.py:124-131, comm.gather over gloo).
            # Here a rank's predictions are [n, 7] float64 rows (image_id, x, y, w, h, score, category_id) - exact images of
            # the float32 detector outputs - and ONE padded all-gather on the rank's device moves them (RCCL over xGMI);
            # InferenceSampler shards are contiguous, so rank order IS dataset order and the table rank 0 evaluates is the
            # single-process table, bit for bit.
            comm.synchronize()
            flat = [r for p in self._predictions for r in p.get("instances", [])]
            # image ids travel as float64: exact for integers below 2^53.  Anything else (string ids, huge ids - the reference pickles
            # arbitrary dicts, so they worked there) takes the pickled gloo gather instead; the choice is made by ALL ranks together.
            numeric = all(isinstance(r["image_id"], (int, np.integer)) and not isinstance(r["image_id"], bool) and abs(int(r["image_id"])) < 2 ** 53
                          for r in flat)
            numeric = all(comm.all_gather(bool(numeric)))
            if numeric:
                mine = [[r["image_id"], *r["bbox"], r["score"], r["category_id"]] for r in flat]
                rows = comm.gather_rows(torch.tensor(mine, dtype=torch.float64).reshape(-1, 7))
                seen = comm.gather_rows(torch.tensor([[float(len(self._predictions))]], dtype=torch.float64))
                if not comm.is_main_process():
                    return {}
                coco = [{"image_id": int(r[0]), "category_id": int(r[6]), "bbox": [r[1], r[2], r[3], r[4]], "score": r[5]} for r in rows.tolist()]
                n_seen = int(seen.sum().item())
            else:
                parts = comm.gather((flat, len(self._predictions)), dst=0)       # FLIR_evaluation.py:124-131
                if not comm.is_main_process():
                    return {}
                coco = [r for part, _ in parts for r in part]
                n_seen = sum(n for _, n in parts)
            self._predictions = [{"image_id": -1, "instances": coco}] if n_seen > 0 else []
        if len(self._predictions) == 0:
            self._logger.warning("[FLIREvaluator] Did not receive valid predictions.")
            return {}
        self._results = OrderedDict()
        self._coco_results = list(itertools.chain(*[x["instances"] for x in self._predictions]))
        rev = getattr(self._metadata, "thing_dataset_id_to_contiguous_id", None)
        if rev:
            rev = {v: k for k, v in rev.items()}
            for r in self._coco_results:
                assert r["category_id"] in rev, "A prediction has category_id={}, which is not available in the dataset.".format(r["category_id"])
                r["category_id"] = rev[r["category_id"]]
        if self._output_dir:
            os.makedirs(self._output_dir, exist_ok=True)
            with open(os.path.join(self._output_dir, "coco_instances_results.json"), "w") as f:
                f.write(json.dumps(self._coco_results))
        if not self._do_evaluation:
            return {}
        names = getattr(self._metadata, "thing_classes", None)
        if len(self._coco_results) == 0:
            self._results["bbox"] = {m: -1 for m in ["AP", "AP50", "AP75", "APs", "APm", "APl"]}
            return copy.deepcopy(self._results)
        ev = COCOevalBBox(self._gt, self._coco_results)
        ev.evaluate()
        ev.accumulate()
        ev.summarize()
        self._last_eval = ev
        res = {m: float(ev.stats[i] * 100) for i, m in enumerate(["AP", "AP50", "AP75", "APs", "APm", "APl"])}
        if names is not None and len(names) > 1:
            prec = ev.eval["precision"]
            assert len(names) == prec.shape[2]
            for idx, name in enumerate(names):
                p = prec[:, :, idx, 0, -1]
                p = p[p > -1]
                res["AP-" + name] = float(np.mean(p) * 100) if p.size else float("nan")
        self._results["bbox"] = res
        path = out_eval_path or (self._out_eval_path if self._save_eval else "")
        if path:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            with open(path, "w") as f:
                json.dump({"stats": ev.stats.tolist(), "results": res}, f)
        return copy.deepcopy(self._results)


def inference_on_dataset(model, data_loader, evaluator):
    """evaluator.py:84-168: run `model` over the loader, feed the evaluator, log total / pure-compute s/img."""
    logger = logging.getLogger(__name__)
    evaluator.reset()
    total = len(data_loader)
    num_warmup = min(5, total - 1)
    start = time.perf_counter()
    compute = 0.0
    with torch.no_grad():
        for idx, inputs in enumerate(data_loader):
            if idx == num_warmup:
                start = time.perf_counter()
                compute = 0.0
            t0 = time.perf_counter()
            outputs = model(inputs)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            compute += time.perf_counter() - t0
            evaluator.process(inputs, outputs)
    tot = time.perf_counter() - start
    n = max(total - num_warmup, 1)
    logger.info("Total inference time: %.3f s (%.6f s / img per device, on %d devices)", tot, tot / n, comm.get_world_size())
    logger.info("Total inference pure compute time: %.3f s (%.6f s / img per device)", compute, compute / n)
    results = evaluator.evaluate()
    return results if results is not None else {}
