"""KAIST multispectral pedestrian benchmark: log-average miss rate, behind the call the reference makes.

    from proben_amd.evalKAIST.evaluation_script import evaluate
    result = evaluate('KAIST_annotation.json', 'KAIST_thermal_only_result.txt', 'Multispectral')
    result['all'].summarize(0); result['day'].summarize(0); result['night'].summarize(0)
    recall_all = 1 - result['all'].eval['yy'][0][-1]

is exactly what demo/KAIST/demo_LAMR_KAIST.py:85,145 and demo/KAIST/demo_train_KAIST.py:9,116-121 bind.

PARITY UNPINNED.  The package `evalKAIST` is NOT in the reference's tree ("missing"), and the way the reference uses
the return value (`.summarize(0)`, `.eval['yy']`) identifies it as the evaluation script distributed with the KAIST benchmark's Python
tooling (MLPD-Multi-Label-Pedestrian-Detection, `evaluation_script/evaluation_script.py`, Kim et al. 2021 - a pycocotools `COCOeval`
subclass; no version is pinned by the reference).  This file restates that script's PUBLISHED protocol from its description, with its own
data structures (flat NumPy tables instead of per-annotation dicts); every rule below is pinned by a hand-worked case in
tests/test_kaist_eval_cpu.py, none by an execution of the third-party script (it is not in this container).

The protocol (`id_setup` 0 = "Reasonable", the one the reference asks for):
  * annotation file: COCO-style `{"images": [{"id", "im_name", ...}], "annotations": [{"image_id", "category_id", "bbox": [x, y, w, h],
    "height", "occlusion", "ignore", "id"}], "categories": [...]}`; only category 1 (person) is evaluated;
  * a ground-truth box is IGNORED (neither a miss nor a target) when its own `ignore` flag is set, its height is outside
    [55, inf), its occlusion is not 0 (none) or 1 (partial), or it leaves the window x >= 5, y >= 5, x + w <= 635, y + h <= 507;
  * result file: text rows `<1-based frame index>,x,y,w,h,score` (image_id = index - 1) or a COCO result json; per image the
    detections are taken in descending score order (stable), at most 1000, and those whose height is outside
    [55 / 1.25, inf * 1.25) are dropped before matching;
  * overlap = intersection / union, but intersection / DETECTION area against an ignored ground-truth box;
  * matching at overlap >= 0.5 (strictly: >= min(0.5, 1 - 1e-10)), greedy in score order: ground truth sorted real-first / ignored-last; a
    detection takes the best-overlapping still-unmatched real box; once it holds any match the scan stops at the first ignored box (so a
    real match is never traded for an ignored one, and among ignored boxes the first one that overlaps enough is kept); ignored boxes can
    absorb any number of detections; a detection matched to an ignored box is neither a true nor a false positive;
  * over all images, the remaining detections in descending score order (stable) give cumulative TP / FP; miss rate = 1 - TP / (number
    of non-ignored ground truth), FPPI = FP / number of images; the miss rate is read at the 9 FPPI reference points
    10^(-2 : 0.25 : 0) (the last operating point with FPPI <= the reference; when there is none the script's negative index reads the LAST
    operating point - kept) and averaged in log space (`MR = exp(mean(log(mr)))`; a 0 among them makes the result 0, as in the script);
  * `all` = every image, `day` = the first 1455 image ids in sorted order, `night` = the rest (the test-all-20 split: 2252 frames).

The published script's row arithmetic is REPRODUCED by default (reproduce the quirk, offer the switch): it computes the
overlap matrix of an image over ALL its detections in descending score order, then drops the detections outside the expanded height
range and looks a kept detection's row up as `id - id of the first kept detection` (ids = 1-based positions in the result file).  That is
the detection's own row only while the file lists an image's rows by descending score and no higher-scoring detection of the image was
dropped; otherwise a kept detection is matched with ANOTHER detection's overlaps (a negative difference wraps around like NumPy's
indexing, a difference beyond the image's detections is an IndexError here as there).  `fix_row_index=True` (`KAISTParams.fixRowIndex`,
`evaluate(..., fix_row_index=True)`, `demo_LAMR_KAIST --fix-row-index`) uses each kept detection's own row - what rounds 4-5 shipped as
the only behaviour.  tests/test_kaist_eval_cpu.py::test_row_index_quirk_and_its_fix works a case by hand where the two differ.
"""
import copy
import json
import os

import numpy as np


class KAISTParams:
    """Evaluation parameters (the script's `KAISTParams`)."""

    def __init__(self):
        self.imgIds = []
        self.catIds = [1]
        self.iouThrs = np.array([0.5])
        self.maxDets = [1000]
        self.fppiThrs = np.array([0.0100, 0.0178, 0.0316, 0.0562, 0.1000, 0.1778, 0.3162, 0.5623, 1.0000])
        self.HtRng = [[55, 1e5 ** 2], [50, 75], [50, 1e5 ** 2], [20, 1e5 ** 2]]
        self.OccRng = [[0, 1], [0, 1], [2], [0, 1, 2]]
        self.SetupLbl = ["Reasonable", "Reasonable_small", "Reasonable_occ=heavy", "All"]
        self.bndRng = [5, 5, 635, 507]
        self.expFilter = 1.25
        self.fixRowIndex = False      # False: the published `id - first kept id` row lookup; True: every kept detection's own row


class KAIST:
    """The annotation file (or a result set attached to one) as flat tables."""

    def __init__(self, annotation_file=None):
        self.dataset = {"images": [], "annotations": [], "categories": []}
        if annotation_file is not None:
            with open(annotation_file) as f:
                ds = json.load(f)
            if not isinstance(ds, dict) or "images" not in ds or "annotations" not in ds:
                raise ValueError(f"{annotation_file}: not a KAIST annotation file (needs 'images' and 'annotations'; the per-frame list "
                                 "format of earlier proben_amd versions is no longer read)")
            self.dataset = ds
        self._index()

    def _index(self):
        self.imgs = {im["id"]: im for im in self.dataset.get("images", [])}
        self.imgToAnns = {}
        for a in self.dataset.get("annotations", []):
            self.imgToAnns.setdefault(a["image_id"], []).append(a)

    def getImgIds(self):
        return list(self.imgs.keys())

    def anns(self, img_id, cat_ids):
        return [a for a in self.imgToAnns.get(img_id, []) if a.get("category_id", 1) in cat_ids]

    @staticmethod
    def txt2json(txt):
        """`<1-based frame index>,x,y,w,h,score` rows -> COCO result dicts (image ids 0-based, category 1)."""
        out = []
        with open(txt) as f:
            for line in f:
                if not line.strip():
                    continue
                v = [float(t) for t in line.split(",")]
                out.append({"image_id": v[0] - 1, "category_id": 1, "bbox": [v[1], v[2], v[3], v[4]], "score": v[5]})
        return out

    def loadRes(self, res_file):
        """A result file (.txt rows or .json list) -> a KAIST object over the same images; ids follow file order, height = bbox h."""
        res = KAIST()
        res.dataset["images"] = list(self.dataset["images"])
        res.dataset["categories"] = copy.deepcopy(self.dataset.get("categories", []))
        if isinstance(res_file, str):
            if res_file.endswith(".json"):
                with open(res_file) as f:
                    anns = json.load(f)
            else:
                anns = self.txt2json(res_file)
        else:
            anns = [dict(a) for a in res_file]
        if not isinstance(anns, list):
            raise ValueError("results must be a list of detections")
        known = set(self.getImgIds())
        stray = sorted({a["image_id"] for a in anns} - known)
        if stray:
            raise ValueError(f"results name image ids that the annotation file does not hold: {stray[:5]}")
        for i, a in enumerate(anns):
            bb = a["bbox"]
            a["area"] = bb[2] * bb[3]
            a["height"] = bb[3]
            a["id"] = i + 1
            a["iscrowd"] = 0
        res.dataset["annotations"] = anns
        res._index()
        return res


def overlaps(dts, gts, gt_ignored):
    """[D, G] overlap of xywh boxes: intersection / union, intersection / detection area against ignored ground truth."""
    dts = np.asarray(dts, dtype=np.float64).reshape(-1, 4)
    gts = np.asarray(gts, dtype=np.float64).reshape(-1, 4)
    ig = np.asarray(gt_ignored, dtype=bool).reshape(-1)
    w = np.minimum(dts[:, None, 0] + dts[:, None, 2], gts[None, :, 0] + gts[None, :, 2]) - np.maximum(dts[:, None, 0], gts[None, :, 0])
    h = np.minimum(dts[:, None, 1] + dts[:, None, 3], gts[None, :, 1] + gts[None, :, 3]) - np.maximum(dts[:, None, 1], gts[None, :, 1])
    hit = (w > 0) & (h > 0)
    inter = np.where(hit, w * h, 0.0)
    darea = (dts[:, 2] * dts[:, 3])[:, None]
    garea = (gts[:, 2] * gts[:, 3])[None, :]
    union = np.where(ig[None, :], darea, darea + garea - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(hit, inter / union, 0.0)


class KAISTPedEval:
    """evaluate(id_setup) -> accumulate() -> summarize(id_setup); `eval['yy'][0]` is the miss-rate curve (one value per counted
    detection), `eval['xx'][0]` the FPPI curve, `eval['TP'][0, :, 0, 0]` the recall at the 9 reference points."""

    def __init__(self, kaistGt=None, kaistDt=None, iouType="bbox", method="unknown"):
        self.cocoGt, self.cocoDt = kaistGt, kaistDt
        self.params = KAISTParams()
        if kaistGt is not None:
            self.params.imgIds = sorted(kaistGt.getImgIds())
        self.method = method
        self.evalImgs, self.eval = [], {}

    # ---- per image ----
    def _gt_table(self, img_id, id_setup):
        p = self.params
        anns = self.cocoGt.anns(img_id, p.catIds)
        box = np.asarray([a["bbox"] for a in anns], dtype=np.float64).reshape(-1, 4)
        ig = np.zeros(len(anns), dtype=bool)
        h0, h1 = p.HtRng[id_setup]
        for i, a in enumerate(anns):
            x, y, w, h = a["bbox"]
            ig[i] = bool(a.get("ignore", 0)) or a["height"] < h0 or a["height"] > h1 or a["occlusion"] not in p.OccRng[id_setup] \
                or x < p.bndRng[0] or y < p.bndRng[1] or x + w > p.bndRng[2] or y + h > p.bndRng[3]
        order = np.argsort(ig, kind="mergesort")       # real boxes first, ignored last (stable)
        return box[order], ig[order]

    def _dt_table(self, img_id, id_setup):
        """Detections of an image in descending score order (stable, at most maxDets): (boxes, scores, ids) of ALL of them and the mask
        of those inside the expanded height range."""
        p = self.params
        anns = self.cocoDt.anns(img_id, p.catIds)
        box = np.asarray([a["bbox"] for a in anns], dtype=np.float64).reshape(-1, 4)
        score = np.asarray([a["score"] for a in anns], dtype=np.float64)
        height = np.asarray([a["height"] for a in anns], dtype=np.float64)
        ids = np.asarray([a["id"] for a in anns], dtype=np.int64)
        order = np.argsort(-score, kind="mergesort")[:p.maxDets[-1]]
        box, score, height, ids = box[order], score[order], height[order], ids[order]
        h0, h1 = p.HtRng[id_setup]
        keep = (height >= h0 / p.expFilter) & (height < h1 * p.expFilter)
        return box, score, ids, keep

    def evaluateImg(self, img_id, id_setup):
        gbox, gig = self._gt_table(img_id, id_setup)
        abox, ascore, aid, keep = self._dt_table(img_id, id_setup)
        dbox, dscore = abox[keep], ascore[keep]
        if len(gbox) == 0 and len(abox) == 0:
            return None
        p = self.params
        if p.fixRowIndex or len(dbox) == 0:
            ov = overlaps(dbox, gbox, gig)
        else:
            # the published lookup: row (id - id of the first kept detection) of the matrix over ALL detections in score order
            rows = aid[keep] - aid[keep][0]
            if rows.max() >= len(abox) or rows.min() < -len(abox):
                raise IndexError(f"image {img_id}: the published row lookup `id - first kept id` = {int(rows.max())} leaves the image's "
                                 f"{len(abox)} detections (the result file does not list the image's rows together); fix_row_index=True uses each detection's own row")
            ov = overlaps(abox, gbox, gig)[rows]
        T, G, D = len(p.iouThrs), len(gbox), len(dbox)
        gtm = np.zeros((T, G), dtype=bool)
        dtm = np.zeros((T, D), dtype=bool)
        dt_ig = np.zeros((T, D), dtype=bool)
        for t, thr in enumerate(p.iouThrs):
            for d in range(D):
                best, bg, real = min(thr, 1 - 1e-10), -1, False
                for g in range(G):
                    if gtm[t, g]:
                        continue                     # a real box already taken (ignored boxes are never marked)
                    if bg >= 0 and gig[g]:
                        break                        # matched already: ignored boxes (sorted last) never replace a match
                    if ov[d, g] < best:
                        continue
                    best, bg, real = ov[d, g], g, not gig[g]
                if bg < 0:
                    continue
                dtm[t, d] = True
                dt_ig[t, d] = gig[bg]
                if real:
                    gtm[t, bg] = True
        return {"image_id": img_id, "dtMatches": dtm, "dtScores": dscore, "dtIgnore": dt_ig, "gtIgnore": gig, "gtMatches": gtm}

    def evaluate(self, id_setup):
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        p.maxDets = sorted(p.maxDets)
        self.evalImgs = [self.evaluateImg(i, id_setup) for i in p.imgIds]

    # ---- over the image set ----
    def accumulate(self, p=None):
        p = p or self.params
        T, R = len(p.iouThrs), len(p.fppiThrs)
        ys = -np.ones((T, R, 1, 1))
        xx, yy = [], []
        n_img = len(p.imgIds)
        E = [e for e in self.evalImgs if e is not None]
        if E:
            scores = np.concatenate([e["dtScores"] for e in E])
            order = np.argsort(-scores, kind="mergesort")
            dtm = np.concatenate([e["dtMatches"] for e in E], axis=1)[:, order]
            dt_ig = np.concatenate([e["dtIgnore"] for e in E], axis=1)[:, order]
            npig = int(sum(np.count_nonzero(~e["gtIgnore"]) for e in E))
            if npig > 0:
                for t in range(T):
                    counted = ~dt_ig[t]
                    tp = np.cumsum(dtm[t][counted]).astype(np.float64)
                    fp = np.cumsum(~dtm[t][counted]).astype(np.float64)
                    fppi = fp / n_img
                    recall = tp / npig
                    xx.append(fppi)
                    yy.append(1 - recall)
                    q = np.zeros(R)
                    if len(tp):
                        inds = np.searchsorted(fppi, p.fppiThrs, side="right") - 1
                        q = recall[inds]            # inds == -1 reads the LAST operating point, as the script's list indexing does
                    ys[t, :, 0, 0] = q
        self.eval = {"params": p, "counts": [T, R, 1, 1], "TP": ys, "xx": xx, "yy": yy}

    def summarize(self, id_setup, res_file=None):
        """Log-average miss rate at IoU 0.5 over the 9 FPPI reference points (-1 when nothing was evaluated)."""
        p = self.params
        if not self.eval:
            raise RuntimeError("run accumulate() first")
        t = int(np.argmin(np.abs(p.iouThrs - 0.5)))
        mrs = 1 - self.eval["TP"][t, :, 0, 0]
        mrs = mrs[mrs < 2]
        if len(mrs) == 0:
            mean_s = -1.0
        else:
            with np.errstate(divide="ignore"):
                mean_s = float(np.exp(np.mean(np.log(mrs))))
        line = " Average Miss Rate  (MR) @ {:<18} [ IoU=0.50 | height={:>6s} | visibility={:>6s} ] = {:0.2f}%".format(
            p.SetupLbl[id_setup], "[{:0.0f}:{:0.0f}]".format(*p.HtRng[id_setup]),
            "[" + "+".join(str(o) for o in p.OccRng[id_setup]) + "]", mean_s * 100)
        if res_file is not None:
            res_file.write(line + "\n")
        return mean_s


DAY_FRAMES = 1455      # test-all-20: image ids 0 .. 1454 are the day sets (set06-08), the rest the night sets (set09-11)


def evaluate(test_annotation_file, user_submission_file, phase_codename="Multispectral", plot=False, fix_row_index=False):
    """The call of demo_LAMR_KAIST.py:145.  Returns {'all' | 'day' | 'night': KAISTPedEval (evaluated, accumulated)} and prints the
    three "Reasonable" miss rates and the recall.  `plot` is accepted and ignored (no matplotlib dependency).  `fix_row_index` (an
    addition, default off): see the module docstring."""
    gt = KAIST(test_annotation_file)
    dt = gt.loadRes(user_submission_file)
    img_ids = sorted(gt.getImgIds())
    method = os.path.basename(user_submission_file).split("_")[0] if isinstance(user_submission_file, str) else "unknown"
    base = KAISTPedEval(gt, dt, "bbox", method)
    base.params.catIds = [1]
    base.params.fixRowIndex = bool(fix_row_index)
    result = {}
    for name, ids in (("all", img_ids), ("day", img_ids[:DAY_FRAMES]), ("night", img_ids[DAY_FRAMES:])):
        ev = copy.copy(base)
        ev.params = copy.deepcopy(base.params)
        ev.params.imgIds = ids
        ev.evaluate(0)
        ev.accumulate()
        result[name] = ev
    mr = {k: v.summarize(0) for k, v in result.items()}
    yy = result["all"].eval["yy"]
    recall_all = float(1 - yy[0][-1]) if yy and len(yy[0]) else float("nan")
    print(f"\n########## Method: {method} ##########\n"
          f"MR_all: {mr['all'] * 100:.2f}\nMR_day: {mr['day'] * 100:.2f}\nMR_night: {mr['night'] * 100:.2f}\n"
          f"recall_all: {recall_all * 100:.2f}\n######################################\n")
    return result
