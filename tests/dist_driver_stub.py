"""Rank entry for tests/test_dist_drivers_cpu.py: run one of the dataset drivers with a STUB predictor (deterministic
detections computed from the pixels, on the host) so the rank plumbing - self-launch, init_distributed over gloo, the
InferenceSampler shard, the row all-gather, rank-0 evaluation / writers - runs without a GPU.  Not product code.

    python tests/dist_driver_stub.py <demo_mAP_FLIR | save_predictions | demo_LAMR_KAIST> <driver args...>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import proben_amd  # noqa: E402,F401
from proben_amd import predictor as P  # noqa: E402
from proben_amd.structures import Boxes, Instances  # noqa: E402


class StubPredictor:
    """Same surface the drivers use (predict_batch, __call__, min_size / max_size); detections are a pure function of the image."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.min_size, self.max_size = 800, 1333
        self.input_format = cfg.INPUT.FORMAT

    def _one(self, img):
        img = np.asarray(img, dtype=np.float64)
        h, w = img.shape[:2]
        rng = np.random.default_rng(int(img[::7, ::5].sum()) % (2 ** 31))
        n = int(rng.integers(0, 9))
        x1, y1 = rng.uniform(0, w - 40, n), rng.uniform(0, h - 40, n)
        b = np.stack([x1, y1, x1 + rng.uniform(8, 39, n), y1 + rng.uniform(8, 39, n)], 1).astype(np.float32).reshape(n, 4)
        logits = rng.normal(0, 2, (n, self.K + 1)).astype(np.float32)
        e = np.exp(logits - logits.max(1, keepdims=True)) if n else logits
        prob = (e / e.sum(1, keepdims=True))[:, :self.K] if n else logits[:, :self.K]
        inst = Instances((h, w))
        inst.pred_boxes = Boxes(torch.from_numpy(b))
        inst.scores = torch.from_numpy(prob.max(1).astype(np.float32)) if n else torch.zeros(0)
        inst.pred_classes = torch.from_numpy(prob.argmax(1).astype(np.int64)) if n else torch.zeros(0, dtype=torch.int64)
        inst.class_logits = torch.from_numpy(logits)
        inst.prob_score = torch.from_numpy(prob.astype(np.float32))
        inst.vars = torch.from_numpy(rng.uniform(0.5, 4, (n, 1)).astype(np.float32))
        return {"instances": inst}

    def predict_batch(self, images):
        return [self._one(im) for im in images]

    def __call__(self, image):
        return self._one(image)


if __name__ == "__main__":
    P.DefaultPredictor = StubPredictor
    proben_amd.DefaultPredictor = StubPredictor
    import importlib
    drv = importlib.import_module("proben_amd.cli." + sys.argv[1])
    drv.main(sys.argv[2:])
