"""`DefaultPredictor` - the reference's single-image predictor (detectron2/engine/defaults.py:133-198):
build the model from cfg, load `cfg.MODEL.WEIGHTS`, take one HWC image in cfg.INPUT.FORMAT order
(BGR / RGB / BGRT / BGRTTT), resize the shortest edge to MIN_SIZE_TEST capped at MAX_SIZE_TEST and return
`{"instances": Instances}` at the original resolution.  `predict_batch` is the batched extension the
MI355X path is built around (configs 2-5)."""
import numpy as np
import torch

from . import _lib
from .data import MetadataCatalog, resize_shortest_edge_shape
from .rcnn import DetectorConfig, GeneralizedRCNN
from .weights import load_state_dict_file


def detector_config_from_cfg(cfg):
    fmt = cfg.INPUT.FORMAT
    assert fmt in ["RGB", "BGR", "BGRT", "BGRTTT"], fmt
    sizes = tuple(s[0] if isinstance(s, (list, tuple)) else s for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES)
    mn = cfg.INPUT.MIN_SIZE_TEST
    return DetectorConfig(
        num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES, input_format=fmt, pixel_mean=tuple(cfg.MODEL.PIXEL_MEAN),
        pixel_std=tuple(cfg.MODEL.PIXEL_STD), anchor_sizes=sizes, aspect_ratios=tuple(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0]),
        pre_nms_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, post_nms_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TEST,
        rpn_nms_thresh=cfg.MODEL.RPN.NMS_THRESH, score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
        nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, detections_per_image=cfg.TEST.DETECTIONS_PER_IMAGE,
        min_size_test=mn[0] if isinstance(mn, (list, tuple)) else mn, max_size_test=cfg.INPUT.MAX_SIZE_TEST,
        output_logits=bool(cfg.MODEL.ROI_BOX_HEAD.OUTPUT_LOGITS), enable_gaussian_nll=bool(cfg.MODEL.ROI_HEADS.ENABLE_GAUSSIANNLLOSS),
        fix_vars=bool(cfg.get("PROBEN", {}).get("FIX_VARS", False)))


def load_weights(cfg):
    """cfg.MODEL.WEIGHTS: a `.pth` state dict (reference format) or `synthetic://<seed>` (seeded random init
    of the configured architecture - there are no checkpoints offline)."""
    w = cfg.MODEL.WEIGHTS
    if isinstance(w, dict):
        return w
    if isinstance(w, str) and w.startswith("synthetic://"):
        from .synthetic import synthetic_state_dict
        nin = {"BGR": 3, "RGB": 3, "BGRT": 4, "BGRTTT": 6}[cfg.INPUT.FORMAT]
        return synthetic_state_dict(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.ROI_HEADS.NUM_CLASSES, nin, seed=int(w.split("//")[1] or 1))
    if not w:
        raise ValueError("cfg.MODEL.WEIGHTS is empty: give a .pth state dict or 'synthetic://<seed>'")
    return load_state_dict_file(w)


class DefaultPredictor:
    def __init__(self, cfg):
        self.cfg = cfg.clone()
        if str(self.cfg.MODEL.DEVICE) != "cuda":
            raise _lib.HipLibraryError(
                f"MODEL.DEVICE={self.cfg.MODEL.DEVICE}: proben_amd ships the MI355X path only (no CPU fallback); "
                "the CPU restatement of the reference lives in oracle/ as test infrastructure.")
        from .modeling import META_ARCH_REGISTRY, ResizeShortestEdge
        META_ARCH_REGISTRY.get(self.cfg.MODEL.META_ARCHITECTURE)      # unknown architectures fail like build_model(cfg)
        # build_model(cfg) + DetectionCheckpointer(model).load(cfg.MODEL.WEIGHTS) of defaults.py:161-168 in one step (the
        # weights are packed for the kernels exactly once)
        self.model = GeneralizedRCNN(detector_config_from_cfg(self.cfg), load_weights(self.cfg))
        self.model.eval()
        self.metadata = MetadataCatalog.get(self.cfg.DATASETS.TEST[0]) if len(self.cfg.DATASETS.TEST) else None
        mn = self.cfg.INPUT.MIN_SIZE_TEST
        mn = mn[0] if isinstance(mn, (list, tuple)) else mn
        self.transform_gen = ResizeShortestEdge([mn, mn], self.cfg.INPUT.MAX_SIZE_TEST)      # defaults.py:170-172
        self.input_format = self.cfg.INPUT.FORMAT
        assert self.input_format in ["RGB", "BGR", "BGRT", "BGRTTT"], self.input_format
        self.min_size, self.max_size = self.model.cfg.min_size_test, self.model.cfg.max_size_test

    def _to_device(self, img):
        img = np.asarray(img)
        if self.input_format == "RGB":
            img = img[:, :, ::-1]  # the caller hands BGR; the model was trained on RGB (defaults.py:188-190)
        # 3-channel images go through Pillow as uint8 in the reference (transform.py:92-95: img.astype(np.uint8));
        # the 4 / 6-channel fusion inputs are floating point there (cv2.resize branch, transform.py:82-91)
        img = img.astype(np.uint8, copy=False) if img.shape[2] == 3 else img.astype(np.float32, copy=False)
        return torch.from_numpy(np.ascontiguousarray(img)).to(self.model.device)

    def predict_batch(self, images):
        """images: list of HWC arrays of one common size -> list[{"instances": Instances}]."""
        h, w = images[0].shape[:2]
        for im in images:
            assert im.shape[:2] == (h, w), "predict_batch expects equally sized frames"
        det = self.model.forward_batch(torch.stack([self._to_device(im) for im in images]), out_sizes=[(h, w)] * len(images),
                                       resize_to=resize_shortest_edge_shape(h, w, self.min_size, self.max_size))
        return self.model.to_instances(det)

    def __call__(self, original_image):
        with torch.no_grad():
            return self.predict_batch([original_image])[0]
