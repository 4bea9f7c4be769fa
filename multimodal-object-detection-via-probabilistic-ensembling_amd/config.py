"""A small attribute-dict config holding the cfg KEYS the reference's demos set on the inference path
(detectron2/config/defaults.py).  YAML files of the reference's `configs/` (with `_BASE_`
inheritance, config/config.py:24-65) merge into it unchanged; unknown keys are allowed, like the
reference's scripts that add new ones on the fly."""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    @staticmethod
    def _coerce(v):
        if isinstance(v, str) and v.startswith("(") and v.endswith(")"):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @classmethod
    def load_yaml_with_base(cls, filename):
        with open(filename) as f:
            cfg = yaml.safe_load(f) or {}
        if "_BASE_" in cfg:
            base = cfg.pop("_BASE_")
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            merged = cls.load_yaml_with_base(base)
            _deep_merge(cfg, merged)
            return merged
        return cfg

    def merge_from_file(self, filename):
        assert os.path.isfile(filename), f"Config file '{filename}' does not exist!"
        self.merge_from_other_cfg(self.load_yaml_with_base(filename))

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else self._coerce(v)

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = self._coerce(v)


def _deep_merge(src, dst):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_merge(v, dst[k])
        else:
            dst[k] = v


_DEFAULTS = {
    "VERSION": 2,
    "OUTPUT_DIR": "./output",
    "MODEL": {
        "META_ARCHITECTURE": "GeneralizedRCNN", "DEVICE": "cuda", "WEIGHTS": "", "MASK_ON": False, "KEYPOINT_ON": False,
        "PIXEL_MEAN": [103.530, 116.280, 123.675], "PIXEL_STD": [1.0, 1.0, 1.0], "BLUR_RGB": False,
        "BACKBONE": {"NAME": "build_resnet_fpn_backbone", "FREEZE_AT": 2},
        "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res2", "res3", "res4", "res5"], "NORM": "FrozenBN", "STRIDE_IN_1X1": True,
                    "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64},
        "FPN": {"IN_FEATURES": ["res2", "res3", "res4", "res5"], "OUT_CHANNELS": 256, "NORM": "", "FUSE_TYPE": "sum"},
        "ANCHOR_GENERATOR": {"NAME": "DefaultAnchorGenerator", "SIZES": [[32], [64], [128], [256], [512]],
                             "ASPECT_RATIOS": [[0.5, 1.0, 2.0]], "OFFSET": 0.0},
        "PROPOSAL_GENERATOR": {"NAME": "RPN", "MIN_SIZE": 0},
        "RPN": {"HEAD_NAME": "StandardRPNHead", "IN_FEATURES": ["p2", "p3", "p4", "p5", "p6"], "PRE_NMS_TOPK_TEST": 1000,
                "POST_NMS_TOPK_TEST": 1000, "NMS_THRESH": 0.7, "BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0)},
        "ROI_HEADS": {"NAME": "StandardROIHeads", "NUM_CLASSES": 80, "IN_FEATURES": ["p2", "p3", "p4", "p5"],
                      "SCORE_THRESH_TEST": 0.05, "NMS_THRESH_TEST": 0.5, "ENABLE_GAUSSIANNLLOSS": False},
        "ROI_BOX_HEAD": {"NAME": "FastRCNNConvFCHead", "NUM_FC": 2, "FC_DIM": 1024, "NUM_CONV": 0, "POOLER_RESOLUTION": 7,
                         "POOLER_SAMPLING_RATIO": 0, "POOLER_TYPE": "ROIAlignV2", "BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0),
                         "CLS_AGNOSTIC_BBOX_REG": False, "OUTPUT_LOGITS": False},
    },
    "INPUT": {"MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "FORMAT": "BGR", "NUM_IN_CHANNELS": 3},
    "DATASETS": {"TRAIN": (), "TEST": ()},
    "TEST": {"DETECTIONS_PER_IMAGE": 100, "KEYPOINT_OKS_SIGMAS": []},
    "PROBEN": {"FIX_VARS": False},
}


def get_cfg():
    """Defaults of the keys on the inference path (drop-in for detectron2.config.get_cfg)."""
    return CfgNode(copy.deepcopy(_DEFAULTS))
