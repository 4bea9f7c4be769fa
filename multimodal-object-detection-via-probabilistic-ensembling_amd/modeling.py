"""The plugin / operator boundary of the reference's modeling layer, for the inference path:

  Registry, META_ARCH_REGISTRY / BACKBONE_REGISTRY / PROPOSAL_GENERATOR_REGISTRY / RPN_HEAD_REGISTRY /
  ANCHOR_GENERATOR_REGISTRY / ROI_HEADS_REGISTRY / ROI_BOX_HEAD_REGISTRY      (utils/registry.py -> fvcore Registry;
                                                                               meta_arch/build.py:4-19, backbone/build.py:7-33,
                                                                               proposal_generator/build.py, rpn.py:16-31,
                                                                               anchor_generator.py:12-19, roi_heads.py:23-38,
                                                                               box_head.py:11-101)
  build_model(cfg)                      meta_arch/build.py:12-19 - selected by cfg.MODEL.META_ARCHITECTURE, returned on
                                        cfg.MODEL.DEVICE, NO weights loaded (random init, like the reference)
  DetectionCheckpointer(model).load(p)  checkpoint/detection_checkpoint.py - `.pth` state dicts (the C2 / pkl zoo formats
                                        are out of scope: SURVEY 2)
  Box2BoxTransform(weights).apply_deltas / get_deltas      modeling/box_regression.py:16-110
  DefaultAnchorGenerator(cfg, input_shape)                 modeling/anchor_generator.py:59-199
  ShapeSpec                                                layers/shape_spec.py
  ResizeShortestEdge(short, max).get_transform(img).apply_image(img)   data/transforms/transform_gen.py:157-213 +
                                                                      transform.py:55-98 (what DefaultPredictor.transform_gen is)

The detector itself is ONE fused implementation (rcnn.GeneralizedRCNN: backbone, RPN and ROI heads are streams of HIP
launches, not swappable nn.Modules), so the component registries hold the reference's NAME strings as markers:
`build_model` looks every configured name up - an unknown name fails with the registry's KeyError exactly like the
reference - and then builds the fused detector.  No CPU fallback: MODEL.DEVICE must be cuda."""
import math
from collections import namedtuple

import numpy as np
import torch

from . import _lib

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Registry:
    """name -> object mapping with the decorator / call registration forms of fvcore.common.registry.Registry."""

    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj):
        assert name not in self._obj_map, "An object named '{}' was already registered in '{}' registry!".format(name, self._name)
        self._obj_map[name] = obj

    def register(self, obj=None):
        if obj is None:                      # used as a decorator: @REGISTRY.register()
            def deco(func_or_class):
                self._do_register(func_or_class.__name__, func_or_class)
                return func_or_class
            return deco
        self._do_register(obj.__name__, obj)  # used as a function call

    def get(self, name):
        ret = self._obj_map.get(name)
        if ret is None:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name))
        return ret

    def __contains__(self, name):
        return name in self._obj_map


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, *, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)


def _fused_component(name, doc):
    """Marker for a component that lives inside the fused detector: looking it up works, calling it explains."""
    def build(*args, **kwargs):
        raise NotImplementedError(f"{name} is part of the fused MI355X detector (proben_amd.rcnn.GeneralizedRCNN); "
                                  "it cannot be instantiated on its own - use build_model(cfg)")
    build.__name__, build.__doc__ = name, doc
    return build


BACKBONE_REGISTRY.register(_fused_component("build_resnet_fpn_backbone", "ResNet-50/101 + FPN (backbone/fpn.py:183-204)"))
PROPOSAL_GENERATOR_REGISTRY.register(_fused_component("RPN", "proposal_generator/rpn.py:88-187"))
RPN_HEAD_REGISTRY.register(_fused_component("StandardRPNHead", "proposal_generator/rpn.py:34-85"))
ROI_HEADS_REGISTRY.register(_fused_component("StandardROIHeads", "roi_heads/roi_heads.py:481-631"))
ROI_BOX_HEAD_REGISTRY.register(_fused_component("FastRCNNConvFCHead", "roi_heads/box_head.py:19-96"))


@META_ARCH_REGISTRY.register()
def GeneralizedRCNN(cfg):
    """cfg -> the fused detector with seeded random-init weights of the configured architecture (the reference's
    build_model does not load weights either; DetectionCheckpointer / load_state_dict does)."""
    from .predictor import detector_config_from_cfg
    from .rcnn import GeneralizedRCNN as Fused
    from .synthetic import synthetic_state_dict
    for reg, name in ((BACKBONE_REGISTRY, cfg.MODEL.BACKBONE.NAME), (PROPOSAL_GENERATOR_REGISTRY, cfg.MODEL.PROPOSAL_GENERATOR.NAME),
                      (RPN_HEAD_REGISTRY, cfg.MODEL.RPN.HEAD_NAME), (ANCHOR_GENERATOR_REGISTRY, cfg.MODEL.ANCHOR_GENERATOR.NAME),
                      (ROI_HEADS_REGISTRY, cfg.MODEL.ROI_HEADS.NAME), (ROI_BOX_HEAD_REGISTRY, cfg.MODEL.ROI_BOX_HEAD.NAME)):
        reg.get(name)
    if str(cfg.MODEL.DEVICE) != "cuda":
        raise _lib.HipLibraryError(f"MODEL.DEVICE={cfg.MODEL.DEVICE}: proben_amd ships the MI355X path only (no CPU fallback); "
                                   "the CPU restatement of the reference lives in oracle/ as test infrastructure.")
    nin = {"BGR": 3, "RGB": 3, "BGRT": 4, "BGRTTT": 6}[cfg.INPUT.FORMAT]
    sd = synthetic_state_dict(cfg.MODEL.RESNETS.DEPTH, cfg.MODEL.ROI_HEADS.NUM_CLASSES, nin, seed=0)
    return Fused(detector_config_from_cfg(cfg), sd)


def build_model(cfg):
    """Build the whole model architecture defined by cfg.MODEL.META_ARCHITECTURE (meta_arch/build.py:12-19)."""
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)


class DetectionCheckpointer:
    """`.pth` state-dict loading into a built model (the slice of checkpoint/detection_checkpoint.py the demos use)."""

    def __init__(self, model, save_dir=""):
        self.model = model

    def load(self, path):
        if not path:
            return {}
        from .weights import load_state_dict_file
        if isinstance(path, str) and path.startswith("synthetic://"):
            from .synthetic import synthetic_state_dict
            cfg = self.model.cfg
            sd = synthetic_state_dict(self.model.depth, cfg.num_classes, cfg.in_channels, seed=int(path.split("//")[1] or 1))
        else:
            sd = path if isinstance(path, dict) else load_state_dict_file(path)
        self.model.load_state_dict(sd)
        return {"model": sd}


# ------------------------------------------------------------------------------------------------------------------
class Box2BoxTransform:
    """(dx, dy, dw, dh) box parameterisation of R-CNN: same constructor and methods as the reference's class."""

    def __init__(self, weights, scale_clamp=_DEFAULT_SCALE_CLAMP):
        self.weights = tuple(float(w) for w in weights)
        self.scale_clamp = scale_clamp

    def apply_deltas(self, deltas, boxes):
        """deltas [N, 4k] (k class-specific transforms per box), boxes [N, 4] -> [N, 4k]; HIP kernel behind the C-ABI."""
        import ctypes
        _lib.require_cuda(deltas, boxes)
        assert deltas.dim() == 2 and deltas.shape[1] % 4 == 0 and boxes.shape == (deltas.shape[0], 4)
        d = deltas.detach().float().contiguous()
        b = boxes.detach().to(d.dtype).contiguous()
        out = torch.empty_like(d)
        w = (ctypes.c_float * 4)(*self.weights)
        st = _lib.lib().pe_box2box_apply_deltas(_lib.ptr(d), _lib.ptr(b), d.shape[0], d.shape[1] // 4, w, float(self.scale_clamp),
                                                _lib.ptr(out), _lib.stream())
        _lib.check(st, "pe_box2box_apply_deltas")
        return out.to(deltas.dtype)

    def get_deltas(self, src_boxes, target_boxes):
        """The inverse (training-side; plain tensor math on whatever device the boxes live on)."""
        assert isinstance(src_boxes, torch.Tensor), type(src_boxes)
        assert isinstance(target_boxes, torch.Tensor), type(target_boxes)
        swh = src_boxes[:, 2:] - src_boxes[:, :2]
        twh = target_boxes[:, 2:] - target_boxes[:, :2]
        sc = src_boxes[:, :2] + 0.5 * swh
        tc = target_boxes[:, :2] + 0.5 * twh
        w = src_boxes.new_tensor(self.weights)
        assert (swh[:, 0] > 0).all().item(), "Input boxes to Box2BoxTransform are not valid!"
        return torch.cat([w[:2] * (tc - sc) / swh, w[2:] * torch.log(twh / swh)], dim=1)


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator:
    """Anchors of every feature level: cell anchors (size x aspect ratio, centred on 0) tiled over the grid with the
    level's stride, order (y, x, anchor).  `forward(features)` returns the reference's list (images) of lists (levels)
    of Boxes.  (Inside the fused detector no anchor tensor exists - rpn.hip evaluates the same closed form per
    surviving candidate.)"""

    def __init__(self, cfg, input_shape):
        from .rcnn import cell_anchor_table
        self.strides = [x.stride for x in input_shape]
        self.offset = float(cfg.MODEL.ANCHOR_GENERATOR.OFFSET)
        assert 0.0 <= self.offset < 1.0, self.offset
        self.num_features = len(self.strides)
        sizes = [list(s) if isinstance(s, (list, tuple)) else [s] for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES]
        ratios = [list(r) for r in cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS]
        if len(sizes) == 1:
            sizes = sizes * self.num_features
        if len(ratios) == 1:
            ratios = ratios * self.num_features
        assert self.num_features == len(sizes) and self.num_features == len(ratios)
        self._cell_host = [np.asarray(cell_anchor_table(s, r), dtype=np.float64).reshape(-1, 4) for s, r in zip(sizes, ratios)]
        self._cell_dev = {}

    @property
    def box_dim(self):
        return 4

    @property
    def num_cell_anchors(self):
        return [len(c) for c in self._cell_host]

    @property
    def cell_anchors(self):
        return [torch.tensor(c, dtype=torch.float32) for c in self._cell_host]

    def generate_cell_anchors(self, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
        from .rcnn import cell_anchor_table
        return torch.tensor(np.asarray(cell_anchor_table(sizes, aspect_ratios)).reshape(-1, 4))

    def grid_anchors(self, grid_sizes, device="cuda"):
        dev = torch.device(device)
        out = []
        for i, (size, stride) in enumerate(zip(grid_sizes, self.strides)):
            key = (i, str(dev))
            if key not in self._cell_dev:
                self._cell_dev[key] = torch.tensor(self._cell_host[i], dtype=torch.float32, device=dev)
            cell = self._cell_dev[key]
            _lib.require_cuda(cell)
            h, w = int(size[0]), int(size[1])
            a = torch.empty((h * w * cell.shape[0], 4), dtype=torch.float32, device=dev)
            st = _lib.lib().pe_grid_anchors(_lib.ptr(cell), cell.shape[0], h, w, int(stride), self.offset, _lib.ptr(a), _lib.stream())
            _lib.check(st, "pe_grid_anchors")
            out.append(a)
        return out

    def forward(self, features):
        from .structures import Boxes
        num_images = len(features[0])
        per_level = self.grid_anchors([f.shape[-2:] for f in features], features[0].device)
        return [[Boxes(a.clone()) for a in per_level] for _ in range(num_images)]

    __call__ = forward


# ------------------------------------------------------------------------------------------------------------------
class ResizeTransform:
    """apply_image of transform.py:55-98: 3-channel images through Pillow BILINEAR on uint8, anything else through
    OpenCV's INTER_LINEAR rule on float (restated: data.cv2_linear_resize_f)."""

    def __init__(self, h, w, new_h, new_w):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w

    def apply_image(self, img):
        assert img.shape[:2] == (self.h, self.w)
        if (self.h, self.w) == (self.new_h, self.new_w):
            return img
        if img.shape[2] == 3:
            from PIL import Image
            return np.asarray(Image.fromarray(img.astype(np.uint8)).resize((self.new_w, self.new_h), Image.BILINEAR))
        from .data import cv2_linear_resize_f
        return cv2_linear_resize_f(img, self.new_h, self.new_w)

    def apply_coords(self, coords):
        coords = np.asarray(coords, dtype=np.float64).copy()
        coords[:, 0] *= self.new_w * 1.0 / self.w
        coords[:, 1] *= self.new_h * 1.0 / self.h
        return coords


class ResizeShortestEdge:
    """DefaultPredictor.transform_gen: `get_transform(img)` -> ResizeTransform with the size rule of
    transform_gen.py:192-213 (sample_style "choice" over the given short edge lengths; inference passes [s, s])."""

    def __init__(self, short_edge_length, max_size=1333, sample_style="choice"):
        self.short_edge_length = (short_edge_length, short_edge_length) if isinstance(short_edge_length, int) else tuple(short_edge_length)
        self.max_size = max_size
        self.sample_style = sample_style

    def get_transform(self, img):
        from .data import resize_shortest_edge_shape
        h, w = img.shape[:2]
        size = int(np.random.choice(self.short_edge_length)) if len(set(self.short_edge_length)) > 1 else self.short_edge_length[0]
        nh, nw = resize_shortest_edge_shape(h, w, size, self.max_size)
        return ResizeTransform(h, w, nh, nw)
