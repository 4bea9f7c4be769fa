"""Build-container-only helper: import pieces of the upstream reference
(/root/reference) so golden vectors can be GENERATED from the reference's own
code.  Never imported by tests, smoke() or bench.py - the reference does not
exist on the GPU box; only the .npz/.json fixtures written by the gen_*.py
scripts travel.

The reference needs third-party packages that are absent here (torchvision,
cv2, fvcore, yacs, configargparse, pycocotools, termcolor).  For the ProbEn
script only EMPTY placeholder modules are registered (its arithmetic uses
NumPy + torch.Tensor only), so the generated vectors are the reference's own
arithmetic.  For the detector pieces see gen_detector.py.
"""
import importlib.util
import sys
import types

REF = "/root/reference"


_PLACED = []


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    _PLACED.append(name)
    return m


def load_reference_proben():
    """Load demo/FLIR/demo_probEn.py by path with empty placeholder imports."""
    import torch  # noqa: F401  (the real one, before placeholders go in)
    try:
        tv = _placeholder("torchvision")
        tv.ops = _placeholder("torchvision.ops", boxes=None, nms=None)
        _placeholder("torchvision.ops.boxes")
        _placeholder("cv2")
        d2 = _placeholder("detectron2")
        for sub, attrs in [
            ("config", {"get_cfg": None}),
            ("data", {"DatasetCatalog": None, "MetadataCatalog": None}),
            ("data.datasets", {"register_coco_instances": None}),
            ("structures", {"Instances": None, "Boxes": None}),
            ("evaluation", {"FLIREvaluator": None}),
            ("layers", {}),
            ("layers.nms", {"batched_nms": None}),
            ("utils", {}),
            ("utils.opt", {"config_parser": None}),
        ]:
            _placeholder("detectron2." + sub, **attrs)
        spec = importlib.util.spec_from_file_location(
            "_reference_demo_probEn", REF + "/demo/FLIR/demo_probEn.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    finally:
        while _PLACED:
            sys.modules.pop(_PLACED.pop(), None)


# ----------------------------------------------------------------------------------------------
# Detector pieces.  The reference's detectron2 fork needs fvcore / yacs / torchvision /
# pycocotools / cv2 / termcolor and a compiled detectron2._C, none of which exist here.  The
# stand-ins below provide ONLY plumbing (config node, registry, weight-init, file helpers);
# two of them carry arithmetic and are therefore flagged where fixtures depend on them:
#   * torchvision.ops.nms / batched_nms  -> oracle.nms (the public torchvision algorithm;
#     "parity unpinned" at the NMS boundary, see oracle/nms.py);
#   * detectron2._C.roi_align_forward    -> the reference's OWN ROIAlign_cpu.cpp, compiled in
#     /tmp from a copy with its two `.type()` tokens changed to `.scalar_type()` (the unmodified
#     file does not compile against torch 2.10).  Build-container scratch only.
# ----------------------------------------------------------------------------------------------
_D2_READY = False


class _Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


def _make_cfgnode():
    import yaml

    class CfgNode(dict):
        def __init__(self, init=None, *a, **k):
            super().__init__()
            for kk, v in (init or {}).items():
                self[kk] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            import copy
            return copy.deepcopy(self)

        def freeze(self):
            pass

        def defrost(self):
            pass

        def is_frozen(self):
            return False

        @staticmethod
        def _conv(v):
            if isinstance(v, str) and v.startswith("(") and v.endswith(")"):
                import ast
                try:
                    return ast.literal_eval(v)
                except Exception:
                    return v
            return v

        @classmethod
        def load_yaml_with_base(cls, filename, allow_unsafe=False):
            import os
            with open(filename) as f:
                cfg = yaml.safe_load(f)

            def merge(a, b):
                for k, v in a.items():
                    if isinstance(v, dict) and isinstance(b.get(k), dict):
                        merge(v, b[k])
                    else:
                        b[k] = v
            if "_BASE_" in cfg:
                base = cfg.pop("_BASE_")
                if not os.path.isabs(base):
                    base = os.path.join(os.path.dirname(filename), base)
                bcfg = cls.load_yaml_with_base(base)
                merge(cfg, bcfg)
                return bcfg
            return cfg

        def merge_from_other_cfg(self, other):
            for k, v in other.items():
                if isinstance(v, dict) and isinstance(self.get(k), dict):
                    self[k].merge_from_other_cfg(v)
                else:
                    self[k] = CfgNode(v) if isinstance(v, dict) else self._conv(v)

        def merge_from_list(self, lst):
            for k, v in zip(lst[0::2], lst[1::2]):
                node = self
                parts = k.split(".")
                for p in parts[:-1]:
                    node = node[p]
                node[parts[-1]] = v
    return CfgNode


def install_detectron2_standins():
    """Make `import detectron2.modeling` work from /root/reference (build container only)."""
    global _D2_READY
    if _D2_READY:
        return
    import os
    import torch
    import PIL.Image
    if not hasattr(PIL.Image, "LINEAR"):
        PIL.Image.LINEAR = PIL.Image.BILINEAR
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import nms as onms

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    # cv2 / pycocotools / termcolor / tabulate-free
    cv2 = mod("cv2", __version__="0.0", INTER_LINEAR=1, INTER_CUBIC=2)
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *_: None)
    cv2.setNumThreads = lambda *_: None
    mod("pycocotools")
    mod("pycocotools.mask")
    mod("termcolor", colored=lambda s, *a, **k: s)
    # torchvision
    def tv_nms(boxes, scores, thr):
        keep = onms.nms_f32(boxes.detach().numpy(), scores.detach().numpy(), thr)
        return torch.from_numpy(keep)

    def tv_batched_nms(boxes, scores, idxs, thr):
        keep = onms.batched_nms_f32(boxes.detach().numpy(), scores.detach().numpy(), idxs.numpy(), thr,
                                    device_type=os.environ.get("PE_NMS_DEVICE_SEMANTICS", "cuda"))
        return torch.from_numpy(keep)
    tv = mod("torchvision", __version__="0.13.0")
    tvops = mod("torchvision.ops", nms=tv_nms, RoIPool=_Dummy)
    tvboxes = mod("torchvision.ops.boxes", batched_nms=tv_batched_nms, nms=tv_nms)
    tvops.boxes = tvboxes
    tv.ops = tvops
    # fvcore
    CfgNode = _make_cfgnode()
    mod("fvcore")
    mod("fvcore.common")
    mod("fvcore.common.config", CfgNode=CfgNode)
    mod("fvcore.common.registry", Registry=_Registry)

    class PathManager:
        isfile = staticmethod(os.path.isfile)
        open = staticmethod(open)
        exists = staticmethod(os.path.exists)
        mkdirs = staticmethod(lambda p: os.makedirs(p, exist_ok=True))
        get_local_path = staticmethod(lambda p: p)
        register_handler = staticmethod(lambda *a, **k: None)
    mod("fvcore.common.file_io", PathManager=PathManager, PathHandler=_Dummy, HTTPURLHandler=_Dummy,
        file_lock=None)
    mod("fvcore.common.history_buffer", HistoryBuffer=_Dummy)
    mod("fvcore.common.timer", Timer=_Dummy)
    mod("fvcore.common.checkpoint", Checkpointer=_Dummy, PeriodicCheckpointer=_Dummy,
        _IncompatibleKeys=_Dummy, _strip_prefix_if_present=None, get_missing_parameters_message=None,
        get_unexpected_parameters_message=None)
    mod("fvcore.common.benchmark", benchmark=None)

    def smooth_l1_loss(*a, **k):
        raise NotImplementedError
    wi = mod("fvcore.nn.weight_init",
             c2_msra_fill=lambda m: (torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu"),
                                     m.bias is not None and torch.nn.init.constant_(m.bias, 0)),
             c2_xavier_fill=lambda m: (torch.nn.init.kaiming_uniform_(m.weight, a=1),
                                       m.bias is not None and torch.nn.init.constant_(m.bias, 0)))
    mod("fvcore.nn", smooth_l1_loss=smooth_l1_loss, sigmoid_focal_loss_jit=None, weight_init=wi,
        giou_loss=None, sigmoid_focal_loss=None)
    mod("fvcore.nn.precise_bn", get_bn_modules=None, update_bn_stats=None)

    class Transform:
        @classmethod
        def register_type(cls, *a, **k):
            pass

        def _set_attributes(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)

    class TransformList(list):
        pass
    tr = mod("fvcore.transforms.transform", Transform=Transform, TransformList=TransformList,
             NoOpTransform=type("NoOpTransform", (Transform,), {}), HFlipTransform=type("HFlipTransform", (Transform,), {}),
             VFlipTransform=type("VFlipTransform", (Transform,), {}), BlendTransform=type("BlendTransform", (Transform,), {}),
             CropTransform=type("CropTransform", (Transform,), {}), GridSampleTransform=type("GridSampleTransform", (Transform,), {}),
             ScaleTransform=type("ScaleTransform", (Transform,), {}))
    tr.__all__ = [k for k in tr.__dict__ if k.endswith("Transform") or k == "TransformList"]
    mod("fvcore.transforms", transform=tr)
    # detectron2._C: the reference's own ROIAlign CPU arithmetic, compiled in /tmp
    mod("detectron2_C_placeholder")
    sys.path.insert(0, REF)
    import detectron2  # noqa: F401  (the reference package itself)
    sys.modules["detectron2._C"] = _load_reference_roialign()
    detectron2._C = sys.modules["detectron2._C"]
    _D2_READY = True


def _load_reference_roialign():
    """Compile the reference's ROIAlign_cpu.cpp (two-token patched scratch copy) in /tmp."""
    import os
    import re
    from torch.utils.cpp_extension import load
    src_dir = REF + "/detectron2/layers/csrc/ROIAlign"
    work = "/tmp/pe_ref_roialign"
    os.makedirs(work, exist_ok=True)
    cpp = open(src_dir + "/ROIAlign_cpu.cpp").read()
    cpp = cpp.replace("input.type()", "input.scalar_type()").replace("grad.type()", "grad.scalar_type()")
    open(work + "/ROIAlign_cpu.cpp", "w").write(cpp)
    hdr = open(src_dir + "/ROIAlign.h").read()
    hdr = hdr.replace(".type().is_cuda()", ".is_cuda()")
    open(work + "/ROIAlign.h", "w").write(hdr)
    open(work + "/bind.cpp", "w").write(
        '#include <torch/extension.h>\n#include "ROIAlign.h"\n'
        'PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {\n'
        '  m.def("roi_align_forward", &detectron2::ROIAlign_forward);\n'
        '  m.def("roi_align_backward", &detectron2::ROIAlign_backward);\n}\n')
    return load(name="pe_ref_roialign", sources=[work + "/ROIAlign_cpu.cpp", work + "/bind.cpp"],
                build_directory=work, verbose=False)
