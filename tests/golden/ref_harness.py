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
