"""MI355X-native RGB+thermal detection and ProbEn fusion (inference hot path).

Host-side mirror of the reference's Python surface for this path
(`DefaultPredictor`, `Instances`, `Boxes`, `fusion`, `batched_nms`, `ROIAlign`,
`FLIREvaluator`, the `demo_probEn.py` flags) on top of the C-ABI library
`libproben_hip.so` (include/proben_hip.h).  There is NO CPU fallback in this
package: every compute entry point raises if the HIP library or a GPU is absent.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
from .config import CfgNode, get_cfg  # noqa: F401
from .structures import Boxes, ImageList, Instances  # noqa: F401


def __getattr__(name):  # heavier modules load lazily
    if name == "DefaultPredictor":
        from .predictor import DefaultPredictor
        return DefaultPredictor
    if name in ("FLIREvaluator", "inference_on_dataset"):
        from . import evaluation
        return getattr(evaluation, name)
    if name == "GeneralizedRCNN":
        from .rcnn import GeneralizedRCNN
        return GeneralizedRCNN
    if name in ("build_model", "Registry", "META_ARCH_REGISTRY", "Box2BoxTransform", "DefaultAnchorGenerator", "ShapeSpec",
                "DetectionCheckpointer", "ResizeShortestEdge"):
        from . import modeling
        return getattr(modeling, name)
    raise AttributeError(name)
