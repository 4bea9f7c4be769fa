"""MI355X-native RGB+thermal detection and ProbEn fusion (inference hot path).

Host-side mirror of the reference's Python surface for this path
(`DefaultPredictor`, `Instances`, `Boxes`, `fusion`, `batched_nms`, `ROIAlign`,
`FLIREvaluator`, the `demo_probEn.py` flags) on top of the C-ABI library
`libproben_hip.so` (include/proben_hip.h).  There is NO CPU fallback in this
package: every compute entry point raises if the HIP library or a GPU is absent.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
