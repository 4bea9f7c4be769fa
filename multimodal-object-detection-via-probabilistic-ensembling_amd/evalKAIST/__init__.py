"""`evalKAIST.evaluation_script.evaluate` - the KAIST miss-rate evaluator the reference imports (demo/KAIST/demo_LAMR_KAIST.py:85)."""
from .evaluation_script import evaluate  # noqa: F401
