"""Lab only: bench.py on a variant build of the library (scripts/lab/build_variant.py), for same-box pipeline A/Bs.
    python scripts/lab/bench_with_lib.py exp_<tag> --steps 100 --warmup 5 --no-cpu-baseline --no-roofline --no-micro"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import proben_amd  # noqa: E402,F401
from proben_amd import _lib  # noqa: E402

name = sys.argv[1]
if name != "product":
    _lib.LIB_PATH = _lib.LIB_PATH.replace(".so", "_%s.so" % name)
    assert os.path.exists(_lib.LIB_PATH), _lib.LIB_PATH
if os.environ.get("PE_RING_WGS"):      # lab: workgroups of the persistent 1x1 kernel (csrc/test_hooks.h)
    _lib.test_hooks().pe_test_set_ring_wgs(int(os.environ["PE_RING_WGS"]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
