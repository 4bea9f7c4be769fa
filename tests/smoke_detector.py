"""Run by __graft_entry__.smoke(): one small R50-FPN forward on cuda:0 checked against the oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_detector_gpu import check_pair, run_pair  # noqa: E402

want, inter, det, _ = run_pair(50, hw=(320, 416), n_images=2)
check_pair(want, inter, det)
print("smoke ok: R50-FPN forward, detections per image:", det["counts"].tolist())
