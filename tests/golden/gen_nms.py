"""Pin greedy NMS with code the REFERENCE holds - build container only, nothing of the reference travels.

torchvision (whose `nms` / `batched_nms` the reference calls, layers/nms.py:16-37) is absent here, but the reference tree holds two
independent statements of horizontal greedy NMS that its own test suite equates with torchvision's:
  (i)  tests/test_nms_rotated.py:11-33  `reference_horizontal_nms` - Python: sort by score descending, keep, drop `iou > thr`
       (asserted equal to the rotated kernel at 0 degrees, :89-101);
  (ii) layers/csrc/nms_rotated/nms_rotated_cpu.cpp:7-60 + box_iou_rotated/box_iou_rotated_utils.h:315-340 - C++ (polygon-clipping
       IoU, suppresses at `iou >= thr`), asserted equal to torchvision's batched_nms at 0 degrees for IoU 0.2 / 0.5 / 0.8 (:45-66).
Both are EXECUTED here on this repo's own fixtures ((i): the method's source is read from the reference file at generation time and
exec'ed with a 5-line box_iou standing in for torchvision.ops.box_iou; (ii): compiled in /tmp from a scratch copy whose `.type()`
tokens are changed to `.scalar_type()` / `.is_cuda()`, as for ROIAlign) and only inputs + keep lists are written to
tests/golden/nms_reference.npz.  tests/test_oracle_nms.py and the -m gpu tests then require oracle/nms.py and the HIP kernel to
reproduce them.

    python tests/golden/gen_nms.py"""
import ast
import os
import sys
import textwrap

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_horizontal_nms():
    src = open(REF + "/tests/test_nms_rotated.py").read()
    tree = ast.parse(src)
    fn = next(n for c in tree.body if isinstance(c, ast.ClassDef) for n in c.body
              if isinstance(n, ast.FunctionDef) and n.name == "reference_horizontal_nms")
    code = textwrap.dedent("\n".join(src.splitlines()[fn.lineno - 1:fn.end_lineno]))

    class ops:                       # torchvision.ops.box_iou's published definition (boxes.py: box_area / box_iou)
        @staticmethod
        def box_iou(a, b):
            area_a, area_b = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]), (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
            wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
            inter = wh[:, :, 0] * wh[:, :, 1]
            return inter / (area_a[:, None] + area_b - inter)
    ns = {"torch": torch, "ops": ops}
    exec(code, ns)
    return lambda boxes, scores, thr: ns["reference_horizontal_nms"](None, boxes, scores, thr)


def load_reference_nms_rotated():
    from torch.utils.cpp_extension import load
    src = REF + "/detectron2/layers/csrc"
    work = "/tmp/pe_ref_nms_rotated"
    os.makedirs(work + "/nms_rotated", exist_ok=True)
    os.makedirs(work + "/box_iou_rotated", exist_ok=True)
    cpp = open(src + "/nms_rotated/nms_rotated_cpu.cpp").read()
    cpp = cpp.replace("dets.type().is_cuda()", "dets.is_cuda()").replace("scores.type().is_cuda()", "scores.is_cuda()")
    cpp = cpp.replace("dets.type() == scores.type()", "dets.scalar_type() == scores.scalar_type()").replace("dets.type()", "dets.scalar_type()")
    open(work + "/nms_rotated/nms_rotated_cpu.cpp", "w").write(cpp)
    open(work + "/nms_rotated/nms_rotated.h", "w").write(open(src + "/nms_rotated/nms_rotated.h").read())
    open(work + "/box_iou_rotated/box_iou_rotated_utils.h", "w").write(open(src + "/box_iou_rotated/box_iou_rotated_utils.h").read())
    open(work + "/nms_rotated/bind.cpp", "w").write(
        '#include <torch/extension.h>\n#include "nms_rotated.h"\n'
        'PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("nms_rotated", &detectron2::nms_rotated_cpu); }\n')
    return load(name="pe_ref_nms_rotated", sources=[work + "/nms_rotated/nms_rotated_cpu.cpp", work + "/nms_rotated/bind.cpp"],
                build_directory=work + "/nms_rotated", verbose=False).nms_rotated


def unique_scores(rng, n):
    s = rng.permutation(n).astype(np.float32) / np.float32(n)          # all distinct in float32: no tie-order question
    return (0.05 + 0.9 * s).astype(np.float32)


def cases():
    rng = np.random.default_rng(20260927)
    out = {}
    # (a) RPN-like: 4624 proposals of one image on an 800 x 1000 canvas, clustered around 60 objects (dense overlaps)
    ctr = rng.uniform([50, 50], [950, 750], (60, 2))
    which = rng.integers(0, 60, 4624)
    c = ctr[which] + rng.normal(0, 18, (4624, 2))
    wh = np.exp(rng.normal(np.log(90), 0.5, (4624, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    b[:, 0::2] = b[:, 0::2].clip(0, 1000)
    b[:, 1::2] = b[:, 1::2].clip(0, 800)
    out["rpn"] = (b.astype(np.float32), unique_scores(rng, 4624))
    # (b) detection-like: 1500 boxes around 25 objects with small jitter (IoU mass near the thresholds)
    ctr = rng.uniform([80, 80], [560, 430], (25, 2))
    which = rng.integers(0, 25, 1500)
    c = ctr[which] + rng.normal(0, 6, (1500, 2))
    wh = np.exp(rng.normal(np.log(70), 0.25, (1500, 2)))
    out["det"] = (np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32), unique_scores(rng, 1500))
    # (c) the reference test's own recipe (tests/test_nms_rotated.py:35-43), N = 2000
    g = torch.Generator().manual_seed(7)
    bb = torch.rand(2000, 4, generator=g) * 100
    bb[:, 2:] += bb[:, :2]
    out["uniform"] = (bb.numpy().astype(np.float32), unique_scores(rng, 2000))
    return out


def main():
    horiz = load_reference_horizontal_nms()
    rotated = load_reference_nms_rotated()
    save = {}
    for name, (b, s) in cases().items():
        tb, ts = torch.from_numpy(b), torch.from_numpy(s)
        rb = torch.zeros(len(b), 5)
        rb[:, 0], rb[:, 1] = (tb[:, 0] + tb[:, 2]) / 2.0, (tb[:, 1] + tb[:, 3]) / 2.0      # the conversion of test_nms_rotated.py:51-55
        rb[:, 2], rb[:, 3] = tb[:, 2] - tb[:, 0], tb[:, 3] - tb[:, 1]
        save[name + "_boxes"], save[name + "_scores"] = b, s
        for thr in (0.5, 0.7):
            k1 = horiz(tb, ts, thr).numpy().astype(np.int64)
            k2 = rotated(rb, ts, float(thr)).numpy().astype(np.int64)
            save[f"{name}_keep_python_{thr}"] = k1
            save[f"{name}_keep_rotated_{thr}"] = k2
            print(f"{name} thr {thr}: {len(b)} boxes -> python reference keeps {len(k1)}, rotated C++ keeps {len(k2)}, identical: {np.array_equal(k1, k2)}")
    np.savez_compressed(os.path.join(HERE, "nms_reference.npz"), **save)


if __name__ == "__main__":
    sys.exit(main())
