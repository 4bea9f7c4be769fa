"""Generate tests/golden/kaist_rows.json: detections in, the text rows the REFERENCE's KAIST writer emits out
(demo/KAIST/demo_LAMR_KAIST.py:127-143) - build container only.

The reference file is a top-level script (it builds a predictor and opens dataset files on import), so it cannot be
imported; this generator reads the writer statements - the body of its per-frame loop from `variance = ...` to
`f.write('\\n')` - from /root/reference AT GENERATION TIME and executes them unchanged on stub predictor outputs built
with the reference's own `Instances` / `Boxes`.  Nothing of the reference's text is stored: the fixture holds the
input detections (float32 bit patterns as hex) and the emitted text."""
import io
import json
import os
import sys
import textwrap

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.install_detectron2_standins()
from detectron2.structures import Boxes, Instances  # noqa: E402  (the reference's containers)

SRC = "/root/reference/demo/KAIST/demo_LAMR_KAIST.py"
lines = open(SRC).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.strip().startswith("variance = outputs['instances']"))
end = next(i for i in range(start, len(lines)) if lines[i].strip() == "f.write('\\n')")
body = textwrap.dedent("\n".join(lines[start:end + 1]))
writer = compile(body, SRC + f":{start + 1}-{end + 1}", "exec")

rng = np.random.default_rng(5)
frames = []
for n in (3, 0, 1, 4):
    x1 = rng.uniform(0, 500, n).astype(np.float32)
    y1 = rng.uniform(0, 400, n).astype(np.float32)
    b = np.stack([x1, y1, x1 + rng.uniform(5, 130, n).astype(np.float32), y1 + rng.uniform(5, 110, n).astype(np.float32)], 1).astype(np.float32).reshape(n, 4)
    frames.append({"boxes": b, "scores": np.sort(rng.uniform(0.5, 1.0, n).astype(np.float32))[::-1].copy(),
                   "vars": rng.uniform(0.5, 3.0, (n, 1)).astype(np.float32)})
# a few awkward values: integers, many digits, tiny width
frames[0]["boxes"][0] = np.array([10.0, 20.0, 30.0, 60.0], np.float32)
frames[0]["scores"][0] = np.float32(1.0)
frames[3]["boxes"][1] = np.array([0.1, 0.2, 0.30000001, 511.99997], np.float32)

f = io.StringIO()
var_dict = {}
for i, fr in enumerate(frames):
    inst = Instances((512, 640))
    inst.pred_boxes = Boxes(torch.from_numpy(fr["boxes"].copy()))
    inst.scores = torch.from_numpy(fr["scores"].copy())
    inst.pred_classes = torch.zeros(len(fr["scores"]), dtype=torch.int64)
    inst.vars = torch.from_numpy(fr["vars"].copy())
    exec(writer, {"outputs": {"instances": inst}, "i": i, "f": f, "var_dict": var_dict, "np": np})

hexf = lambda a: np.asarray(a, np.float32).reshape(-1).view(np.uint32).tolist()  # noqa: E731
out = {"source": "demo/KAIST/demo_LAMR_KAIST.py:%d-%d executed on stub outputs (numpy %s, torch %s)" % (start + 1, end + 1, np.__version__, torch.__version__),
       "frames": [{"n": len(fr["scores"]), "boxes_u32": hexf(fr["boxes"]), "scores_u32": hexf(fr["scores"]), "vars_u32": hexf(fr["vars"])} for fr in frames],
       "text": f.getvalue(), "var_keys": sorted(int(k) for k in var_dict)}
json.dump(out, open(os.path.join(HERE, "kaist_rows.json"), "w"), indent=1)
print(out["text"])
print(out["var_keys"], out["source"])
