"""Generate tests/golden/j1_case.json: detections in, the prediction file the REFERENCE's writer emits out
(demo/FLIR/demo_FLIR_save_predictions.py:133-176) - build container only.

`save_predictions` builds a predictor and reads a dataset, so it cannot be called here; like gen_kaist.py this generator reads the writer
statements - the per-image block from `predictions = predict['instances'].to('cpu')` to `var_dict.append(out_vars)` and the tail from
`out_dicts['image'] = image_dict` to the `json.dump` - from /root/reference AT GENERATION TIME and executes them unchanged on stub
predictor outputs built with the reference's own `Instances` / `Boxes`.  Nothing of the reference's text is stored: the fixture holds the
input detections (float32 bit patterns) and the emitted file text."""
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

SRC = "/root/reference/demo/FLIR/demo_FLIR_save_predictions.py"
lines = open(SRC).read().split("\n")
b0 = next(i for i, l in enumerate(lines) if l.strip().startswith("predictions = predict['instances'].to('cpu')"))
b1 = next(i for i in range(b0, len(lines)) if lines[i].strip() == "var_dict.append(out_vars)")
t0 = next(i for i in range(b1, len(lines)) if lines[i].strip().startswith("out_dicts['image'] = image_dict"))
t1 = next(i for i in range(t0, len(lines)) if "json.dump(out_dicts, outfile, indent=2)" in lines[i])
body = compile(textwrap.dedent("\n".join(lines[b0:b1 + 1])), SRC + f":{b0 + 1}-{b1 + 1}", "exec")
tail_src = textwrap.dedent("\n".join(lines[t0:t1 + 1]))

rng = np.random.default_rng(11)
K = 3
frames = []
for n in (4, 0, 1, 6):
    x1 = rng.uniform(0, 500, n).astype(np.float32)
    y1 = rng.uniform(0, 400, n).astype(np.float32)
    b = np.stack([x1, y1, x1 + rng.uniform(5, 130, n).astype(np.float32), y1 + rng.uniform(5, 110, n).astype(np.float32)], 1).astype(np.float32).reshape(n, 4)
    logits = rng.normal(0, 2, (n, K + 1)).astype(np.float32)
    probs = torch.softmax(torch.from_numpy(logits), 1)[:, :K].numpy()
    frames.append({"boxes": b, "scores": probs.max(1), "classes": probs.argmax(1).astype(np.int64), "logits": logits, "probs": probs,
                   "vars": rng.uniform(0.5, 3.0, (n, 1)).astype(np.float32)})
# awkward values: integers, many digits; classes above 2 (dropped by the writer: the COCO model's other categories)
frames[0]["boxes"][0] = np.array([10.0, 20.0, 30.0, 60.0], np.float32)
frames[0]["scores"][0] = np.float32(1.0)
frames[0]["classes"][1] = 7
frames[3]["classes"][2] = 3
frames[3]["classes"][5] = 79
frames[3]["boxes"][1] = np.array([0.1, 0.2, 0.30000001, 511.99997], np.float32)
files_names = ["FLIR_08865.jpg", "FLIR_09001.jpg", "FLIR_08999.jpg", "FLIR_10000.jpg"]       # os.listdir(RGB) order: whatever the directory gives
name_to_id = {"FLIR_00001": 0, "FLIR_00002": 17, "FLIR_00003": 2, "FLIR_00004": 1365}
stems = list(name_to_id)

g = {"image_dict": [], "boxes_dict": [], "scores_dict": [], "classes_dict": [], "class_logits_dict": [], "prob_dict": [], "img_id_dict": [],
     "var_dict": [], "out_dicts": {}, "files_names": files_names, "name_to_id_dict": name_to_id, "json": json}
for i, fr in enumerate(frames):
    inst = Instances((512, 640))
    inst.pred_boxes = Boxes(torch.from_numpy(fr["boxes"].copy()))
    inst.scores = torch.from_numpy(fr["scores"].copy())
    inst.pred_classes = torch.from_numpy(fr["classes"].copy())
    inst.class_logits = torch.from_numpy(fr["logits"].copy())
    inst.prob_score = torch.from_numpy(fr["probs"].copy())
    inst.vars = torch.from_numpy(fr["vars"].copy())
    g.update({"predict": {"instances": inst}, "i": i, "file_name": stems[i]})
    exec(body, g)
buf = io.StringIO()
# the tail opens the output file itself: `with open(out_pred_file, 'w') as outfile:` - give it an in-memory one
g["out_pred_file"] = "unused"
g["open"] = lambda *a, **k: type("F", (), {"__enter__": lambda s: buf, "__exit__": lambda s, *e: False})()
exec(compile(tail_src, SRC + f":{t0 + 1}-{t1 + 1}", "exec"), g)

u32 = lambda a: np.asarray(a, np.float32).reshape(-1).view(np.uint32).tolist()  # noqa: E731
out = {"source": "demo/FLIR/demo_FLIR_save_predictions.py:%d-%d and :%d-%d executed on stub outputs (numpy %s, torch %s)" % (b0 + 1, b1 + 1, t0 + 1, t1 + 1, np.__version__, torch.__version__),
       "K": K, "files_names": files_names, "image_ids": [name_to_id[s] for s in stems],
       "frames": [{"n": len(fr["scores"]), "boxes_u32": u32(fr["boxes"]), "scores_u32": u32(fr["scores"]), "classes": fr["classes"].tolist(),
                   "logits_u32": u32(fr["logits"]), "probs_u32": u32(fr["probs"]), "vars_u32": u32(fr["vars"])} for fr in frames],
       "text": buf.getvalue()}
json.dump(out, open(os.path.join(HERE, "j1_case.json"), "w"), indent=1)
print(out["text"][:600])
print(len(out["text"]), "bytes of prediction file;", out["source"])
