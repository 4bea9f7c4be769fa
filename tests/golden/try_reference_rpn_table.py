"""Build-container-only ATTEMPT to pin the torchvision-NMS stand-in (oracle/nms.py) through the one
reference-held vector that passes through NMS: the post-NMS proposal table of /root/reference/tests/test_rpn.py:47-65
(torch.manual_seed(121), default C4 RPN, features torch.rand(2, 1024, 1, 2)).

    python tests/golden/try_reference_rpn_table.py

Outcome in this container (torch 2.10.0, recorded in DESIGN.md section 4): the reference's own modules run (eval mode: the
proposals are computed under no_grad from the same predict_proposals / find_top_rpn_proposals calls as in training mode), the
STRUCTURE of the table reproduces - 2 proposals survive NMS 0.7 for image 0 and 5 for image 1, the first box of image 0 is
the full clipped image - but the VALUES do not (top logit 0.1220 here vs 0.12254 in the table): they are functions of the
RPN head's `normal_(std=0.01)` weights and of `torch.rand` features, i.e. of the CPU generator stream behind seed 121, and the
torch 1.4-era stream that produced the table is not what torch 2.10 draws (the table was never regenerated upstream).  No
NMS rule can be tuned to close a difference that is already present in the pre-NMS logits, so the stand-in stays
"parity unpinned"; nothing is written under tests/golden/ by this script."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PE_NMS_DEVICE_SEMANTICS"] = "cpu"
import ref_harness as H  # noqa: E402

H.install_detectron2_standins()
import torch  # noqa: E402
from detectron2.config import get_cfg  # noqa: E402
from detectron2.modeling.backbone import build_backbone  # noqa: E402
from detectron2.modeling.proposal_generator.build import build_proposal_generator  # noqa: E402
from detectron2.structures import ImageList  # noqa: E402

EXPECTED_BOXES = [[[0, 0, 10, 10], [7.3365392685, 0, 10, 10]],
                  [[0, 0, 30, 20], [0, 0, 16.7862777710, 13.1362524033], [0, 0, 30, 13.3173446655], [0, 0, 10.8602609634, 20],
                   [7.7165775299, 0, 27.3875980377, 20]]]
EXPECTED_LOGITS = [[0.1225359365, -0.0133192837], [0.1415634006, 0.0989848152, 0.0565387346, -0.0072308783, -0.0428492837]]

torch.manual_seed(121)
cfg = get_cfg()
cfg.MODEL.PROPOSAL_GENERATOR.NAME = "RPN"
cfg.MODEL.ANCHOR_GENERATOR.NAME = "DefaultAnchorGenerator"
cfg.MODEL.RPN.BBOX_REG_WEIGHTS = (1, 1, 1, 1)
backbone = build_backbone(cfg)
pg = build_proposal_generator(cfg, backbone.output_shape())
images = ImageList(torch.rand(2, 20, 30), [(10, 10), (20, 30)])
features = {"res4": torch.rand(2, 1024, 1, 2)}
pg.eval()
with torch.no_grad():
    proposals, _ = pg(images, features, None)
ok = True
for p, eb, el in zip(proposals, EXPECTED_BOXES, EXPECTED_LOGITS):
    same_n = len(p) == len(eb)
    same_v = same_n and torch.allclose(p.proposal_boxes.tensor, torch.tensor(eb, dtype=torch.float32)) and \
        torch.allclose(p.objectness_logits, torch.tensor(el))
    print("survivors", len(p), "expected", len(eb), "| values reproduce:", bool(same_v))
    print("  logits here    ", [round(float(x), 7) for x in p.objectness_logits])
    print("  logits expected", el)
    ok = ok and same_v
print("reference table reproduced:", ok)
