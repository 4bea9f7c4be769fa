"""Counterpart of demo/FLIR/demo_probEn.py:300-342: read val_<method>_predictions.json files, ProbEn-fuse
them on the GPU, evaluate with FLIREvaluator.

    python -m proben_amd.cli.demo_probEn --dataset_path DATA/FLIR/val --prediction_path out/ \
        --score_fusion probEn --box_fusion v-avg [--detectors thermal_only,early_fusion,middle_fusion]
"""
import json
import os

from .. import get_cfg
from ..data import DatasetCatalog, register_coco_instances
from ..evaluation import FLIREvaluator
from ..late_fusion import apply_late_fusion_and_evaluate, read_j1
from ..opt import config_parser


def main(cmd=None):
    args = config_parser(cmd)
    names = [n for n in args.detectors.split(",") if n]
    assert 2 <= len(names) <= 3, "--detectors takes 2 or 3 names"
    files = [os.path.join(args.prediction_path, f"val_{n}_predictions.json") for n in names]
    for i, f in enumerate(files):
        print(f"detection file {i + 1}:", f)
    val_json = os.path.join(args.dataset_path, "FLIR_thermal_RGBT_pairs_val.json")
    os.makedirs(args.outfolder, exist_ok=True)
    register_coco_instances(args.dataset_name, {}, val_json, os.path.join(args.dataset_path, "thermal_8_bit"))
    DatasetCatalog.get(args.dataset_name)
    cfg = get_cfg()
    cfg.OUTPUT_DIR = args.outfolder
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.5
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 3
    cfg.DATASETS.TEST = (args.dataset_name,)
    dets = [read_j1(f) for f in files]
    with open(val_json) as f:
        hw = {im["id"]: (im["height"], im["width"]) for im in json.load(f)["images"]}
    ev = FLIREvaluator(args.dataset_name, cfg, False, output_dir=args.outfolder, save_eval=True,
                       out_eval_path=os.path.join(args.outfolder, "FLIR_probEn_eval.json"))
    res = apply_late_fusion_and_evaluate(cfg, ev, dets[0], dets[1], [args.score_fusion, args.box_fusion],
                                         det_3=dets[2] if len(dets) > 2 else "", image_hw=hw, device=args.device)
    print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    main()
