"""Counterpart of demo/FLIR/demo_probEn.py:300-342: read val_<method>_predictions.json files, ProbEn-fuse
them on the GPU, evaluate with FLIREvaluator.

    python -m proben_amd.cli.demo_probEn --dataset_path DATA/FLIR/val --prediction_path out/ \
        --score_fusion probEn --box_fusion v-avg [--detectors thermal_only,early_fusion,middle_fusion]
"""
import json
import os
import sys

from .. import comm, get_cfg, launch
from ..data import DatasetCatalog, register_coco_instances
from ..evaluation import FLIREvaluator
from ..late_fusion import apply_late_fusion_and_evaluate, read_j1, shard_j1
from ..opt import config_parser


def main(cmd=None):
    argv = list(cmd) if cmd is not None else sys.argv[1:]
    args = config_parser(argv)
    # --world-size N: the image list is cut into N contiguous blocks, every rank fuses its block on its own GPU and the
    # evaluator gathers the fused rows to rank 0 (one tensor all-gather over RCCL)
    launch.maybe_self_launch(args.world_size, argv, module="proben_amd.cli.demo_probEn", device=args.device)
    rank, world, dev = launch.init_distributed(args.device, expect_world=args.world_size)
    names = [n for n in args.detectors.split(",") if n]
    assert 2 <= len(names) <= 3, "--detectors takes 2 or 3 names"
    files = [os.path.join(args.prediction_path, f"val_{n}_predictions.json") for n in names]
    if comm.is_main_process():
        for i, f in enumerate(files):
            print(f"detection file {i + 1}:", f)
        os.makedirs(args.outfolder, exist_ok=True)
    val_json = os.path.join(args.dataset_path, "FLIR_thermal_RGBT_pairs_val.json")
    register_coco_instances(args.dataset_name, {}, val_json, os.path.join(args.dataset_path, "thermal_8_bit"))
    DatasetCatalog.get(args.dataset_name)
    cfg = get_cfg()
    cfg.OUTPUT_DIR = args.outfolder
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.5
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 3
    cfg.DATASETS.TEST = (args.dataset_name,)
    dets = [read_j1(f) for f in files]
    # the files are sharded with ONE index range: they must list the same images in the same order (the reference pairs by det_2's
    # position too, demo_probEn.py:205-233, and silently mis-pairs otherwise)
    for f, d in zip(files[:-1], dets[:-1]):
        if d["image_id"] != dets[-1]["image_id"]:
            raise ValueError(f"{f} and {files[-1]} do not list the same images in the same order ({len(d['image_id'])} vs "
                             f"{len(dets[-1]['image_id'])} entries): late fusion pairs detections by position")
    mine = comm.shard_range(len(dets[-1]["image"]))
    dets = [shard_j1(d, mine) for d in dets]
    with open(val_json) as f:
        hw = {im["id"]: (im["height"], im["width"]) for im in json.load(f)["images"]}
    main_rank = comm.is_main_process()
    ev = FLIREvaluator(args.dataset_name, cfg, world > 1 or comm.is_distributed(), output_dir=args.outfolder if main_rank else None, save_eval=main_rank,
                       out_eval_path=os.path.join(args.outfolder, "FLIR_probEn_eval.json"))
    res = apply_late_fusion_and_evaluate(cfg, ev, dets[0], dets[1], [args.score_fusion, args.box_fusion],
                                         det_3=dets[2] if len(dets) > 2 else "", image_hw=hw, device=str(dev))
    if main_rank:
        print(json.dumps(res, indent=1))
    if comm.is_distributed():
        launch.shutdown()
    return res


if __name__ == "__main__":
    main()
