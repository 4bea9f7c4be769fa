"""Counterpart of demo/FLIR/demo_mAP_FLIR.py: single-detector mAP through DefaultPredictor ->
build_detection_test_loader -> inference_on_dataset -> FLIREvaluator.

    python -m proben_amd.cli.demo_mAP_FLIR --dataset_path DATA/FLIR/val --fusion_method thermal_only --model_path m.pth
"""
import json
import os
import sys

import torch

from .. import comm, launch
from ..data import DatasetCatalog, build_detection_test_loader, read_image, register_coco_instances, resize_shortest_edge_shape
from ..evaluation import FLIREvaluator, inference_on_dataset
from ..opt import config_parser
from .save_predictions import build_cfg


def main(cmd=None):
    from ..predictor import DefaultPredictor
    argv = list(cmd) if cmd is not None else sys.argv[1:]
    args = config_parser(argv)
    # one process per GPU like the reference's launch(main, num_gpus) (engine/launch.py:24-84); the test loader shards by rank
    # (InferenceSampler) and FLIREvaluator(distributed=True) gathers every rank's rows to rank 0
    launch.maybe_self_launch(args.world_size, argv, module="proben_amd.cli.demo_mAP_FLIR", device=args.device)
    rank, world, dev = launch.init_distributed(args.device, expect_world=args.world_size)
    val_json = os.path.join(args.dataset_path, "FLIR_thermal_RGBT_pairs_val.json")
    register_coco_instances(args.dataset_name, {}, val_json, os.path.join(args.dataset_path, "thermal_8_bit"))
    dicts = DatasetCatalog.get(args.dataset_name)
    cfg = build_cfg(args)
    cfg.DATASETS.TEST = (args.dataset_name,)
    cfg.MODEL.DEVICE = dev.type      # "cuda" = this rank's current device (LOCAL_RANK); "cpu" raises: no CPU detector in the product
    predictor = DefaultPredictor(cfg)

    def mapper(d):
        # the reference registers <val>/thermal_8_bit/ as image root (demo_mAP_FLIR.py:27-31) while the pairs json of
        # demo_FLIR_save_predictions.py:95 names files "thermal_8_bit/<stem>.jpeg": accept both spellings
        path = d["file_name"]
        if not os.path.exists(path):
            path = os.path.join(args.dataset_path, "thermal_8_bit", os.path.basename(path))
        img = read_image(path, cfg.INPUT.FORMAT)
        return {"image_np": img, "height": d["height"], "width": d["width"], "image_id": d["image_id"], "file_name": d["file_name"]}

    def model(inputs):
        return predictor.predict_batch([x["image_np"] for x in inputs])
    ev = FLIREvaluator(args.dataset_name, cfg, world > 1 or comm.is_distributed(), output_dir=args.outfolder if comm.is_main_process() else None)
    res = inference_on_dataset(model, build_detection_test_loader(dicts, mapper), ev)
    if comm.is_main_process():
        print(json.dumps(res, indent=1))
        if args.outfolder:     # the AP table as a file: what a multi-rank run hands back to its caller (rank 0 only)
            os.makedirs(args.outfolder, exist_ok=True)
            with open(os.path.join(args.outfolder, "FLIR_mAP_results.json"), "w") as f:
                json.dump({"world_size": world, "results": res}, f)
    if comm.is_distributed():
        launch.shutdown()
    return res


if __name__ == "__main__":
    main()
