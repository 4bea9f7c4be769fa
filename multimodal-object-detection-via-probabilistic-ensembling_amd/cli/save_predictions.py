"""Counterpart of demo/FLIR/demo_FLIR_save_predictions.py: run ONE detector over the FLIR val set (batched
on the GPU, images sharded over ranks) and write val_<method>_predictions.json in the reference's schema.

    python -m proben_amd.cli.save_predictions --dataset_path DATA/FLIR/val --fusion_method thermal_only \
        --model_path model.pth --prediction_path out/ [--batch 16]
Input building per method follows :98-121 (early: B,G,R,T(ch 0); middle: B,G,R,T,T,T; RGB resized to the
thermal size with OpenCV's 2x2-tap bilinear rule, data.cv2_linear_resize_u8)."""
import json
import os

import numpy as np

import sys

from .. import comm, get_cfg, launch
from ..data import cv2_linear_resize_u8, read_image
from ..late_fusion import predictions_to_j1, write_j1
from ..opt import config_parser

THERMAL_MEAN = 135.438
COCO_ZOO_WEIGHTS = "trained_models/Detectron2_pretrained/model_final_f6e8b1.pkl"   # demo_FLIR_save_predictions.py:60


def build_cfg(args, config_dir=None):
    cfg = get_cfg()
    cfg.MODEL.RESNETS.DEPTH = 101
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.5
    cfg.MODEL.ROI_BOX_HEAD.OUTPUT_LOGITS = True
    cfg.MODEL.ROI_HEADS.ENABLE_GAUSSIANNLLOSS = True
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 3
    cfg.MODEL.WEIGHTS = args.model_path or "synthetic://1"
    m = args.fusion_method
    if m == "rgb_only":
        # demo_FLIR_save_predictions.py:59-61: the RGB detector is the COCO model-zoo R101-FPN (80 classes, zoo pickle), whatever
        # --model_path says; its detections of classes > 2 are dropped when the predictions are written (:149, predictions_to_j1).
        # The zoo model has no var_pred layer (the reference leaves it at its random initialisation): variance 1 here.
        cfg.MODEL.ROI_HEADS.NUM_CLASSES = 80
        if os.path.exists(COCO_ZOO_WEIGHTS):
            cfg.MODEL.WEIGHTS = COCO_ZOO_WEIGHTS
    elif m == "early_fusion":
        cfg.INPUT.FORMAT, cfg.INPUT.NUM_IN_CHANNELS = "BGRT", 4
        cfg.MODEL.PIXEL_MEAN = [103.53, 116.28, 123.675, THERMAL_MEAN]
        cfg.MODEL.PIXEL_STD = [1.0, 1.0, 1.0, 1.0]
    elif m == "middle_fusion":
        cfg.INPUT.FORMAT, cfg.INPUT.NUM_IN_CHANNELS = "BGRTTT", 6
        cfg.MODEL.PIXEL_MEAN = [103.53, 116.28, 123.675, THERMAL_MEAN, THERMAL_MEAN, THERMAL_MEAN]
        cfg.MODEL.PIXEL_STD = [1.0] * 6
    return cfg


def resize_bilinear(img, hw):
    """RGB frame -> thermal frame size with OpenCV's 8-bit INTER_LINEAR rule (what `cv2.resize(rgb, size, cv2.INTER_CUBIC)`
    at demo_FLIR_save_predictions.py:109 really computes: the flag sits in the `dst` slot)."""
    return cv2_linear_resize_u8(img, hw[0], hw[1])


def load_input(method, rgb_file, thermal_file):
    if method == "rgb_only":
        return read_image(rgb_file, "BGR")
    t = read_image(thermal_file, "BGR")
    if method == "thermal_only":
        return t
    rgb = resize_bilinear(read_image(rgb_file, "BGR"), t.shape[:2])
    if method == "early_fusion":
        return np.concatenate([rgb, t[:, :, :1]], axis=2).astype(np.float64)
    return np.concatenate([rgb, t], axis=2).astype(np.float64)


def main(cmd=None):
    from ..predictor import DefaultPredictor
    argv = list(cmd) if cmd is not None else sys.argv[1:]
    args = config_parser(argv)
    launch.maybe_self_launch(args.world_size, argv, module="proben_amd.cli.save_predictions", device=args.device)
    rank, world, dev = launch.init_distributed(args.device, expect_world=args.world_size)
    with open(os.path.join(args.dataset_path, "FLIR_thermal_RGBT_pairs_val.json")) as f:
        data = json.load(f)
    cfg = build_cfg(args)
    cfg.MODEL.DEVICE = dev.type      # "cuda" = this rank's current device (LOCAL_RANK); "cpu" raises: no CPU detector in the product
    predictor = DefaultPredictor(cfg)
    images = data["images"]
    idx = list(comm.shard_range(len(images)))     # InferenceSampler: this rank's contiguous block
    names, ids, insts = [], [], []
    for b0 in range(0, len(idx), args.batch):
        chunk = [images[i] for i in idx[b0:b0 + args.batch]]
        batch = []
        for im in chunk:
            stem = os.path.splitext(os.path.basename(im["file_name"]))[0]
            batch.append(load_input(args.fusion_method, os.path.join(args.dataset_path, "RGB", stem + ".jpg"),
                                    os.path.join(args.dataset_path, "thermal_8_bit", stem + ".jpeg")))
            names.append(stem + ".jpeg")
            ids.append(im["id"])
        insts += [o["instances"] for o in predictor.predict_batch(batch)]
    pred = predictions_to_j1(names, ids, insts)
    out = os.path.join(args.prediction_path or args.outfolder, "val_" + args.fusion_method + "_predictions.json")
    if comm.is_main_process():
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    gathered = comm.gather(pred, dst=0)           # ragged per-image lists (logits, probs, vars): pickled, over the gloo side group
    if comm.is_main_process():
        merged = {k: sum((g[k] for g in gathered), []) for k in pred}      # rank order == dataset order
        write_j1(out, merged)
        print("out file:", out)
    if comm.is_distributed():
        launch.shutdown()
    return out


if __name__ == "__main__":
    main()
