"""The ProbEn CLI flag set (detectron2/utils/opt.py:3-19; the reference uses configargparse, plain argparse
here) plus the MI355X additions --device / --batch / --world-size."""
import argparse


def config_parser(cmd=None):
    p = argparse.ArgumentParser()
    p.add_argument("--outfolder", type=str, default="out", help="name of output folder")
    p.add_argument("--dataset_name", type=str, default="FLIR", help="name of dataset")
    p.add_argument("--dataset_path", type=str, default=None, help="path to dataset")
    p.add_argument("--prediction_path", type=str, default=None, help="path to model predictions")
    p.add_argument("--fusion_method", type=str, default="middle_fusion",
                   choices=["rgb_only", "thermal_only", "early_fusion", "middle_fusion"], help="Which fusion method to use?")
    p.add_argument("--model_path", type=str, default=None, help="path to trained model")
    p.add_argument("--score_fusion", type=str, default="probEn", choices=["avg", "max", "probEn"], help="Which fusion method to use?")
    p.add_argument("--box_fusion", type=str, default="v-avg", choices=["avg", "s-avg", "v-avg", "argmax"], help="Which fusion method to use?")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--world-size", type=int, default=1,
                   help="ranks (one per GPU) the dataset is sharded over; > 1 re-launches the driver under torch.distributed.run (launch.py)")
    p.add_argument("--detectors", type=str, default="thermal_only,early_fusion,middle_fusion",
                   help="comma separated prediction files to fuse, in order (val_<name>_predictions.json)")
    return p.parse_args(cmd) if cmd is not None else p.parse_args()
