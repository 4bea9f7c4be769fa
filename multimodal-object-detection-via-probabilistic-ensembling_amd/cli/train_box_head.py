"""Fine-tune the box head (incl. the Gaussian-NLL variance head) of a Faster R-CNN on frozen features, one process per GPU - the
MI355X-native slice of demo/FLIR/demo_train_FLIR.py (DefaultTrainer + DistributedDataParallel + SGD).  No FLIR annotations exist
offline, so the default data are synthetic labelled frames (`synthetic.labelled_frames`: rectangles whose texture encodes their class);
a real run passes its own (frames, boxes, classes) iterator to `finetune.BoxHeadFineTuner.step`.

    python -m proben_amd.cli.train_box_head --steps 200 --images-per-step 4 --world-size 2 --out model.pth [--weights model.pth] [--resume model.pth]

`--out` writes what the reference's DetectionCheckpointer writes (engine/defaults.py:264-275): {"model": the WHOLE detector's state dict
under the reference's keys - the frozen layers as loaded, the trained head un-permuted and split back into cls_score / bbox_pred /
var_pred -, "optimizer": master weights' momentum + step count + sampler state, "iteration"}; the file loads through `--weights` /
`--model_path` of every driver (`weights.load_state_dict_file` reads {"model": ...}) and `--resume` continues the run from it.
"""
import argparse
import json
import sys
import time

import torch

from .. import comm, launch
from ..finetune import BoxHeadFineTuner, warmup_lr


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--images-per-step", type=int, default=4, help="per rank (SOLVER.IMS_PER_BATCH / world size)")
    ap.add_argument("--lr", type=float, default=0.001, help="SOLVER.BASE_LR (demo_train_FLIR.py:62)")
    ap.add_argument("--clip-grad-norm", type=float, default=0.0, help="SOLVER.CLIP_GRADIENTS.CLIP_VALUE with CLIP_TYPE norm, NORM_TYPE 2: every parameter tensor is "
                    "clipped by its own norm (solver/build.py:19-36); 0 = off = SOLVER.CLIP_GRADIENTS.ENABLED False, the reference's default")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--depth", type=int, default=101)
    ap.add_argument("--num-classes", type=int, default=3)
    ap.add_argument("--weights", default="", help="state dict (.pth / .pkl / the tests' .npz fixture); default: seeded random")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--world-size", type=int, default=1)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--out", default="")
    ap.add_argument("--resume", default="", help="a file written by --out: model, optimizer state and iteration are taken from it")
    return ap.parse_args(argv)


def main(cmd=None):
    from ..data import resize_shortest_edge_shape
    from ..rcnn import DetectorConfig, GeneralizedRCNN
    from ..synthetic import labelled_frames, synthetic_state_dict
    argv = list(cmd) if cmd is not None else sys.argv[1:]
    args = parse(argv)
    launch.maybe_self_launch(args.world_size, argv, module="proben_amd.cli.train_box_head", device=args.device)
    rank, world, dev = launch.init_distributed(args.device, expect_world=args.world_size)
    cfg = DetectorConfig(num_classes=args.num_classes)      # the depth is read off the state dict
    ckpt = torch.load(args.resume, map_location="cpu") if args.resume else None
    if ckpt is not None:
        sd = ckpt["model"]
    elif args.weights:
        from ..weights import load_state_dict_file
        sd = load_state_dict_file(args.weights)
    else:
        sd = synthetic_state_dict(args.depth, 3, args.num_classes, seed=args.seed)
    model = GeneralizedRCNN(cfg, sd, device=dev)
    tuner = BoxHeadFineTuner(model, lr=args.lr, seed=args.seed, init_from_model=bool(args.weights or args.resume), clip_grad_norm=args.clip_grad_norm)
    first = 0
    if ckpt is not None:
        tuner.load_state_dict(ckpt["optimizer"])
        first = int(ckpt["iteration"])
    new_hw = resize_shortest_edge_shape(512, 640, 800, 1333)
    log, t0 = [], time.time()
    for step in range(first, first + args.steps):
        # every rank draws its own frames (the reference's TrainingSampler shards an infinite stream by rank)
        frames, gts = labelled_frames(args.images_per_step, seed=args.seed * 100003 + step * world + rank + 1)
        losses = tuner.step(torch.from_numpy(frames).to(dev), [torch.from_numpy(b) for b, _ in gts], [torch.from_numpy(c) for _, c in gts],
                            resize_to=new_hw, lr=warmup_lr(args.lr, step, args.warmup))
        log.append(losses)
        if comm.is_main_process() and (step % 20 == 0 or step == first + args.steps - 1):
            print(json.dumps({"step": step, **{k: round(v, 4) for k, v in losses.items()}}), flush=True)
    torch.cuda.synchronize()
    import hashlib
    digest = hashlib.sha256(tuner.head.flat.master.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    backend = torch.distributed.get_backend() if comm.is_distributed() else "none"      # "nccl" = RCCL: what two visible devices must give
    print(json.dumps({"rank": rank, "weights_sha": digest, "last_loss_cls": round(log[-1]["loss_cls"], 6), "backend": backend,
                      "reducer_on_device": bool(tuner.reducer.backend_is_device)}), flush=True)    # every rank: DDP keeps the weights equal
    if comm.is_main_process():
        dt = time.time() - t0
        print(json.dumps({"steps": args.steps, "world_size": world, "images_per_s": round(args.steps * args.images_per_step * world / dt, 1)}))
        if args.out:
            tuner.export()
            full = {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in sd.items()}
            full.update(tuner.reference_state_dict())
            torch.save({"model": full, "optimizer": tuner.state_dict(), "iteration": first + args.steps}, args.out)
    if comm.is_distributed():
        launch.shutdown()
    return log


if __name__ == "__main__":
    main()
