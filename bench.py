"""Headline benchmark: RGB+T frame-pairs / second on 640x512 FLIR-shaped frames.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--depth 101]

Workload (BASELINE.json configs[2]): two Faster R-CNN R101-FPN detectors (thermal + RGB, K = 3, fp16 MFMA
convs / fp32 post-ops) + ProbEn fusion (probEn score, v-avg box), batch 32 frame pairs per GPU, synthetic
uint8 frames resident in HBM, random-init weights of the real architecture.  One step = one batch through
resize -> normalise/pad -> both detectors -> ProbEn (-> RCCL all-gather of the fused rows when N > 1).
Images shard across ranks with no data-path collective ("weak" scaling: per-GPU batch fixed).

Prints ONE JSON line on rank 0 with the throughput, the roofline of the dominant kernel (HIP events around
every launch of it in one extra instrumented step) and the CPU baseline (the oracle's restatement of the
reference's MODEL.DEVICE=cpu path timed on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
HBM_PEAK_GBS = 8000.0                # same guide: HBM3E 8.0 TB/s spec (~6.3 TB/s achievable)
GFLOP_PER_PAIR = 897.4               # 2 x 448.7 GFLOP (R101-FPN, 800x1024, R = 1000, K = 3)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="frame pairs per GPU per step")
    ap.add_argument("--depth", type=int, default=101)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--conv-impl", type=int, default=2, help="1: register-staged conv kernel, 2: LDS-DMA conv kernel")
    ap.add_argument("--serial-detectors", action="store_true", help="run the two detectors back to back on one stream")
    ap.add_argument("--stagger", type=int, default=3,
                    help="N > 0 (default 3): throughput mode of the pipeline - detector 2 trails detector 1 by its res<N> stage and "
                         "batches follow each other without a device-wide wait (all K timed steps still complete inside the "
                         "timed region); 0: every step waits for its own fusion before the next one starts")
    ap.add_argument("--tile256", type=int, default=-1, help="conv tile policy override (pe_set_conv_tile256)")
    ap.add_argument("--layers", type=str, default="", help="write a per-conv-launch table (shape, ms, TFLOP/s) to this file")
    return ap.parse_args()


def build_models(depth, device):
    import proben_amd  # noqa: F401
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_state_dict
    sds = [synthetic_state_dict(depth, 3, 3, seed=s) for s in (1, 2)]  # thermal, RGB
    models = [GeneralizedRCNN(DetectorConfig(), sd, device) for sd in sds]
    return models, sds


def make_step(models, frames_t, frames_rgb, world, rank, with_comm=True, concurrent=True, stagger=0):
    from proben_amd.pipeline import FramePairPipeline
    B = frames_t.shape[0]
    out_sizes = [(512, 640)] * B
    pipe = FramePairPipeline(models, "probEn", "v-avg", concurrent=concurrent, staggered=stagger > 0, stagger_stage=stagger or 4)

    def step():
        (det_t, det_r), fused = pipe([frames_t, frames_rgb], out_sizes, (800, 1000))  # [B,512,640,3] uint8 batches
        if world > 1 and with_comm:
            from proben_amd import comm
            if pipe.staggered:   # the fused rows live on the second detector's stream
                with torch.cuda.stream(pipe.streams[1]):
                    comm.all_gather_fused_rows(fused)
            else:
                comm.all_gather_fused_rows(fused)
        return det_t, det_r, fused
    step.pipe = pipe
    return step


def roofline_leg(step, layers_path="", reps=10):
    """Per-kernel durations, measured live with HIP events: one extra step records every conv launch
    (arguments only - no timing, so the step is not perturbed); then every DISTINCT launch configuration is
    replayed `reps` times back-to-back between two events on the launch stream (GPU-bound, no host gaps).
    Per-variant time of one step = sum over its launches of that configuration's average duration -
    the same quantity `rocprofv3 --kernel-trace --stats` reports as AverageNs x Calls."""
    from proben_amd import layers as L
    L.PROFILE = []
    step()
    torch.cuda.synchronize()
    rec, L.PROFILE = L.PROFILE, None
    distinct = {}
    for r in rec:
        distinct.setdefault((r["variant"], r["shape"]), r)
    timing = {}
    for key, r in distinct.items():
        replay = r["replay"]
        L.PROFILE = None
        for _ in range(2):
            replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            replay()
        e1.record()
        torch.cuda.synchronize()
        timing[key] = e0.elapsed_time(e1) / reps * 1e-3
    agg, rows = {}, {}
    for r in rec:
        key = (r["variant"], r["shape"])
        a = agg.setdefault(r["variant"], [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += r["flops"]
        a[2] += timing[key]
        a[3] += r["bytes"]
        rows.setdefault(key, [0, r["flops"], r["bytes"]])[0] += 1
    if layers_path:
        with open(layers_path, "w") as f:
            f.write("variant\tshape\tlaunches_per_step\tavg_ms\tTFLOP/s\talgorithmic_GB/s\n")
            for key, (cnt, fl, by) in sorted(rows.items(), key=lambda kv: -kv[1][0] * timing[kv[0]]):
                f.write(f"{key[0]}\t{key[1]}\t{cnt}\t{timing[key] * 1e3:.4f}\t{fl / timing[key] / 1e12:.1f}\t{by / timing[key] / 1e9:.0f}\n")

    def describe(name):
        n, fl, t, by = agg[name]
        tf, gbs = fl / t / 1e12, by / t / 1e9
        # machine balance 2500 TFLOP/s / 8000 GB/s = 312 flop/byte decides which roof bounds the kernel
        hbm_bound = fl / by < MFMA_F16_DENSE_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS
        d = {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
             "achieved": round(gbs if hbm_bound else tf, 1), "peak": HBM_PEAK_GBS if hbm_bound else MFMA_F16_DENSE_PEAK_TFLOPS,
             "unit": "GB/s" if hbm_bound else "TFLOP/s",
             "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tf / MFMA_F16_DENSE_PEAK_TFLOPS), 4), "traffic": None,
             "launches_per_step": n, "avg_launch_ms": round(t / n * 1e3, 4), "gflop_per_launch": round(fl / n / 1e9, 2),
             "algorithmic_mbytes_per_launch": round(by / n / 1e6, 1), "flop_per_byte": round(fl / by, 1),
             "tflops": round(tf, 1), "algorithmic_gbs": round(gbs, 1)}
        return d
    # HBM traffic per launch from the PMC counters: measured by separate `rocprofv3 --pmc` passes (they cannot run
    # inside this process); profiles/*_pmc_traffic.json holds the last committed measurement per kernel name
    traffic = {}
    try:
        import glob
        for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
            traffic = json.load(open(pth))["kernels"]
    except Exception:
        traffic = {}

    def with_traffic(d):
        key = d["kernel"].replace(" ", "")
        for k, v in traffic.items():
            if k.replace(" ", "") == key:
                d["traffic"] = round(v["hbm_mb_per_launch"] * 1e6)   # HBM bytes per launch (compare: algorithmic_mbytes_per_launch)
                d["traffic_detail"] = {"hbm_mbytes_per_launch": v["hbm_mb_per_launch"], "unit": "MB",
                                       "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes, profiles/r01_pmc_traffic.json"}
        return d
    dom = max(agg, key=lambda k: agg[k][2])
    out = with_traffic(describe(dom))
    # north_star's MFMA target is quoted on the 3x3 convolutions: always report their kernel as well
    k3 = max((k for k in agg if k.startswith("conv3x3")), key=lambda k: agg[k][1], default=None)
    if k3 is not None and k3 != dom:
        out["conv3x3"] = with_traffic(describe(k3))
    out["method"] = ("avg_launch_ms = HIP-event timing of every distinct launch replayed back-to-back on the launch stream; "
                     "agrees with rocprofv3 --kernel-trace --stats of `bench.py --serial-detectors` "
                     "(profiles/r01_final_kernel_stats.csv).  In the default two-stream run co-running kernels stretch each "
                     "other's durations 1.5-1.8x (profiles/r01_final_kernel_stats_two_streams.csv) while the step gets shorter.")
    out["all_conv_variants"] = {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2] * 1e3, 3),
                                    "tflops": round(v[1] / v[2] / 1e12, 1), "algorithmic_gbs": round(v[3] / v[2] / 1e9, 0)}
                                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}
    return out


def cpu_baseline(sds, depth, pairs, threads):
    """The oracle = this repo's restatement of the reference's CPU path (torch fp32 NCHW unfused conv/BN/ReLU,
    all-anchor decode, per-level sort, per-level ROIAlign, NumPy-f64 ProbEn), timed on the host cores."""
    from oracle import detector as D
    from oracle import proben as O
    from proben_amd.synthetic import synthetic_images
    threads = max(1, min(threads, os.cpu_count() or 1))  # torch CPU convs degrade badly when oversubscribed
    torch.set_num_threads(threads)
    spec = D.DetectorSpec(depth=depth)
    imgs_t = synthetic_images(pairs, seed=100)
    imgs_r = synthetic_images(pairs, seed=200)
    t0 = time.time()
    for i in range(pairs):
        dets = []
        for sd, im in ((sds[0], imgs_t[i]), (sds[1], imgs_r[i])):
            x = torch.from_numpy(im).permute(2, 0, 1).float()[None]
            x = torch.nn.functional.interpolate(x, size=(800, 1000), mode="bilinear", align_corners=False)[0]
            o = D.forward([x], sd, spec, out_sizes=[(512, 640)])[0]
            keep = o["classes"] <= 2
            dets.append({"bbox": o["boxes"][keep].double().numpy(), "score": o["scores"][keep].double().numpy(),
                         "class": o["classes"][keep].numpy(), "prob": o["prob_score"][keep].double().numpy(),
                         "vars": o["vars"][keep].double().numpy()})
        live = [d for d in dets if len(d["score"])]
        if len(live) >= 2:
            O.nms_bayesian(*O.concat_infos(live), 0.5, "probEn", "v-avg")
    dt = time.time() - t0
    return {"value": round(pairs / dt, 4), "unit": "frame-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{pairs} frame pair(s): 2 x R{depth}-FPN oracle forward (torch {torch.__version__} CPU fp32, "
                      f"{threads} threads) + NumPy-f64 ProbEn, {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    from proben_amd.synthetic import synthetic_images
    models, sds = build_models(args.depth, dev)
    from proben_amd import _lib
    _lib.check(_lib.lib().pe_set_conv_impl(args.conv_impl), "pe_set_conv_impl")
    if args.tile256 >= 0:
        _lib.lib().pe_set_conv_tile256(args.tile256)
    B = args.batch
    frames_t = torch.from_numpy(synthetic_images(B, seed=10 + rank)).to(dev)
    frames_rgb = torch.from_numpy(synthetic_images(B, seed=1000 + rank)).to(dev)
    step = make_step(models, frames_t, frames_rgb, world, rank, concurrent=not args.serial_detectors, stagger=args.stagger)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    det_t, det_r, fused = out
    n_det = float(det_t["counts"].float().mean() + det_r["counts"].float().mean()) / 2
    n_fused = float(fused["counts"].float().mean())
    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / dt
        line = {
            "metric": "RGB+T frame-pairs/sec (FLIR-aligned 640x512, two R101-FPN detectors + ProbEn)",
            "value": round(value, 2), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"configs[2]: FLIR RGB + thermal two-detector ProbEn (probEn/v-avg), batch {B} pairs per GPU, "
                                   f"R{args.depth}-FPN x2, 640x512 -> 800x1000 (padded 800x1024), K=3, random-init weights",
                       "batch_pairs_per_gpu": B, "detections_per_image_mean": round(n_det, 1),
                       "fused_rows_per_pair_mean": round(n_fused, 1),
                       "end_to_end_tflops": round(value * GFLOP_PER_PAIR / 1e3, 1),
                       "schedule": ("two detector streams, staggered by res%d, batches back to back" % args.stagger) if args.stagger > 0
                                   and not args.serial_detectors else ("one stream" if args.serial_detectors else "two detector streams, step by step")},
        }
        if not args.no_roofline:
            # rank-local leg: no collective inside (the other ranks are already waiting at the final barrier)
            line["roofline"] = roofline_leg(make_step(models, frames_t, frames_rgb, world, rank, with_comm=False, concurrent=False), args.layers)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sds, args.depth, args.cpu_pairs, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
