"""Headline benchmark: RGB+T frame-pairs / second on 640x512 FLIR-shaped frames.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1|2|3|4] [--batch B] [--feed hbm|host]

Default workload (BASELINE.json configs[2]): two Faster R-CNN R101-FPN detectors (thermal + RGB, K = 3, fp16 MFMA
convs / fp32 post-ops) + ProbEn fusion (probEn score, v-avg box), batch 32 frame pairs per GPU, synthetic uint8
frames, random-init weights of the real architecture.  One step = one batch through resize -> normalise/pad ->
every detector -> ProbEn (-> RCCL all-gather of the fused rows when N > 1).  Images shard across ranks with no
data-path collective ("weak" scaling: per-GPU batch fixed).
  --config 1: configs[1] thermal-only R101-FPN, batch 16           (metric unit: frames/s)
  --config 3: configs[3] thermal (3-ch) + early (4-ch) + middle (6-ch, two backbone passes, 512-ch heads) + 3-way ProbEn
  --config 4: configs[4] KAIST: two R50-FPN detectors, K = 1, binary ProbEn
  --feed hbm (default): the frames are resident in HBM before the timed region (the metric's definition);
  --feed host: frames start in pinned host memory and a double-buffered uploader moves every batch over PCIe INSIDE
               the timed loop (reported as such; never the headline value).

Multi-GPU: `python bench.py --gpus N` launches N ranks itself (torch.distributed.run, one process per GPU, backend
nccl = RCCL) when it is not already running under a launcher; `n_gpus` in the line is the world size RCCL reports.

Prints ONE JSON line on rank 0 with the throughput, the roofline of the dominant kernel (HIP events around every
distinct launch replayed back-to-back), the ProbEn kernel's B = 4096 micro-figure and the CPU baseline (the oracle's
restatement of the reference's MODEL.DEVICE=cpu path timed on the host cores, all cores and 1 thread, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
HBM_PEAK_GBS = 8000.0                # same guide: HBM3E 8.0 TB/s spec (~6.3 TB/s achievable)

# workloads: detectors = input channels per detector; GFLOP per unit from SURVEY 8(d)
CONFIGS = {
    1: dict(title="configs[1]: FLIR thermal-only R101-FPN inference", detectors=(3,), depth=101, K=3, batch=16,
            fuse=None, unit="frames/s", gflop=448.7),
    2: dict(title="configs[2]: FLIR RGB + thermal two-detector ProbEn (probEn/v-avg)", detectors=(3, 3), depth=101, K=3,
            batch=32, fuse=("probEn", "v-avg"), unit="frame-pairs/s", gflop=897.4),
    3: dict(title="configs[3]: FLIR thermal + early (4-ch) + middle (6-ch) three-model ProbEn (probEn/v-avg)",
            detectors=(3, 4, 6), depth=101, K=3, batch=32, fuse=("probEn", "v-avg"), unit="frame-pairs/s", gflop=None),
    4: dict(title="configs[4]: KAIST RGB + thermal ProbEn (binary probEn/v-avg), R50-FPN x2, K=1", detectors=(3, 3), depth=50,
            K=1, batch=32, fuse=("probEn_binary", "v-avg"), unit="frame-pairs/s", gflop=None),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default 250 = ~10 s of device work at ~40 ms / step)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="frame pairs per GPU per step (0 = the config's own)")
    ap.add_argument("--depth", type=int, default=0, help="override the backbone depth (50 | 101)")
    ap.add_argument("--feed", choices=("hbm", "host"), default="hbm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=8)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--no-wd", action="store_true", help="A/B: do not use the weights-direct 3x3 kernel")
    ap.add_argument("--no-tail-fusion", action="store_true", help="A/B: run conv2 and conv3 of the res4 bottlenecks as two launches")
    ap.add_argument("--no-res2-fusion", action="store_true", help="A/B: run res2 as separate conv launches instead of the fused 64-wide chain")
    ap.add_argument("--serial-detectors", action="store_true", help="run the detectors back to back on one stream")
    ap.add_argument("--wd9-wgs", type=int, default=0, help="A/B: workgroups of the persistent pure 3x3 kernel (0 = the library's default)")
    ap.add_argument("--no-power", action="store_true", help="do not poll rocm-smi for board power during the timed region")
    ap.add_argument("--conv-policy", type=int, default=-1, help="A/B: tile_bits of pe_test_set_conv_policy (csrc/test_hooks.h; default 329)")
    ap.add_argument("--roi-sort", type=int, default=1, help="0: ROIAlign takes the proposals in RPN order (A/B; identical results)")
    ap.add_argument("--roi-fast", type=int, default=1, help="0: ROIAlign's per-lane form instead of the wave-uniform form (A/B; identical results)")
    ap.add_argument("--nms-presorted", type=int, default=1, help="0: batched NMS always runs its sorting network (A/B; identical results)")
    ap.add_argument("--wd9-mode", type=int, default=-1,
                    help="A/B (csrc/test_hooks.h): 0 = two-wave weights-direct kernels only (csrc/conv_wd.h), 1 = persistent one-wave-per-SIMD "
                         "kernel for the pure 3x3 launches, 8 = for the fused RPN head, 9 = both; -1 = the library's default (9)")
    ap.add_argument("--stagger", type=int, default=3,
                    help="N > 0 (default 3, two-detector configs): throughput mode of the pipeline - detector 2 trails detector 1 by its "
                         "res<N> stage and batches follow each other without a device-wide wait (all K timed steps still complete "
                         "inside the timed region); 0: every step waits for its own fusion before the next one starts")
    ap.add_argument("--layers", type=str, default="", help="write a per-conv-launch table (shape, ms, TFLOP/s) to this file")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# multi-rank launch (replaces the reference's detectron2/engine/launch.py:24-84 mp.spawn + init_process_group)
# ----------------------------------------------------------------------------------------------------------------
def launch_command(argv, nproc, port, script=None):
    """The command `bench.py --gpus N` re-executes itself with when it is not already under a launcher."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_self_launch(args, argv):
    """--gpus N > 1 without RANK / WORLD_SIZE in the environment: spawn the N ranks and exit with their status."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to report a "
                 f"{args.gpus}-GPU line from fewer devices")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL / cross-process tensor sharing on this driver)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    sys.exit(subprocess.call(launch_command(argv, args.gpus, free_port()), env=env))


# ----------------------------------------------------------------------------------------------------------------
def build_models(cfg, depth, device, use_wd=True, fuse_tails=True, fuse_res2=True):
    import proben_amd  # noqa: F401
    from proben_amd.rcnn import DetectorConfig, GeneralizedRCNN
    from proben_amd.synthetic import synthetic_state_dict
    fmt = {3: "BGR", 4: "BGRT", 6: "BGRTTT"}
    models, sds = [], []
    for i, ch in enumerate(cfg["detectors"]):
        sd = synthetic_state_dict(depth, cfg["K"], ch, seed=i + 1)
        mean = (103.53, 116.28, 123.675) + (135.438,) * (ch - 3)
        m = GeneralizedRCNN(DetectorConfig(num_classes=cfg["K"], input_format=fmt[ch], pixel_mean=mean, pixel_std=(1.0,) * ch), sd, device)
        m.use_wd = use_wd
        m.fuse_tails = fuse_tails
        m.fuse_res2 = fuse_res2
        models.append(m)
        sds.append(sd)
    return models, sds


def make_frames(cfg, B, rank, device=None, pinned=False):
    """One uint8 [B,512,640,C] batch per detector (synthetic FLIR-shaped frames)."""
    import torch
    from proben_amd.synthetic import synthetic_images
    out = []
    for i, ch in enumerate(cfg["detectors"]):
        t = torch.from_numpy(synthetic_images(B, channels=ch, seed=10 + 990 * i + rank))
        out.append(t.pin_memory() if pinned else t.to(device))
    return out


def comm_active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def count_collectives():
    """Wraps torch.distributed's tensor collectives with a counter (bench.py only: the line reports how many the timed loop issued
    per step; the product documents ONE).  Returns the counter dict {"n": calls so far}."""
    import torch.distributed as dist
    c = {"n": 0}
    for name in ("all_gather_into_tensor", "all_gather", "all_reduce", "broadcast", "reduce_scatter_tensor", "all_to_all_single", "gather", "reduce"):
        fn = getattr(dist, name, None)
        if fn is None or getattr(fn, "_counted", False):
            continue

        def wrap(*a, _fn=fn, **k):
            c["n"] += 1
            return _fn(*a, **k)
        wrap._counted = True
        setattr(dist, name, wrap)
    return c


def make_step(models, frames, cfg, world, with_comm=True, concurrent=True, stagger=0, feeder=None):
    import torch
    from proben_amd.pipeline import FramePairPipeline
    B = frames[0].shape[0]
    out_sizes = [(512, 640)] * B
    fuse = cfg["fuse"] or ("probEn", "v-avg")
    pipe = FramePairPipeline(models, fuse[0], fuse[1], concurrent=concurrent, staggered=stagger > 0 and len(models) == 2,
                             stagger_stage=stagger or 4, fuse=cfg["fuse"] is not None)

    def step():
        batch = feeder.next() if feeder is not None else frames
        dets, fused = pipe(batch, out_sizes, (800, 1000))
        if feeder is not None:
            feeder.mark_consumed(*(pipe.streams or []))
        if (world > 1 or comm_active()) and with_comm:
            from proben_amd import comm
            payload = fused if fused is not None else {k: dets[0][k] for k in ("boxes", "scores", "classes", "counts")}
            if pipe.staggered:   # the fused rows live on the second detector's stream
                with torch.cuda.stream(pipe.streams[1]):
                    comm.all_gather_fused_rows(payload)
            else:
                comm.all_gather_fused_rows(payload)
        return dets, fused
    step.pipe = pipe
    return step


def roofline_leg(step, layers_path="", reps=10):
    """Per-kernel durations, measured live with HIP events: one extra step records every conv launch
    (arguments only - no timing, so the step is not perturbed); then every DISTINCT launch configuration is
    replayed `reps` times back-to-back between two events on the launch stream (GPU-bound, no host gaps).
    Per-variant time of one step = sum over its launches of that configuration's average duration -
    the same quantity `rocprofv3 --kernel-trace --stats` reports as AverageNs x Calls."""
    import torch
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
        for _ in range(2):
            replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            replay()
        e1.record()
        torch.cuda.synchronize()
        timing[key] = e0.elapsed_time(e1) / reps * 1e-3
    def label(r):
        """bench label of a launch: the kernel name, with the backbone's own 3x3 launches (res3: 128 -> 128, res5: 512 -> 512) told apart
        from the FPN / RPN 3x3 launches that run on the same kernels"""
        sh = r["shape"]
        tag = " @res3" if "Cin128 Cout128 k3" in sh else " @res5" if "Cin512 Cout512 k3" in sh else " @box_head" if " 1x1 Cin" in sh else ""
        return r["variant"] + tag
    agg, rows = {}, {}
    for r in rec:
        key = (r["variant"], r["shape"])
        a = agg.setdefault(label(r), [0, 0.0, 0.0, 0.0])
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
        return {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
                "achieved": round(gbs if hbm_bound else tf, 1), "peak": HBM_PEAK_GBS if hbm_bound else MFMA_F16_DENSE_PEAK_TFLOPS,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round((gbs / HBM_PEAK_GBS) if hbm_bound else (tf / MFMA_F16_DENSE_PEAK_TFLOPS), 4), "traffic": None,
                "launches_per_step": n, "avg_launch_ms": round(t / n * 1e3, 4), "gflop_per_launch": round(fl / n / 1e9, 2),
                "algorithmic_mbytes_per_launch": round(by / n / 1e6, 1), "flop_per_byte": round(fl / by, 1),
                "tflops": round(tf, 1), "algorithmic_gbs": round(gbs, 1)}
    # HBM traffic per launch from the PMC counters and the in-network launch durations of rocprofv3: measured by separate
    # `rocprofv3` runs (they cannot run inside this process; scripts/collect_profiles.sh) and committed under profiles/.  ONE round's
    # record is used: profiles/current_pmc_traffic.json if present, else the highest round's `rNN_pmc_traffic.json` (the round's FINAL
    # record - mid-round files like `rNN_<tag>_pmc_traffic.json` are never picked), with the `rNN_kernel_stats.csv` of the same round.
    import csv
    import re
    prof = os.path.join(ROOT, "profiles")
    traffic, traffic_src, rocprof, rocprof_src = {}, "", {}, ""
    try:
        cur = os.path.join(prof, "current_pmc_traffic.json")
        cands = [f for f in os.listdir(prof) if re.fullmatch(r"r\d+_pmc_traffic\.json", f)]
        pick = cur if os.path.exists(cur) else (os.path.join(prof, max(cands, key=lambda f: int(f[1:].split("_")[0]))) if cands else "")
        if pick:
            traffic_src = os.path.basename(pick)
            traffic = {k.replace(" ", ""): v for k, v in json.load(open(pick))["kernels"].items()}
            m = re.match(r"(r\d+)_", traffic_src)
            stats = os.path.join(prof, (m.group(1) if m else "current") + "_kernel_stats.csv")
            if os.path.exists(stats):
                rocprof_src = os.path.basename(stats)
                for row in csv.DictReader(open(stats)):
                    rocprof[row["Name"].replace(" ", "").replace("wd::", "").replace("wd9::", "")] = (int(row["Calls"]), float(row["TotalDurationNs"]))
    except Exception:
        traffic, rocprof = {}, {}

    def same_template(table, kernel):
        """entries of `table` for this bench label: the exact rocprof name, else every instantiation of the same kernel template (the bench
        groups the wd9 launches of all image widths under one label; rocprof names them per template argument; kernels in an anonymous
        namespace appear mangled)"""
        kernel = kernel.split(" @")[0]
        key = kernel.replace(" ", "")
        if key in table:
            return [table[key]]
        base = kernel.split("<")[0]
        return [v for k, v in table.items() if k.split("<")[0].split("::")[-1] == base or (base + "E") in k or k.split("(")[0].endswith(base)]

    def with_traffic(d, kernels=None):
        """adds the counter traffic per launch and rocprof's in-network average duration (launch-weighted over the label's kernels)"""
        names = kernels or [d["kernel"]]
        tv = [v for n in names for v in same_template(traffic, n)]
        if tv:
            n = sum(v.get("launches", 1) for v in tv)
            mb = round(sum(v["hbm_mb_per_launch"] * v.get("launches", 1) for v in tv) / n, 1)
            d["traffic"] = round(mb * 1e6)                        # HBM bytes per launch (compare: algorithmic_mbytes_per_launch)
            d["traffic_detail"] = {"hbm_mbytes_per_launch": mb, "unit": "MB", "vs_algorithmic": round(mb / d["algorithmic_mbytes_per_launch"], 3),
                                   "source": f"rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes, profiles/{traffic_src}"}
        rv = [v for n in names for v in same_template(rocprof, n)]
        if rv:
            calls, total = sum(v[0] for v in rv), sum(v[1] for v in rv)
            # rocprof knows kernel NAMES, the bench labels also tell shapes apart ("... @box_head", "... @res5"): launches of the same kernel
            # that belong to OTHER labels are taken out of the name's total at their replay time (the only per-shape time there is)
            base = {n.split(" @")[0] for n in names}
            others = [k for k in agg if k not in names and k.split(" @")[0] in base and not k.startswith("resnet") and not k.startswith("1x1 class")]
            own = sum(agg[n][0] for n in names if n in agg)
            note = ""
            if others and own:
                steps_prof = calls / (own + sum(agg[k][0] for k in others))
                total -= sum(agg[k][2] for k in others) * 1e9 * steps_prof
                calls -= sum(agg[k][0] for k in others) * steps_prof
                note = "; launches of the same kernel under other labels (" + ", ".join(others) + ") removed at their replay time"
            avg = total / calls * 1e-9
            work = d["gflop_per_launch"] * 1e9 if d["bound"] == "mfma" else d["algorithmic_mbytes_per_launch"] * 1e6
            rate = work / avg / (1e12 if d["bound"] == "mfma" else 1e9)
            d["in_network"] = {"avg_launch_ms": round(avg * 1e3, 4), "achieved": round(rate, 1), "frac": round(rate / d["peak"], 4),
                               "source": f"rocprofv3 --kernel-trace --stats of `bench.py --serial-detectors`, profiles/{rocprof_src} ({round(calls)} launches{note})"}
        return d

    def describe_group(label, members):
        """one roofline record over several bench labels (launch-weighted: total work / total time)"""
        n = sum(agg[k][0] for k in members)
        fl, t, by = (sum(agg[k][i] for k in members) for i in (1, 2, 3))
        agg[label] = [n, fl, t, by]
        d = with_traffic(describe(label), kernels=members)
        del agg[label]
        d["members"] = {k: {"launches": agg[k][0], "ms": round(agg[k][2] * 1e3, 3), "tflops": round(agg[k][1] / agg[k][2] / 1e12, 1),
                            "algorithmic_gbs": round(agg[k][3] / agg[k][2] / 1e9, 0)} for k in members}
        return d

    dom = max(agg, key=lambda k: agg[k][2])
    out = with_traffic(describe(dom))
    # north_star's MFMA target is quoted on "the ResNet-101 3x3 convs": the kernels that CONTAIN the bottom-up 3x3 convolutions - res2 as
    # fused 64-wide chains (bneck64: 1x1 + 3x3 + 1x1 per launch), res3's kw-reuse kernel, the fused res4 tail (3x3 + conv3 + shortcut)
    # and res5's weights-direct kernel - with their flops (whole launches, i.e. including the fused 1x1 halves), their time and the
    # launch-weighted fraction of the fp16 MFMA peak
    bottom_up = [k for k in agg if k.startswith("bneck64_kernel") or k.endswith(", 0, 2>") or k.startswith("conv3x3_wd9_tail") or " @res" in k]
    if bottom_up:
        d = describe_group("resnet bottom-up 3x3 (res2 chains, res3, res4 tail, res5)", bottom_up)
        d.update({"bound": "mfma", "achieved": d["tflops"], "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(d["tflops"] / MFMA_F16_DENSE_PEAK_TFLOPS, 4)})
        if "in_network" in d:
            # per member: its flops at rocprof's average duration of its kernel NAME (res5 shares a name with the p5 output conv: 2 of 8
            # launches of that name are not the backbone's)
            tot = sum(agg[k][0] * (sum(v[1] for v in same_template(rocprof, k)) / max(sum(v[0] for v in same_template(rocprof, k)), 1)) for k in bottom_up) * 1e-9
            tf = sum(agg[k][1] for k in bottom_up) / tot / 1e12
            d["in_network"].update({"avg_launch_ms": round(tot / sum(agg[k][0] for k in bottom_up) * 1e3, 4), "achieved": round(tf, 1),
                                    "frac": round(tf / MFMA_F16_DENSE_PEAK_TFLOPS, 4)})
        out["resnet3x3"] = d
    # the FPN output convolutions and the RPN head's 3x3 (pure 3x3 launches; "..., 1>" = 3x3 + RPN head)
    pure = [k for k in agg if k.startswith("conv3x3_wd") and k.endswith(", 0>")] or [k for k in agg if k.startswith("conv3x3") and k not in bottom_up]
    k3 = max(pure, key=lambda k: agg[k][1], default=None)
    if k3 is not None and k3 != dom:
        out["conv3x3"] = with_traffic(describe(k3))
    # the HBM-bound 1x1 class: every 1x1 launch below the machine balance (conv_igemm2's 128-wide tiles and the persistent ring kernel;
    # the box head's long-K GEMMs are MFMA-bound and listed on their own)
    ring_hbm = [k for k in agg if k == "conv1x1_ring_kernel"]
    one = [k for k in agg if k.startswith("conv_igemm2_kernel<128, 128") and " @" not in k] + ring_hbm
    if one:
        out["conv1x1"] = describe_group("1x1 class (conv_igemm2<128,128> + conv1x1_ring)", one)
    if ring_hbm and ring_hbm[0] != dom:
        out["conv1x1_ring"] = with_traffic(describe(ring_hbm[0]))
    # ... and the fused res4 bottleneck tail
    kt = max((k for k in agg if k.endswith(", 0, 2>") or k.startswith("conv3x3_wd9_tail")), key=lambda k: agg[k][2], default=None)
    if kt is not None and kt != dom:
        out["res4_tail"] = with_traffic(describe(kt))
    out["method"] = ("avg_launch_ms / achieved / frac = HIP-event timing of every distinct launch replayed back-to-back on the launch stream: "
                     "WARM-cache figures (a launch's inputs may still sit in the 256 MiB Infinity Cache from its previous replay; inside the "
                     "network they were just written by the producer or are cold) - `in_network` holds rocprofv3's average duration of the "
                     "same kernels inside `bench.py --serial-detectors` from the committed profile of the round (" + (rocprof_src or "none committed yet") +
                     "), typically 3-8 % slower.  In the default two-stream run co-running kernels stretch each other's durations while the step gets shorter.")
    out["all_conv_variants"] = {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "ms": round(v[2] * 1e3, 3),
                                    "tflops": round(v[1] / v[2] / 1e12, 1), "algorithmic_gbs": round(v[3] / v[2] / 1e9, 0)}
                                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}
    return out


def proben_micro(device, B=4096, reps=20):
    """SURVEY 8(d) config-3 micro-benchmark of the fused ProbEn kernel: B = 4096 images, n1, n2 ~ U{0..100}, 30 % cross-detector
    near-duplicates, K = 3, seed 2; algorithmic bytes = N*(4+1+K+1)*8 + N*4 in, M*(4+1+1)*4 out."""
    import numpy as np
    import torch
    from proben_amd import fusion as F
    from proben_amd.synthetic import synth_detections
    per_image = synth_detections(B, seed=2, kdet=2, nmax=100, K=3)
    b, s, p, v, c, offs = F.pack_infos(per_image, device)
    nmax = int((offs[1:] - offs[:-1]).max().item())
    out = F.fuse_batch(b, s, p, v, c, offs, "probEn", "v-avg", max_rows=nmax)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = F.fuse_batch(b, s, p, v, c, offs, "probEn", "v-avg", max_rows=nmax)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) / reps * 1e-3
    n_in, n_out = int(b.shape[0]), int(out["counts"].clamp(min=0).sum().item())
    nbytes = n_in * ((4 + 1 + 3 + 1) * 8 + 4) + n_out * (4 + 1 + 1) * 4
    return {"images": B, "rows_in_mean": round(n_in / B, 1), "rows_out_mean": round(n_out / B, 1), "ms_per_launch": round(sec * 1e3, 4),
            "us_per_image": round(sec / B * 1e6, 4), "images_per_s": round(B / sec), "algorithmic_gbs": round(nbytes / sec / 1e9, 2),
            "note": "one 1024-thread workgroup per image (pair tests into LDS bit matrices, one wave walks the rows, a thread per cluster), float64; "
                    "latency / LDS bound, not HBM bound (the whole batch is " + str(round(nbytes / 1e6, 1)) + " MB)"}


class BoardPower:
    """rocm-smi polled by ONE helper process (started before the timed region: no fork from this process, no Python thread, while the
    steps are being timed) on rank 0's device: board power and shader clock.  The hot kernels of this path run AT the board's power cap
    (profiles/r04_power_kernels.txt): the cap, not the nominal MFMA peak, is what bounds them, so the line reports it.
    Best effort: no rocm-smi -> {"available": false}."""

    def __init__(self, device_index):
        self.dev, self.proc, self.cap, self.path = str(int(device_index)), None, None, None

    @staticmethod
    def _numbers(card):
        import re
        return {k: float(re.search(r"[-+]?\d+(\.\d+)?", str(v)).group(0)) for k, v in card.items() if re.search(r"\d", str(v))}

    def start(self):
        import subprocess
        import tempfile
        try:
            out = subprocess.run(["rocm-smi", "-d", self.dev, "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20).stdout
            card = next(v for v in json.loads(out[out.index("{"):]).values() if isinstance(v, dict))
            self.cap = next(v for k, v in self._numbers(card).items() if "Power" in k)
            fd, self.path = tempfile.mkstemp(prefix="bench_power_", suffix=".txt")
            loop = f"while true; do date +%s.%N; rocm-smi -d {self.dev} --showpower --showclocks --json 2>/dev/null; done"
            self.proc = subprocess.Popen(["bash", "-c", loop], stdout=fd, stderr=subprocess.DEVNULL, start_new_session=True)
            os.close(fd)
        except Exception:
            self.proc = None

    def result(self, w0, w1, units, windows=None):
        """w0 / w1: time.time() at the two ends of the timed region; windows: [(start, end)] wall-clock sub-windows of it -> the median
        shader clock and board power of the samples inside each (None where no sample fell)."""
        import signal
        if self.proc is None:
            return {"available": False, "cap_w": self.cap}
        try:
            os.killpg(self.proc.pid, signal.SIGTERM)     # the helper's own process group (start_new_session), nothing else
        except Exception:
            pass
        rows, stamp = [], None
        try:
            for ln in open(self.path):
                ln = ln.strip()
                if ln and ln[0].isdigit():
                    stamp = float(ln)
                elif ln.startswith("{") and stamp is not None:
                    try:
                        c = self._numbers(next(v for v in json.loads(ln).values() if isinstance(v, dict)))
                        rows.append((stamp, next(v for k, v in c.items() if "Power" in k), c.get("sclk clock speed:")))
                    except Exception:
                        pass
            os.unlink(self.path)
        except Exception:
            pass
        all_rows = rows
        rows = [r for r in rows if w0 <= r[0] and r[0] + 0.1 <= w1]     # a sample is taken shortly AFTER its stamp
        if not rows:
            return {"available": False, "cap_w": self.cap}
        med = lambda v: sorted(v)[len(v) // 2] if v else None
        per_window = None
        if windows:
            per_window = {"sclk_mhz": [med([r[2] for r in all_rows if a <= r[0] + 0.05 < b and r[2] is not None]) for a, b in windows],
                          "board_w": [med([r[1] for r in all_rows if a <= r[0] + 0.05 < b]) for a, b in windows]}
        w = med([r[1] for r in rows])
        clk = [r[2] for r in rows if r[2] is not None]
        return {"available": True, "board_w_median": w, "board_w_max": max(r[1] for r in rows), "cap_w": self.cap,
                "sclk_mhz_median": med(clk) if clk else None, "samples": len(rows), "per_window": per_window,
                "joules_per_unit": round(w * (w1 - w0) / units, 3),
                "source": "rocm-smi --showpower --showclocks polled by a helper process during the timed region (socket package power)"}


def cpu_baseline(sds, cfg, depth, pairs, threads):
    """The oracle = this repo's restatement of the reference's CPU path (torch fp32 NCHW unfused conv/BN/ReLU,
    all-anchor decode, per-level sort, per-level ROIAlign, NumPy-f64 ProbEn), timed on the host cores: all `threads`
    on `pairs` units, then ONE thread on one unit (SURVEY 8d asks for both)."""
    import numpy as np
    import torch
    from oracle import detector as D
    from oracle import proben as O
    from proben_amd.synthetic import synthetic_images
    chans = cfg["detectors"]
    specs = [D.DetectorSpec(depth=depth, in_channels=ch, num_classes=cfg["K"]) for ch in chans]

    def run(n_units, nthreads):
        torch.set_num_threads(nthreads)
        imgs = [synthetic_images(n_units, channels=ch, seed=100 * (i + 1)) for i, ch in enumerate(chans)]
        t0 = time.time()
        for u in range(n_units):
            dets = []
            for sd, spec, im in zip(sds, specs, imgs):
                if im[u].shape[2] == 3:      # 3-channel frames: the reference resizes through Pillow (transform.py:92-97), like the GPU path
                    from PIL import Image
                    x = torch.from_numpy(np.array(Image.fromarray(im[u]).resize((1000, 800), Image.BILINEAR))).permute(2, 0, 1).float().contiguous()
                else:                        # 4- / 6-channel fusion inputs: OpenCV's float bilinear rule (transform.py:82-91)
                    from oracle.resize import cv2_linear_resize_f64
                    x = torch.from_numpy(np.ascontiguousarray(cv2_linear_resize_f64(im[u].astype(np.float64), 800, 1000))).permute(2, 0, 1).float().contiguous()
                o = D.forward([x], sd, spec, out_sizes=[(512, 640)])[0]
                keep = o["classes"] <= 2
                dets.append({"bbox": o["boxes"][keep].double().numpy(), "score": o["scores"][keep].double().numpy(),
                             "class": o["classes"][keep].numpy(), "prob": o["prob_score"][keep].double().numpy(),
                             "vars": o["vars"][keep].double().numpy()})
            live = [d for d in dets if len(d["score"])]
            if cfg["fuse"] is not None and len(live) >= 2:
                O.nms_bayesian(*O.concat_infos(live), 0.5, cfg["fuse"][0].replace("probEn_binary", "probEn_binary"), cfg["fuse"][1])
        return time.time() - t0
    threads = max(1, min(threads, os.cpu_count() or 1))  # torch CPU convs degrade badly when oversubscribed
    dt = run(pairs, threads)
    dt1 = run(1, 1)
    what = f"{len(chans)} x R{depth}-FPN oracle forward (torch {torch.__version__} CPU fp32) + NumPy-f64 ProbEn"
    return {"value": round(pairs / dt, 4), "unit": cfg["unit"], "cores": threads, "kind": "port",
            "sample": f"{pairs} unit(s): {what}, {threads} threads, {dt:.1f} s",
            "single_thread": {"value": round(1 / dt1, 4), "unit": cfg["unit"], "cores": 1, "sample": f"1 unit, 1 thread, {dt1:.1f} s"}}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    maybe_self_launch(args, argv)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("PROBEN_FORCE_DIST") == "1" and "RANK" in os.environ   # one-rank RCCL group (exercises the nccl path on a one-GPU box)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        world = dist.get_world_size()   # the line reports what RCCL saw
    cfg = CONFIGS[args.config]
    depth = args.depth or cfg["depth"]
    B = args.batch or cfg["batch"]
    if args.wd9_wgs > 0:
        from proben_amd import _lib
        _lib.test_hooks().pe_test_set_wd9_wgs(args.wd9_wgs, 0)
    if not args.roi_sort:
        from proben_amd import layers as _layers
        _layers.ROI_SORT = False
    if not args.nms_presorted:
        from proben_amd import _lib
        _lib.test_hooks().pe_test_set_nms_presorted(0)
    if args.roi_fast != 1:
        from proben_amd import _lib
        _lib.test_hooks().pe_test_set_roi_fast(args.roi_fast)
    if args.conv_policy >= 0:
        from proben_amd import _lib
        _lib.test_hooks().pe_test_set_conv_policy(args.conv_policy, 1)
    if args.wd9_mode >= 0:
        from proben_amd import _lib
        _lib.test_hooks().pe_test_set_wd9_mode(args.wd9_mode)
    models, sds = build_models(cfg, depth, dev, use_wd=not args.no_wd, fuse_tails=not args.no_tail_fusion, fuse_res2=not args.no_res2_fusion)
    feeder = None
    if args.feed == "host":
        from proben_amd.pipeline import HostFeeder
        frames = make_frames(cfg, B, rank, pinned=True)
        feeder = HostFeeder(frames, dev)
        frames_dev = [f.to(dev) for f in frames]
    else:
        frames = frames_dev = make_frames(cfg, B, rank, dev)
    stagger = args.stagger if len(models) == 2 and not args.serial_detectors else 0
    step = make_step(models, frames_dev, cfg, world, concurrent=not args.serial_detectors, stagger=stagger, feeder=feeder)

    def fence():
        if world > 1 or comm_active():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    power = BoardPower(local) if rank == 0 and not args.no_power else None
    if power is not None:
        power.start()
    # The timed region is cut into up to 5 equal sub-windows by EVENTS (recorded on every stream the step launches on; no
    # synchronisation inside the region): `value_windows` in the line shows whether a run-to-run difference is a clock ramp at the
    # start of the region or the box.
    streams = [torch.cuda.current_stream()] + list(getattr(step.pipe, "streams", None) or [])
    n_win = min(5, args.steps)
    bounds = [round(i * args.steps / n_win) for i in range(n_win + 1)]
    marks = []

    def mark():
        evs = [torch.cuda.Event(enable_timing=True) for _ in streams]
        for e, st in zip(evs, streams):
            e.record(st)
        marks.append(evs)
    calls = count_collectives() if (world > 1 or comm_active()) else None
    fence()
    w0, t0 = time.time(), time.perf_counter()
    mark()
    if calls is not None:
        calls["n"] = 0
    for i in range(args.steps):
        out = step()
        if i + 1 in bounds[1:]:
            mark()
    n_coll = calls["n"] if calls is not None else 0
    fence()
    dt = time.perf_counter() - t0
    w1 = time.time()
    # boundary b's offset from the start of the region = the LATEST of the streams' events (ms)
    offs = [max(m0.elapsed_time(m) for m0, m in zip(marks[0], evs)) for evs in marks]
    win_ms = [offs[i + 1] - offs[i] for i in range(n_win)]
    win_steps = [bounds[i + 1] - bounds[i] for i in range(n_win)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    dets, fused = out
    from proben_amd.fusion import check_candidate_overflow
    for d in dets:           # outside the timed region: a box head that ran out of candidate slots would have dropped detections
        check_candidate_overflow(d)
    n_det = float(sum(d["counts"].float().mean() for d in dets)) / len(dets)
    if rank == 0:
        units = world * B * args.steps
        value = units / dt
        sched = ("one stream" if args.serial_detectors else
                 ("two detector streams, staggered by res%d, batches back to back" % stagger) if stagger > 0 else
                 "%d detector stream(s), step by step" % len(models))
        line = {
            "metric": ("RGB+T frame-pairs/sec (FLIR-aligned 640x512, two R101-FPN detectors + ProbEn)" if args.config == 2 else
                       "frames or frame-pairs/sec, 640x512, " + cfg["title"]),
            "value": round(value, 2), "unit": cfg["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "value_windows": [round(world * B * n / (ms * 1e-3), 1) if ms > 0 else None for n, ms in zip(win_steps, win_ms)],
            "window_steps": win_steps,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{cfg['title']}, batch {B} per GPU, R{depth}-FPN x{len(models)}, 640x512 -> 800x1000 (padded 800x1024), "
                                   f"K={cfg['K']}, random-init weights",
                       "batch_per_gpu": B, "detections_per_image_mean": round(n_det, 1),
                       "fused_rows_per_pair_mean": round(float(fused["counts"].float().mean()), 1) if fused is not None else None,
                       "end_to_end_tflops": round(value * cfg["gflop"] / 1e3, 1) if cfg["gflop"] else None,
                       "input_residency": ("frames resident in HBM before the timed region (no H2D inside it)" if feeder is None else
                                           "frames in pinned host memory; double-buffered H2D upload of every batch INSIDE the timed region"),
                       "timed_seconds": round(dt, 2), "schedule": sched,
                       "persistent_kernels": ("csrc/conv_wd9.h (pure 3x3 and the fused RPN head, one 512-register workgroup per CU) takes the 256 -> 256 launches of >= 128 tiles; "
                                              "csrc/conv1x1_ring.hip (loader / consumer 1x1 kernel, one 160-KiB workgroup per CU) takes the stride-1 residual-free 1x1 "
                                              "layers with K >= 512 and Cout % 256 == 0 (res4 / res5 conv1, top lateral, fc1, fc2)")},
        }
        if power is not None:
            line["power"] = power.result(w0, w1, B * args.steps,     # rank 0's board, rank 0's units
                                         windows=[(w0 + offs[i] * 1e-3, w0 + offs[i + 1] * 1e-3) for i in range(n_win)])
            pw = (line["power"] or {}).get("per_window")
            if pw:
                line["sclk_mhz_windows"] = pw["sclk_mhz"]
        if world > 1 or comm_active():
            line["config"]["collective"] = "one all_gather_into_tensor of the fused rows per step (RCCL)" + ("" if world > 1 else
                                           "; PROBEN_FORCE_DIST: a ONE-rank RCCL group, the collective runs but moves nothing between devices")
            line["config"]["collective_calls_per_step"] = n_coll / args.steps     # counted at torch.distributed's entry points inside the timed loop
        if not args.no_roofline:
            # rank-local leg: no collective inside (the other ranks are already waiting at the final barrier)
            line["roofline"] = roofline_leg(make_step(models, frames_dev, cfg, world, with_comm=False, concurrent=False), args.layers)
        if not args.no_micro:
            line["proben_micro"] = proben_micro(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sds, cfg, depth, args.cpu_pairs, args.cpu_threads)
        print(json.dumps(line), flush=True)
    if world > 1 or comm_active():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
