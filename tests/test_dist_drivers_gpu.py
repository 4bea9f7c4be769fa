"""VERDICT r02 item 3 on hardware: `python -m proben_amd.cli.<driver> --world-size 2` with the REAL detectors.
Two devices visible -> ranks over RCCL ("nccl"), one per GPU.  One device visible -> (a) the drivers must refuse to fake two
ranks, and (b) the same sharded code path runs with PROBEN_DIST_BACKEND=gloo, where the two ranks share the GPU and only the
collectives differ (host tensors instead of RCCL) - the detector, the shard rule, the row tables, the rank-0 evaluation and
writers are the ones a 2-GPU run uses.  Either way the gathered result must equal the single-rank result bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(two_devices):
    import proben_amd  # noqa: F401
    from proben_amd import launch
    env = launch.launch_env()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PROBEN_DIST_BACKEND"):
        env.pop(k, None)
    if not two_devices:
        env["PROBEN_DIST_BACKEND"] = "gloo"
    return env


def _cli(module, args, env, expect_ok=True):
    p = subprocess.run([sys.executable, "-m", "proben_amd.cli." + module] + args, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if expect_ok:
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return p


def test_world_size_2_equals_world_size_1_on_the_flir_drivers(tmp_path):
    import torch
    from test_boundary_gpu import _write_flir
    two = torch.cuda.device_count() >= 2
    root = tmp_path / "val"
    _write_flir(root, 7, 256, 320)
    if not two:     # refusal path: no override, one device, two ranks requested
        env = _env(True)
        p = _cli("demo_mAP_FLIR", ["--dataset_path", str(root), "--fusion_method", "thermal_only", "--world-size", "2"], env, expect_ok=False)
        assert p.returncode != 0 and "only 1 GPU" in (p.stdout + p.stderr)
    env = _env(two)
    # demo_mAP_FLIR: detector -> sharded loader -> row all-gather -> rank-0 COCOeval
    res = {}
    for world in (1, 2):
        out = tmp_path / f"map{world}"
        _cli("demo_mAP_FLIR", ["--dataset_path", str(root), "--fusion_method", "thermal_only", "--outfolder", str(out),
                               "--dataset_name", f"flir_w{world}", "--world-size", str(world)], env)
        res[world] = (json.load(open(out / "FLIR_mAP_results.json")), json.load(open(out / "coco_instances_results.json")))
    assert res[2][0]["world_size"] == 2
    assert res[1][1] == res[2][1] and len(res[1][1]) > 20
    assert res[1][0]["results"] == res[2][0]["results"]
    # save_predictions (two methods) -> demo_probEn on the device route, sharded over the ranks
    pred = {}
    for world in (1, 2):
        pdir = tmp_path / f"pred{world}"
        for method in ("thermal_only", "early_fusion"):
            _cli("save_predictions", ["--dataset_path", str(root), "--fusion_method", method, "--prediction_path", str(pdir), "--batch", "4",
                                      "--world-size", str(world)], env)
        pred[world] = {m: json.load(open(pdir / f"val_{m}_predictions.json")) for m in ("thermal_only", "early_fusion")}
    assert pred[1] == pred[2]
    fused = {}
    for world in (1, 2):
        out = tmp_path / f"fuse{world}"
        _cli("demo_probEn", ["--dataset_path", str(root), "--prediction_path", str(tmp_path / "pred1"), "--detectors", "thermal_only,early_fusion",
                             "--outfolder", str(out), "--dataset_name", f"flir_p{world}", "--world-size", str(world)], env)
        fused[world] = (json.load(open(out / "FLIR_probEn_eval.json")), json.load(open(out / "coco_instances_results.json")))
    assert fused[1] == fused[2] and len(fused[1][1]) > 10


def test_world_size_2_kaist_rows_are_byte_identical(tmp_path):
    import torch
    from PIL import Image
    from proben_amd.synthetic import synthetic_images
    two = torch.cuda.device_count() >= 2
    root = tmp_path / "KAIST"
    th, rgb = synthetic_images(6, 256, 320, seed=61), synthetic_images(6, 256, 320, seed=62)
    lines = []
    for i in range(6):
        d = root / "test" / "set06" / f"V00{i % 2}"
        (d / "lwir").mkdir(parents=True, exist_ok=True)
        (d / "visible").mkdir(parents=True, exist_ok=True)
        Image.fromarray(th[i]).save(d / "lwir" / f"I{i:05d}.jpg", quality=95)
        Image.fromarray(rgb[i]).save(d / "visible" / f"I{i:05d}.jpg", quality=95)
        lines.append(f"set06/V00{i % 2}/I{i:05d}")
    split = tmp_path / "split.txt"
    split.write_text("\n".join(lines) + "\n")
    env = _env(two)
    for method in ("thermal_only", "probEn"):      # single detector; configs[4]'s two-detector binary ProbEn
        blobs = {}
        for world in (1, 2):
            out = tmp_path / f"{method}{world}"
            _cli("demo_LAMR_KAIST", ["--dataset_path", str(root), "--split_file", str(split), "--fusion_method", method,
                                     "--out_folder", str(out), "--batch", "2", "--world-size", str(world)], env)
            blobs[world] = open(out / f"KAIST_{method}_result.txt", "rb").read()
        assert blobs[1] == blobs[2] and blobs[1].count(b"\n") > 5, method


def test_one_rank_rccl_group_runs_the_nccl_code_path(tmp_path):
    """RCCL refuses two ranks on one device, so on a one-GPU box the two-rank tests above use gloo.  This test puts RCCL itself
    under the drivers: ONE rank under the launcher with PROBEN_FORCE_DIST=1 -> init_process_group("nccl") builds a communicator,
    the evaluator's row gather and bench.py's per-step all_gather_into_tensor run as RCCL collectives on device tensors
    (NCCL_DEBUG=VERSION makes the library announce itself), and the results equal the plain single-process run."""
    import proben_amd  # noqa: F401
    from proben_amd import launch
    from test_boundary_gpu import _write_flir
    root = tmp_path / "val"
    _write_flir(root, 5, 256, 320)
    env = _env(True)
    plain = tmp_path / "plain"
    _cli("demo_mAP_FLIR", ["--dataset_path", str(root), "--fusion_method", "thermal_only", "--outfolder", str(plain), "--dataset_name", "flir_plain"], env)
    env["PROBEN_FORCE_DIST"] = "1"
    env["NCCL_DEBUG"] = "VERSION"
    forced = tmp_path / "forced"
    argv = ["--dataset_path", str(root), "--fusion_method", "thermal_only", "--outfolder", str(forced), "--dataset_name", "flir_forced"]
    p = subprocess.run(launch.launch_command(argv, 1, launch.free_port(), module="proben_amd.cli.demo_mAP_FLIR"), capture_output=True, text=True,
                       env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "RCCL version" in (p.stdout + p.stderr), "RCCL did not initialise"      # "RCCL version : 2.26.6-..." (NCCL_DEBUG=VERSION)
    a, b = json.load(open(plain / "FLIR_mAP_results.json")), json.load(open(forced / "FLIR_mAP_results.json"))
    assert a["results"] == b["results"]
    assert json.load(open(plain / "coco_instances_results.json")) == json.load(open(forced / "coco_instances_results.json"))
    # bench.py: the per-step all-gather of the fused rows over the one-rank RCCL group
    args = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4", "--depth", "50", "--no-cpu-baseline", "--no-roofline", "--no-micro"]
    p = subprocess.run(launch.launch_command(args, 1, launch.free_port(), script=os.path.join(ROOT, "bench.py")), capture_output=True, text=True,
                       env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and "all_gather_into_tensor" in line["config"]["collective"] and line["value"] > 0
    assert line["config"]["collective_calls_per_step"] == 1, line["config"]      # ONE RCCL collective per step, counted inside the timed loop
    assert len(line["value_windows"]) == 3 and all(v > 0 for v in line["value_windows"])
    # the box-head trainer: parameter broadcast + the bucketed gradient all-reduce over the one-rank RCCL group == the plain run
    logs = []
    for forced_run in (False, True):
        e = dict(env)
        if not forced_run:
            e.pop("PROBEN_FORCE_DIST")
        argv = ["--steps", "6", "--images-per-step", "2", "--seed", "4"]
        cmd = launch.launch_command(argv, 1, launch.free_port(), module="proben_amd.cli.train_box_head") if forced_run else \
            [sys.executable, "-m", "proben_amd.cli.train_box_head"] + argv
        p = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
        logs.append([json.loads(l) for l in p.stdout.splitlines() if l.startswith('{"step"')])
    assert logs[0] and logs[0] == logs[1], logs


def test_world_size_2_box_head_training_keeps_the_ranks_in_step():
    """cli/train_box_head.py --world-size 2 (two GPUs over RCCL; on a one-GPU box the two ranks share the device and the reducer's
    collectives go over gloo): the ranks see DIFFERENT frames (their losses differ) and end with the SAME head, bit for bit - rank 0's
    initial parameters were broadcast, every step applied the summed gradients x 1 / world on both - and that head differs from the
    one a single rank trains on its own frames."""
    import torch
    env = _env(torch.cuda.device_count() >= 2)
    runs = {}
    for world in (2, 1):
        p = _cli("train_box_head", ["--steps", "4", "--images-per-step", "2", "--seed", "6", "--world-size", str(world)], env)
        rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{"rank"')]
        assert len(rows) == world, (p.stdout[-1500:], p.stderr[-1500:])
        runs[world] = {r["rank"]: r for r in rows}
    two = runs[2]
    if torch.cuda.device_count() >= 2:      # two devices: the ranks MUST be on RCCL and the reducer on its async device all-reduce branch
        assert all(r["backend"] == "nccl" and r["reducer_on_device"] for r in two.values()), two
    else:                                   # one device shared by two ranks: RCCL refuses that, the collectives are staged over gloo
        assert all(r["backend"] == "gloo" and not r["reducer_on_device"] for r in two.values()), two
    assert two[0]["weights_sha"] == two[1]["weights_sha"]
    assert two[0]["last_loss_cls"] != two[1]["last_loss_cls"]
    assert runs[1][0]["weights_sha"] != two[0]["weights_sha"]
