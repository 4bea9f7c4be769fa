"""the dataset drivers honour --world-size.  Two gloo ranks (CPU, stub predictor - the detectors have no
CPU path) against one rank: the gathered AP table, the merged prediction JSON and the KAIST text must be IDENTICAL, because
shards are contiguous (rank order == dataset order) and rows cross ranks as exact float tables.
Reference being replaced: engine/launch.py:24-84, evaluation/evaluator.py:84-168, evaluation/FLIR_evaluation.py:124-131,
data/samplers/distributed_sampler.py:172-199."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "dist_driver_stub.py")


def _write_flir(root, n, H=64, W=80):
    from PIL import Image
    rng = np.random.default_rng(3)
    (root / "thermal_8_bit").mkdir(parents=True)
    (root / "RGB").mkdir()
    images, anns, aid = [], [], 1
    for i in range(n):
        stem = f"FLIR_{i:05d}"
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(root / "thermal_8_bit" / (stem + ".jpeg"), quality=95)
        Image.fromarray(rng.integers(0, 255, (H, W, 3), dtype=np.uint8)).save(root / "RGB" / (stem + ".jpg"), quality=95)
        images.append({"id": 100 + i, "file_name": f"thermal_8_bit/{stem}.jpeg", "height": H, "width": W})
        for _ in range(int(rng.integers(1, 5))):
            w, h = rng.uniform(8, 39), rng.uniform(8, 39)
            x, y = rng.uniform(0, W - w), rng.uniform(0, H - h)
            anns.append({"id": aid, "image_id": 100 + i, "category_id": int(rng.integers(1, 4)), "bbox": [x, y, w, h], "area": w * h, "iscrowd": 0})
            aid += 1
    cats = [{"id": 1, "name": "person"}, {"id": 2, "name": "bicycle"}, {"id": 3, "name": "car"}]
    json.dump({"images": images, "annotations": anns, "categories": cats}, open(root / "FLIR_thermal_RGBT_pairs_val.json", "w"))


def _run(driver, args, world):
    """world 1: the stub entry directly; world 2: through the launcher command the drivers build for themselves."""
    sys.path.insert(0, ROOT)
    import proben_amd  # noqa: F401
    from proben_amd import launch
    env = launch.launch_env()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    argv = [driver] + args + ["--device", "cpu", "--world-size", str(world)]
    cmd = [sys.executable, STUB] + argv if world == 1 else launch.launch_command(argv, world, launch.free_port(), script=STUB)
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return p


def test_demo_map_flir_two_ranks_equal_one_rank(tmp_path):
    root = tmp_path / "val"
    _write_flir(root, 11)          # 11 images over 2 ranks: blocks of 6 and 5 (ceil rule of InferenceSampler)
    res = {}
    for world in (1, 2):
        out = tmp_path / f"out{world}"
        _run("demo_mAP_FLIR", ["--dataset_path", str(root), "--fusion_method", "thermal_only", "--outfolder", str(out),
                               "--dataset_name", f"flir_dist_{world}"], world)
        res[world] = json.load(open(out / "FLIR_mAP_results.json"))
        res[world]["coco"] = json.load(open(out / "coco_instances_results.json"))
    assert res[2]["world_size"] == 2 and res[1]["world_size"] == 1
    assert res[1]["coco"] == res[2]["coco"] and len(res[1]["coco"]) > 10        # same rows in the same order, exact floats
    assert res[1]["results"] == res[2]["results"]                                # AP table bit for bit
    assert res[1]["results"]["bbox"]["AP50"] >= 0


def test_save_predictions_two_ranks_merge_in_dataset_order(tmp_path):
    root = tmp_path / "val"
    _write_flir(root, 7)
    outs = {}
    for world in (1, 2):
        out = tmp_path / f"pred{world}"
        _run("save_predictions", ["--dataset_path", str(root), "--fusion_method", "thermal_only", "--prediction_path", str(out), "--batch", "3"], world)
        outs[world] = json.load(open(out / "val_thermal_only_predictions.json"))
    assert outs[1] == outs[2] and len(outs[1]["image"]) == 7
    assert outs[1]["image_id"] == [100 + i for i in range(7)]


def test_kaist_two_ranks_write_the_same_bytes(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(8)
    root = tmp_path / "KAIST"
    lines = []
    for i in range(9):
        d = root / "test" / "set06" / f"V00{i % 2}"
        for sub in ("lwir", "visible"):
            (d / sub).mkdir(parents=True, exist_ok=True)
            Image.fromarray(rng.integers(0, 255, (64, 80, 3), dtype=np.uint8)).save(d / sub / f"I{i:05d}.jpg", quality=95)
        lines.append(f"set06/V00{i % 2}/I{i:05d}")
    split = tmp_path / "split.txt"
    split.write_text("\n".join(lines) + "\n")
    blobs = {}
    for world in (1, 2):
        out = tmp_path / f"k{world}"
        _run("demo_LAMR_KAIST", ["--dataset_path", str(root), "--split_file", str(split), "--fusion_method", "thermal_only",
                                 "--out_folder", str(out), "--batch", "4"], world)
        txt = open(out / "KAIST_thermal_only_result.txt", "rb").read()
        var = np.load(out / "KAIST_thermal_only_variance.npz", allow_pickle=True)["vars"].item()
        blobs[world] = (txt, {k: np.asarray(v).tolist() for k, v in var.items()}, json.load(open(out / "KAIST_thermal_only_summary.json")))
    assert blobs[1][0] == blobs[2][0] and len(blobs[1][0]) > 100      # byte-identical text rows
    assert blobs[1][1] == blobs[2][1] and sorted(blobs[1][1]) == list(range(1, 10))
    assert blobs[2][2]["world_size"] == 2 and blobs[1][2]["rows"] == blobs[2][2]["rows"]


def test_world_size_refuses_fewer_gpus_and_joins_under_a_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import proben_amd  # noqa: F401
    from proben_amd import launch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("PROBEN_DIST_BACKEND", raising=False)
    with pytest.raises(SystemExit) as e:
        launch.maybe_self_launch(64, ["--world-size", "64"], module="proben_amd.cli.demo_mAP_FLIR", device="cuda")
    assert "only" in str(e.value) and "64" in str(e.value)
    called = []
    monkeypatch.setattr(launch.subprocess, "call", lambda *a, **k: called.append(a) or 0)
    launch.maybe_self_launch(1, [], module="m")                      # nothing to launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    launch.maybe_self_launch(2, ["--world-size", "2"], module="m")   # already under a launcher: join only
    assert not called
    cmd = launch.launch_command(["--x", "1"], 8, 4242, module="proben_amd.cli.demo_probEn")
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and cmd[-4:] == ["-m", "proben_amd.cli.demo_probEn", "--x", "1"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
