"""The N > 1 path on CPU: world-size-2 gloo processes exercise the shard rule, the row all-gather (rank
order == single-process order) and the distributed evaluator."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, golden):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import proben_amd  # noqa: F401
    from proben_amd import comm, data, evaluation
    z = json.load(open(os.path.join(golden, "cocoeval_case.json")))
    dets = z["dets"]
    rows_all = torch.tensor([[d["image_id"], *d["bbox"], d["score"], d["category_id"]] for d in dets], dtype=torch.float64)
    mine = list(comm.shard_range(len(dets)))
    gathered = comm.all_gather_rows(rows_all[mine[0]:mine[-1] + 1] if mine else rows_all[:0])
    assert torch.equal(gathered, rows_all), "rank-ordered concatenation must equal the single-process order"
    pad = comm.all_gather_padded(torch.full((3,), float(rank)))
    assert pad.tolist() == [[0.0] * 3, [1.0] * 3]
    # distributed evaluator: every rank processes its shard, rank 0 evaluates the union
    gt_path = os.path.join(tmp, "gt.json")
    if rank == 0:
        json.dump(z["gt"], open(gt_path, "w"))
    comm.synchronize()
    data.register_coco_instances("flir_dist", {}, gt_path, tmp)
    ev = evaluation.FLIREvaluator("flir_dist", proben_amd.get_cfg(), True)
    rows = rows_all[mine[0]:mine[-1] + 1].clone() if mine else rows_all[:0]
    ev.process_rows(rows.numpy())
    res = ev.evaluate()
    if rank == 0:
        assert abs(res["bbox"]["AP50"] - z["stats"][1] * 100) < 1e-9
        open(os.path.join(tmp, "ok"), "w").write("1")
    else:
        assert res == {}
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path, golden_dir):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), golden_dir), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _bench_step_worker(rank, world, port, tmp):
    """bench.py's timed step with a stub pipeline: the per-step collective is exactly the product's all_gather_fused_rows."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import proben_amd  # noqa: F401
    from proben_amd import comm, pipeline

    class StubPipeline:            # what FramePairPipeline returns, without detectors: [B, D] padded rows + counts, rank-specific values
        streams, staggered = None, False

        def __init__(self, *a, **k):
            pass

        def __call__(self, batch, out_sizes, resize_to):
            B, D = 3, 5
            fused = {"boxes": torch.full((B, D, 4), float(rank), dtype=torch.float64), "scores": torch.full((B, D), 0.5 + rank), "classes": torch.full((B, D), float(rank)),
                     "counts": torch.tensor([1 + rank, 2 + rank, 0], dtype=torch.int32)}
            return [fused], fused
    pipeline.FramePairPipeline = StubPipeline
    seen = []
    orig = comm.all_gather_fused_rows
    comm.all_gather_fused_rows = lambda payload: seen.append(orig(payload)) or seen[-1]
    frames = [torch.zeros((3, 8, 8, 3), dtype=torch.uint8)]
    step = bench.make_step([None], frames, {"fuse": ("probEn", "v-avg"), "detectors": [3]}, world)
    calls = bench.count_collectives()       # counts at torch.distributed's entry points (what bench.py reports as collective_calls_per_step)
    for _ in range(2):
        step()
    assert len(seen) == 2
    assert calls["n"] == 2, f"ONE collective per step is the documented contract (DESIGN 5), counted {calls['n']} in 2 steps"
    g = seen[-1]
    assert g["boxes"].shape == (world, 3, 5, 4) and g["boxes"].dtype == torch.float64 and g["counts"].tolist() == [[1, 2, 0], [2, 3, 0]]
    assert g["boxes"][1].unique().tolist() == [1.0] and g["classes"].dtype == torch.float32 and g["counts"].dtype == torch.int32
    assert g["scores"][0].unique().tolist() == [0.5] and g["scores"][1].unique().tolist() == [1.5]     # rank order == GPU order
    if rank == 0:
        open(os.path.join(tmp, "ok_step"), "w").write("1")
    dist.destroy_process_group()


def test_bench_step_collective_world_2_gloo(tmp_path):
    """The per-step all-gather of bench.py (`--gpus N`, one rank per GPU over RCCL on the device) with two gloo ranks and a stub
    pipeline: every rank contributes its padded fused rows, every rank receives [world, ...] in rank order."""
    port = _free_port()
    mp.spawn(_bench_step_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_step").exists()


def test_external_launcher_with_more_ranks_than_gpus_is_refused():
    """engine/launch.py:24-84 asserts `num_gpus_per_machine <= torch.cuda.device_count()`; under an EXTERNAL launcher (RANK / LOCAL_RANK
    in the environment) init_distributed must refuse the same way instead of dying in set_device or inside RCCL.  This container
    has no GPU: local rank 1 of 2 has no device."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="1", WORLD_SIZE="2", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("PROBEN_DIST_BACKEND", None)
    code = "import sys; sys.path.insert(0, %r); import proben_amd; from proben_amd import launch; launch.init_distributed('cuda')" % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: local rank 1 is legitimate here")
    assert p.returncode != 0 and "refusing to run 2 ranks on fewer devices" in p.stderr
