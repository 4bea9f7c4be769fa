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
