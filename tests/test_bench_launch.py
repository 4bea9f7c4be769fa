"""bench.py's multi-rank entry (`--gpus N` must launch N ranks itself and never report an N-GPU line from
fewer devices).  Replaces what detectron2/engine/launch.py:24-84 does for the reference."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_is_one_process_per_gpu_over_loopback():
    import bench
    cmd = bench.launch_command(["--gpus", "4", "--steps", "7"], 4, 12345)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")


def test_no_self_launch_when_already_under_a_launcher(monkeypatch):
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    called = []
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: called.append(a) or 0)
    bench.maybe_self_launch(bench.parse(["--gpus", "2"]), ["--gpus", "2"])   # returns: the ranks already exist
    bench.maybe_self_launch(bench.parse(["--gpus", "1"]), [])
    assert not called


def test_refuses_more_gpus_than_visible(monkeypatch):
    """CPU container: 0 devices visible -> `--gpus 2` exits with a message instead of measuring one device."""
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.maybe_self_launch(bench.parse(["--gpus", "64"]), ["--gpus", "64"])
    assert "only" in str(e.value) and "64" in str(e.value)


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` spawns two ranks, RCCL all-gathers the fused rows every step, and the line says
    n_gpus 2 (skipped on a 1-GPU box; there the refusal path is checked instead)."""
    import torch
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    args = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4", "--depth", "50", "--no-cpu-baseline", "--no-roofline", "--no-micro"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=900)
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and ("only %d GPU" % torch.cuda.device_count()) in (p.stderr + p.stdout)
        pytest.skip("one GPU visible: multi-rank launch not exercised (refusal path checked)")
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "all_gather" in line["config"]["collective"]
