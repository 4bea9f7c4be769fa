"""One process per GPU for the dataset drivers (`--world-size N`).

Replaces detectron2/engine/launch.py:24-84 (mp.spawn of `main_func` + init_process_group("NCCL") + a gloo side group
created lazily by utils/comm.py:36-48) for the inference path:

  * a driver started with `--world-size N` and NOT already under a launcher re-executes itself under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P -m <module> ...`
    (the same launcher the round driver uses for bench.py); it REFUSES when fewer than N devices are visible, so an N-rank
    result never comes from fewer GPUs;
  * every rank then joins through `init_distributed()`: `torch.cuda.set_device(LOCAL_RANK)`, backend "nccl" (= RCCL over
    xGMI) bound to that device, plus a gloo side group for the few pickled host objects (prediction-JSON shards) - the
    evaluation rows themselves travel as tensors through comm.all_gather_rows on the device;
  * under an external launcher (RANK / WORLD_SIZE in the environment) the driver only joins.
`--device cpu` selects gloo for everything: the rank plumbing is testable without a GPU (tests/test_dist_drivers_cpu.py),
the detectors themselves have no CPU path.  PROBEN_DIST_BACKEND=gloo does the same for CUDA ranks and lets them SHARE
devices (rank r uses GPU r % device_count; collectives run on host tensors): how the sharded drivers are exercised end to
end - real detectors, two ranks - on a one-GPU box, where RCCL would refuse two ranks on one device.
"""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, nproc, port=None, script=None, module=None):
    """The `torch.distributed.run` command line for `nproc` local ranks of a script path or a `-m` module.  port None: the launcher
    binds its own rendezvous port on 127.0.0.1 (`--standalone`; no bind-release-rebind race with other jobs on a busy host)."""
    assert (script is None) != (module is None), "give a script path or a module name"
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}"]
    head += ["--standalone", "--local-addr", "127.0.0.1"] if port is None else ["--master-addr", "127.0.0.1", "--master-port", str(port)]
    return head + (["-m", module] if module else [script]) + list(argv)


def under_launcher():
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / cross-process device memory on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env["PYTHONPATH"] = ROOT + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")   # the `proben_amd` import shim
    return env


def maybe_self_launch(world_size, argv, module=None, script=None, device="cuda", what=None):
    """`--world-size N > 1` outside a launcher: spawn the N ranks and exit with their status.  Returns when there is
    nothing to launch (N <= 1, or the ranks already exist)."""
    if world_size <= 1 or under_launcher():
        return
    what = what or module or script
    if str(device).startswith("cuda") and os.environ.get("PROBEN_DIST_BACKEND", "nccl") != "gloo":
        import torch
        have = torch.cuda.device_count()
        if have < world_size:
            sys.exit(f"{what}: --world-size {world_size} requested but only {have} GPU(s) are visible; refusing to run "
                     f"{world_size} ranks on fewer devices")
    sys.exit(subprocess.call(launch_command(argv, world_size, None, script=script, module=module), env=launch_env()))


def init_distributed(device="cuda", expect_world=None):
    """Join the process group the environment describes (no-op for a single process).  Returns (rank, world, torch.device)."""
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cuda = str(device).startswith("cuda")
    share = cuda and os.environ.get("PROBEN_DIST_BACKEND", "nccl") == "gloo"
    if share:
        local %= max(torch.cuda.device_count(), 1)
    if cuda and not share:
        # an external launcher (torchrun with more ranks than GPUs) must get the same refusal as the self-launch path, not an
        # "invalid device ordinal" from set_device or an RCCL error from two ranks sharing a device
        have = torch.cuda.device_count()
        if local >= have:
            sys.exit(f"rank {rank}: LOCAL_RANK {local} but only {have} GPU(s) are visible; refusing to run {world} ranks on fewer devices "
                     "(one process per GPU; PROBEN_DIST_BACKEND=gloo lets ranks share a device for testing)")
    dev = torch.device("cuda", local) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(local)
    if expect_world is not None and expect_world > 1 and world != expect_world:
        sys.exit(f"--world-size {expect_world} but the launcher started {world} rank(s)")
    # PROBEN_FORCE_DIST=1: build the group even for ONE rank under a launcher - RCCL initialises a communicator and runs the
    # drivers' collectives for real (comm.is_distributed()); how the nccl code path is executed on a one-GPU box
    if world > 1 or (os.environ.get("PROBEN_FORCE_DIST") == "1" and under_launcher()):
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if cuda and not share:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        from . import comm
        comm.set_device(torch.device("cpu") if share else dev)
    return rank, world, dev


def shutdown(barrier=True):
    """Leave the process group.  `barrier=False` for a rank that exits on an error path (or has nothing left to do while rank 0
    still writes results): the others must not sit in a collective that this rank never joins."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if barrier:
            try:
                dist.barrier()
            except Exception:       # a peer has gone: nothing left to synchronise with
                pass
        dist.destroy_process_group()
    from . import comm
    comm._object_group.cache_clear()
    comm._DEVICE = None
