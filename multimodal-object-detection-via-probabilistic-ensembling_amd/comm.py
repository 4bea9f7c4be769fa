"""Rank helpers and the ONE collective of the inference path.

Mirrors detectron2/utils/comm.py (get_world_size :20-25, get_rank :28-33, is_main_process :51-52,
synchronize :55-67, all_gather :139-174, gather :177-217).  The reference pickles Python lists and moves
them as padded uint8 tensors over a gloo (CPU/TCP) group; here the evaluation rows stay tensors:
    counts  int32 [W]            one all_gather
    rows    float32 [max_n, 7]   one all_gather   (image_id, x, y, w, h, score, category_id)
(dataset drivers, once per dataset); the per-step gather of a batch's fused rows (bench.py --gpus N) is ONE
all_gather_into_tensor of a packed byte buffer (all_gather_fused_rows)
over RCCL/xGMI for CUDA tensors (backend "nccl") or gloo for CPU tensors (tests), concatenated in RANK
ORDER so the row order equals the single-GPU order (InferenceSampler shards are contiguous).
The payload is tiny (<= 3 MB per 1000 images): one latency-bound call, no bucketing.
"""
import functools

import torch
import torch.distributed as dist

_DEVICE = None   # the rank's device for tensor collectives (launch.init_distributed sets it); None = where the tensor lives


def set_device(device):
    global _DEVICE
    _DEVICE = torch.device(device)


@functools.lru_cache()
def _object_group():
    """Pickled host objects go over gloo (utils/comm.py:36-48 `_get_global_gloo_group`): with the NCCL/RCCL backend the
    object collectives would stage every pickle through device memory."""
    if dist.get_backend() == "nccl":
        return dist.new_group(backend="gloo")
    return dist.group.WORLD


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_distributed():
    """A process group exists - also a ONE-rank one (PROBEN_FORCE_DIST=1 under a launcher: how the RCCL code path is executed
    on a one-GPU box; the collectives then run for real instead of taking the world-size-1 shortcut)."""
    return dist.is_available() and dist.is_initialized()


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def shard_range(num_items, rank=None, world=None):
    """InferenceSampler (data/samplers/distributed_sampler.py:172-199): contiguous blocks of ceil(N/W)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    shard = (num_items - 1) // world + 1 if num_items > 0 else 0
    begin = shard * rank
    return range(min(begin, num_items), min(shard * (rank + 1), num_items))


def all_gather_rows(rows, group=None):
    """rows: [n, C] tensor (any n per rank).  Returns [sum n, C] on every rank, rank order preserved."""
    world = get_world_size()
    if not is_distributed():
        return rows
    n = torch.tensor([rows.shape[0]], dtype=torch.int32, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros((m, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def all_gather_padded(t, group=None):
    """Fixed-shape tensor per rank -> [W, ...] (no host sync: used inside the timed bench step)."""
    world = get_world_size()
    if not is_distributed():
        return t.unsqueeze(0)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=group)
    return out


FUSED_KEYS = ("boxes", "scores", "classes", "counts")


def all_gather_fused_rows(fused, group=None):
    """Gather one batch's fused detections (padded layout, no host sync) with ONE collective: boxes (float64), scores, classes
    (float32) and counts (int32) travel as one byte buffer - boxes first, so every section starts on a multiple of its item size;
    the buffer is padded to 16 bytes - through a single `all_gather_into_tensor`, and come back as [W, ...] views of the received
    buffer in rank order.  (Replaces the pickled per-image lists of utils/comm.py:177-217; four separate all-gathers until round 5
    = four latency-bound RCCL launches per step.)"""
    if not is_distributed():
        return {k: fused[k].unsqueeze(0) for k in FUSED_KEYS}
    world = get_world_size()
    parts = [fused[k].contiguous() for k in FUSED_KEYS]
    assert parts[0].element_size() >= max(p.element_size() for p in parts[1:]), "the widest item type goes first"
    flat = [p.view(-1).view(torch.uint8) for p in parts]
    total = sum(f.numel() for f in flat)
    padded = (total + 15) // 16 * 16
    buf = torch.zeros((padded,), dtype=torch.uint8, device=parts[0].device) if padded != total else \
        torch.empty((padded,), dtype=torch.uint8, device=parts[0].device)
    torch.cat(flat, out=buf[:total])
    got = torch.empty((world, padded), dtype=torch.uint8, device=buf.device)
    dist.all_gather_into_tensor(got.view(-1), buf, group=group)
    out, off = {}, 0
    for k, p, f in zip(FUSED_KEYS, parts, flat):
        out[k] = got[:, off:off + f.numel()].view(p.dtype).view((world,) + tuple(p.shape))
        off += f.numel()
    return out


def gather(data, dst=0, group=None):
    """API-compatible object gather (utils/comm.py:177-217) for small Python objects (metrics dicts)."""
    world = get_world_size()
    if world == 1:
        return [data]
    out = [None] * world if get_rank() == dst else None
    dist.gather_object(data, out, dst=dst, group=group or _object_group())
    return out if get_rank() == dst else []


def all_gather(data, group=None):
    world = get_world_size()
    if world == 1:
        return [data]
    out = [None] * world
    dist.all_gather_object(out, data, group=group or _object_group())
    return out


def gather_rows(rows):
    """Evaluation rows of this rank ([n, C] tensor or array, any n) -> the rows of ALL ranks in rank order, on the host.
    The payload crosses ranks as ONE padded tensor all-gather on the rank's device (RCCL over xGMI for CUDA ranks) - the
    reference pickles per-image dict lists through gloo (evaluation/FLIR_evaluation.py:124-131)."""
    t = torch.as_tensor(rows)
    if not is_distributed():
        return t.cpu()
    dev = _DEVICE if _DEVICE is not None else t.device
    return all_gather_rows(t.to(dev).contiguous()).cpu()
