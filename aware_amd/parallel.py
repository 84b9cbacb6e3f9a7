"""Data parallelism over clips: one process per GPU, no collective on the data path.

Every clip is an independent optimisation problem against a detector that each rank regenerates
bit-identically from the seed (reference: service/embed.py:52-53 runs even the two stereo channels
as separate problems; multibit_detector_net.py:78 fixes the seed), so ranks only exchange the
final counters: bit errors, bits, waveform-seconds (SUM) and wall time (MAX).  torch.distributed's
"nccl" backend is RCCL on ROCm; "gloo" is used on CPU-only hosts (tests)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# AWARE_FORCE_COLLECTIVES=1: create the process group and run every collective even with ONE rank -- the way to execute the
# RCCL code path (communicator set-up, device all-reduce of float64 / float32 buffers, barrier) on a single-GPU box
_FORCE = os.environ.get("AWARE_FORCE_COLLECTIVES", "0") not in ("", "0")


def _active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def init_distributed(cpu_only: bool = False):
    """(rank, world_size, local_rank) from the torchrun environment; initialises the process
    group when WORLD_SIZE > 1 (RCCL on GPUs; gloo when there is no GPU or `cpu_only`, which never touches HIP)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = (not cpu_only) and torch.cuda.is_available()
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl" if gpu else "gloo", rank=rank, world_size=world)
    elif gpu:
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def shard_by_cost(costs, world: int):
    """Longest-processing-time assignment of clips to ranks: cost is proportional to the frame
    count T.  Returns a list of index lists, one per rank; deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    for s in shards:
        s.sort()
    return shards


def reduce_metrics(sums: dict, maxes: dict, device=None):
    """All-reduce a handful of scalars: `sums` with SUM, `maxes` with MAX.  No-op for one rank."""
    if not _active():
        return dict(sums), dict(maxes)
    device = device or (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
    ks, km = sorted(sums), sorted(maxes)
    ts = torch.tensor([float(sums[k]) for k in ks], dtype=torch.float64, device=device)
    tm = torch.tensor([float(maxes[k]) for k in km], dtype=torch.float64, device=device)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    return {k: float(v) for k, v in zip(ks, ts.tolist())}, {k: float(v) for k, v in zip(km, tm.tolist())}


def barrier():
    if _active():
        dist.barrier()


def all_reduce_gradients(grads, average: bool = True):
    """EXTENSION (detector training; BASELINE north_star "RCCL all-reduce over xGMI on the ... detector gradients"):
    sum (or average) a list of gradient tensors over the ranks with ONE collective on a flat bucket -- 1 681 960 floats =
    6.7 MB for the model card's detector, far below the point where bucketing would pay; on 8 MI355X a ring all-reduce of
    that size is latency / per-link bound (7 links x ~153 GB/s), one launch is the right granularity.  In place; no-op for
    one rank.  RCCL on GPUs ("nccl" backend), gloo on CPU tensors."""
    if not _active():
        return grads
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n
    return grads
