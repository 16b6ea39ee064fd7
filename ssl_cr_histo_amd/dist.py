"""One process per GPU: torch.distributed (backend "nccl" == RCCL on ROCm, or gloo on CPU for the tests) is used only
to bootstrap -- rank/world discovery and broadcasting the RCCL unique id of the engine's own communicator.  Every collective after
that (gradient buckets, BatchNorm sums, the loss meters' one all-reduce per print_freq) is issued by the engine itself.

Sharding: pure data parallel; rank r takes labeled rows [r*b/G, (r+1)*b/G) and unlabeled rows
[r*mu*b/G, ...) of the global batch; every loss is scaled by the GLOBAL count so the all-reduced SUM of the per-rank
gradients is exactly the single-device gradient.
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(backend=None):
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_global, rank, world):
    """contiguous shard [lo, hi) of n_global rows for `rank`; remainders go to the lowest ranks."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(t, rank, world):
    lo, hi = shard_range(t.shape[0], rank, world)
    return t[lo:hi]


def broadcast_bytes(payload, src=0):
    """host-side broadcast of a small byte string (the 128-byte RCCL unique id) from `src`."""
    if not dist.is_initialized():
        return payload
    obj = [payload]
    dist.broadcast_object_list(obj, src=src)
    return obj[0]


def attach_engine(engine):
    """create the engine's RCCL communicator over the initialised process group."""
    rank, world, _ = env_rank_world()
    if world > 1:
        engine.init_comm(rank, world, broadcast_bytes)
    return engine
