"""vorbis_amd/sharding.py -- the multi-GPU plumbing of the batch path: one process per GPU,
contiguous block ranges per rank, ONE collective at start-up (the setup blob from rank 0), none on the data
path, and the max-over-ranks reduction of a timing.  `bench.py` runs on this module; the world-size-2 CPU test
(tests/test_abi_and_host.py) drives the very same functions over gloo.

torch.distributed backend "nccl" is RCCL on ROCm (xGMI between the GPUs of a node); "gloo" serves the CPU tests.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


# where the collectives' tensors live: the rank's device (RCCL), or host memory when the backend cannot take device
# tensors (a gloo build without GPU support: found out once, by all ranks together, when the group is formed)
_staging = {"host": False}


def _coll(t):
    return t.cpu() if _staging["host"] and t.is_cuda else t


def _run(fn, t, *args, **kw):
    """One collective on `t` (in place), through host memory when the group decided so at start-up (_probe_device_collectives).
    An error here is an error: every rank must be inside the same collective, so nothing is retried on another path."""
    h = _coll(t)
    fn(h, *args, **kw)
    if h is not t:
        t.copy_(h)
    return t


def _probe_device_collectives(device):
    """Can this backend take device tensors?  Decided ONCE, by all ranks together: each tries one all-reduce on a device
    tensor, then the outcomes are agreed with an all-reduce on a HOST tensor (which every backend takes) -- if any rank
    failed, all ranks stage through host memory from here on.  (Round 5 switched per rank, on the first failing
    collective: a rank that failed alone would have re-entered a collective its peers had already left -- ADVICE r05.)"""
    if dist.get_backend() == "nccl" or device.type != "cuda":
        return
    failed = 0
    try:
        dist.all_reduce(torch.zeros(1, device=device))
    except RuntimeError:
        failed = 1
    flag = torch.tensor([failed], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    _staging["host"] = bool(flag.item())


def collectives_on():
    return "host" if _staging["host"] else "device"


def init_from_env(backend="nccl", use_cuda=True, share_gpu=False):
    """Join the process group described by RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).
    Returns (rank, world, device).  A single process needs no group: (0, 1, device).
    share_gpu: every rank takes cuda:0 (the one-GPU rehearsal of the multi-rank path; not with nccl, which wants a
    device per rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if share_gpu and backend == "nccl" and world > 1:
        raise ValueError("--share-gpu needs --backend gloo: RCCL wants one device per rank")
    if use_cuda:
        dev = torch.device("cuda", local_rank if (world > 1 and not share_gpu) else 0)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            _probe_device_collectives(dev)
    return rank, world, dev


def active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(total_blocks, rank, world):
    """Contiguous [lo, hi) of `total_blocks` for `rank`; sizes differ by at most one."""
    base, rem = divmod(total_blocks, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob_or_none, device):
    """Rank 0 passes the setup blob (numpy uint8); every rank returns an identical numpy copy.  Two broadcasts:
    the length, then the bytes (~220 KB: latency-bound, link bandwidth irrelevant)."""
    if not active():
        return np.ascontiguousarray(blob_or_none)
    rank = dist.get_rank()
    if rank == 0:
        t = torch.from_numpy(np.ascontiguousarray(blob_or_none).copy()).to(device)
        size = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    else:
        size = torch.zeros(1, dtype=torch.int64, device=device)
    _run(dist.broadcast, size, 0)
    if rank != 0:
        t = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    _run(dist.broadcast, t, 0)
    return t.cpu().numpy()


def barrier():
    if active():
        dist.barrier()


def max_over_ranks(value, device):
    """The slowest rank's figure (every rank gets it)."""
    if not active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    _run(dist.all_reduce, t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not active():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    _run(dist.all_reduce, t, op=dist.ReduceOp.SUM)
    return int(t.item())


def finish():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
