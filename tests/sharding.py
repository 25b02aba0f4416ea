"""Host logic shared by bench.py-style drivers: contiguous block ranges per rank and the
one-off setup-blob broadcast (RCCL on GPUs, gloo in the CPU tests).  Blocks are independent, so
there is no collective on the data path."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(total_blocks, rank, world):
    """Contiguous [lo, hi) of `total_blocks` for `rank`; sizes differ by at most one."""
    base, rem = divmod(total_blocks, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob_or_none, device):
    """Rank 0 passes the blob (numpy uint8); every rank returns an identical numpy copy."""
    rank = dist.get_rank()
    if rank == 0:
        t = torch.from_numpy(np.ascontiguousarray(blob_or_none).copy()).to(device)
        size = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    else:
        size = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(size, 0)
    if rank != 0:
        t = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(t, 0)
    return t.cpu().numpy()
