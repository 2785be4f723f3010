"""Data-parallel plumbing for the Lloyd iteration over the GPUs of one node (one process per
GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The reference has no distributed code.  Points shard naturally: rank r owns the contiguous
block [r*n/W, (r+1)*n/W) of columns; centres are replicated.  Per iteration there is exactly ONE
exchange -- a SUM all-reduce of the reduce buffer [sums p*K | counts p*K | nk K | obj2 1]
(include/spkm.h spkm_reduce_len) -- after which every rank finalises the same centres.  The only
other exchange is the rare EmptyAction='singleton' pick (kmeans_sparsified.m:436-437), a MAXLOC
over (max distance, global index) plus a one-column broadcast from the owner.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """[first, last) of the contiguous block of points owned by ``rank``."""
    return rank * n_total // world, (rank + 1) * n_total // world


def reduce_layout(p: int, K: int) -> dict:
    """Slices of the reduce buffer (must match spkm_reduce_len / spkm_accumulate_dev)."""
    pk = p * K
    return dict(sums=slice(0, pk), counts=slice(pk, 2 * pk), nk=slice(2 * pk, 2 * pk + K),
                obj2=slice(2 * pk + K, 2 * pk + K + 1), length=2 * pk + K + 1)


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def allreduce_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of the per-shard reduce buffer (no-op when not distributed)."""
    if is_distributed():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def global_first_argmax(local_max: float, local_index: int, first: int, group=None) -> tuple[int, int, float]:
    """MATLAB's [~,iMax] = max(distances) over the whole (sharded) vector: the largest value,
    lowest GLOBAL index on ties.  Every rank passes its shard's (max, first local index of it)
    and its block offset; returns (owner rank, global index, value) identically on all ranks."""
    if not is_distributed():
        return 0, first + local_index, local_max
    world = dist.get_world_size(group)
    mine = torch.tensor([local_max, float(first + local_index)], dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        mine = mine.cuda()
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    vals = torch.stack(allv).cpu()
    best = 0
    for r in range(1, world):
        if vals[r, 0] > vals[best, 0] or (vals[r, 0] == vals[best, 0] and vals[r, 1] < vals[best, 1]):
            best = r
    return best, int(vals[best, 1].item()), float(vals[best, 0].item())


def broadcast_column(col: torch.Tensor, owner: int, group=None) -> torch.Tensor:
    """Owner sends the (densified) column picked by global_first_argmax to everyone."""
    if is_distributed():
        dist.broadcast(col, src=owner, group=group)
    return col
