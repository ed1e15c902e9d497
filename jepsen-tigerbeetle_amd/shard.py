"""Multi-GPU layout of the batch path: histories (and jepsen.independent keys) are
independent units, so rank r of N simply owns every N-th history -- no data-path
collective.  torch.distributed (RCCL on the GPU box, gloo in the CPU tests) is used
only to agree on the verdict summary and on the max-over-ranks clock."""
from __future__ import annotations

import numpy as np


def shard_indices(n_items: int, rank: int, world: int) -> np.ndarray:
    """Round-robin: item i belongs to rank i % world (keeps per-rank work balanced
    when neighbouring histories have similar cost)."""
    return np.arange(rank, n_items, world, dtype=np.int64)


def merge_verdicts(local_verdicts: np.ndarray, n_items: int, rank: int, world: int, dist=None) -> np.ndarray:
    """All ranks end up with the full verdict vector (1 valid / 0 invalid / -1 unknown).
    One small all_reduce(SUM) over an int32 vector offset by +2 so that 'not mine' = 0."""
    full = np.zeros(n_items, np.int32)
    full[shard_indices(n_items, rank, world)] = np.asarray(local_verdicts, np.int32) + 2
    if world > 1:
        import torch
        t = torch.from_numpy(full)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        full = t.cpu().numpy()
    return full - 2


def max_over_ranks(seconds: float, world: int, dist=None) -> float:
    if world == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
