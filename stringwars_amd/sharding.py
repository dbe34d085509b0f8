"""Multi-GPU plumbing for the pairwise batch: pairs are independent, so rank r scores a
contiguous shard and the only collective is the gather of the u32 distances to rank 0 that the north-star
names. `torch.distributed` is the transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" on CPU in
tests); nothing here computes distances.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous count-balanced range [lo, hi) of rank `rank` (strong scaling over a fixed batch)."""
    return total * rank // world, total * (rank + 1) // world


def shard_ranges_by_cells(lengths_a: np.ndarray, lengths_b: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous ranges balanced on the prefix sum of len(a_i)*len(b_i) -- DP cells, not pair counts
    (SURVEY.md 8e "cells-balanced"). Every pair lands in exactly one range; ranges may be empty."""
    cells = lengths_a.astype(np.int64) * lengths_b.astype(np.int64)
    prefix = np.concatenate([[0], np.cumsum(cells)])
    total = int(prefix[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(prefix, total * r / world, side="left")))
    cuts.append(len(cells))
    cuts = [min(max(c, cuts[i - 1] if i else 0), len(cells)) for i, c in enumerate(cuts)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def weak_shard_first(rank: int, pairs_per_rank: int) -> int:
    """First pair index of rank `rank`'s shard of the seeded synthetic stream (weak scaling, bench.py)."""
    return rank * pairs_per_rank


def gather_distances(local, counts: Optional[Sequence[int]] = None, dst: int = 0):
    """Gathers every rank's result tensor to `dst` in rank order and returns the concatenation there
    (None elsewhere). Equal shard sizes use one `dist.gather`; ragged shards are padded to the longest."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if counts is None:
        counts = [int(local.numel())] * world
    longest = max(counts)
    padded = local
    if int(local.numel()) < longest:
        padded = torch.zeros(longest, dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local
    buffers = [torch.zeros(longest, dtype=local.dtype, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(padded, buffers, dst=dst)
    if rank != dst:
        return None
    return torch.cat([buffers[r][: counts[r]] for r in range(world)])
