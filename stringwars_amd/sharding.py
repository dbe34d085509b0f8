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


def shard_ranges_by_block_cells(block_cells: Sequence[int], block: int, total: int, world: int) -> List[Tuple[int, int]]:
    """The same balance when no rank holds all the lengths: `block_cells[k]` is the cell count of pairs
    [k*block, (k+1)*block) (each rank sums the blocks of the slice it generated; one small all-gather puts the
    list together). Cuts fall inside a block by linear interpolation -- exact to within one block's
    non-uniformity, which for the i.i.d. synthetic stream is < 0.1 %."""
    cells = np.asarray(block_cells, dtype=np.float64)
    prefix = np.concatenate([[0.0], np.cumsum(cells)])
    cuts = [0]
    for r in range(1, world):
        target = prefix[-1] * r / world
        k = int(np.searchsorted(prefix, target, side="right")) - 1
        k = min(max(k, 0), len(cells) - 1)
        inside = (target - prefix[k]) / cells[k] if cells[k] > 0 else 0.0
        cuts.append(min(total, int(round((k + inside) * block))))
    cuts.append(total)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def weak_shard_first(rank: int, pairs_per_rank: int) -> int:
    """First pair index of rank `rank`'s shard of the seeded synthetic stream (weak scaling, bench.py)."""
    return rank * pairs_per_rank


def chunk_ranges(count: int, chunks: int) -> List[Tuple[int, int]]:
    """Splits a shard of `count` results into up to `chunks` contiguous pieces (the gather of piece j overlaps the
    scoring of piece j + 1)."""
    chunks = max(1, min(chunks, count)) if count else 1
    return [(count * j // chunks, count * (j + 1) // chunks) for j in range(chunks)]


class ChunkedGather:
    """Variable-size gather of per-rank result slices into ONE vector on `dst`, without padding, in pieces: every
    rank cuts its shard with `chunk_ranges`, so the root knows from the shard sizes alone where piece j of rank r
    lands and posts a receive straight into `full[...]` for it; the others send their piece (`batch_isend_irecv` =
    one ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd per call on RCCL). Piece j travels while piece j + 1 is
    still being scored.

    `full` may be handed in (a buffer per pipeline slot, reused from step to step: allocating and zeroing 4 B per pair
    on the root every step is a memset the size of the whole result). `transport` narrower than the result type
    (`torch.uint8` when no distance can exceed 255: both strings of every pair are that short) sends a quarter of the
    bytes over xGMI -- the root's inbound links are what bounds a gather of word-sized pairs -- and widens the pieces
    into `full` on arrival; the vector the root ends up with is the same u32 vector either way."""

    def __init__(self, ranges: Sequence[Tuple[int, int]], chunks: int, dtype, device, dst: int = 0, full=None, transport=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.ranges, self.dst, self.chunks = list(ranges), dst, chunks
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.pieces = [chunk_ranges(hi - lo, chunks) for lo, hi in self.ranges]
        self.staged = dist.get_backend() == "gloo" and torch.device(device).type != "cpu"   # gloo moves host memory
        self.transport = transport if transport is not None and transport != dtype else None
        where = "cpu" if self.staged else device
        if self.rank == dst:
            self.full = full if full is not None else torch.zeros(self.ranges[-1][1], dtype=dtype, device=where)
            # narrow transport: the pieces arrive in a staging vector of the transport type, laid out like `full`
            self.landing = torch.empty(self.ranges[-1][1], dtype=self.transport, device=where) if self.transport is not None else self.full
        else:
            self.full = self.landing = None
        self.pending = []
        self.arrived = []          # (lo, hi) ranges of `landing` to widen into `full` once their receive has completed

    def piece(self, rank: int, j: int) -> Tuple[int, int]:
        """Piece j of rank `rank`, as indices into that rank's own shard."""
        pieces = self.pieces[rank]
        return pieces[j] if j < len(pieces) else (0, 0)

    def send_chunk(self, local, j: int) -> None:
        """Rank-local results of piece j (`local` holds the whole shard) go to their place on the root."""
        dist = self.dist
        lo, hi = self.piece(self.rank, j)
        ops = []
        if self.rank == self.dst:
            mine = self.ranges[self.rank][0]
            if hi > lo:
                self.full[mine + lo:mine + hi].copy_(local[lo:hi])
            for r in range(self.world):
                r_lo, r_hi = self.piece(r, j)
                if r != self.dst and r_hi > r_lo:
                    base = self.ranges[r][0]
                    ops.append(dist.P2POp(dist.irecv, self.landing[base + r_lo:base + r_hi], r))
                    if self.transport is not None:
                        self.arrived.append((base + r_lo, base + r_hi))
        elif hi > lo:
            piece = local[lo:hi] if self.transport is None else local[lo:hi].to(self.transport)
            ops.append(dist.P2POp(dist.isend, piece.cpu() if self.staged else piece, self.dst))
        if ops:
            self.pending.extend(dist.batch_isend_irecv(ops))

    def wait(self):
        for work in self.pending:
            work.wait()
        self.pending = []
        for lo, hi in self.arrived:
            self.full[lo:hi].copy_(self.landing[lo:hi])      # widening copy, ordered behind the receive it follows
        self.arrived = []
        return self.full


def gather_distances(local, counts: Optional[Sequence[int]] = None, dst: int = 0):
    """Gathers every rank's result tensor to `dst` in rank order and returns the concatenation there (None
    elsewhere). Equal shard sizes use one `dist.gather`; ragged shards go through `ChunkedGather` (no padding)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if counts is None:
        counts = [int(local.numel())] * world
    if len(set(counts)) == 1:
        buffers = [torch.zeros(counts[0], dtype=local.dtype, device=local.device) for _ in range(world)] if rank == dst else None
        dist.gather(local, buffers, dst=dst)
        return torch.cat(buffers) if rank == dst else None
    starts = np.concatenate([[0], np.cumsum(counts)])
    gather = ChunkedGather([(int(starts[r]), int(starts[r + 1])) for r in range(world)], 1, local.dtype, local.device, dst)
    gather.send_chunk(local, 0)
    full = gather.wait()
    if rank != dst:
        return None
    return full.to(local.device) if full.device != local.device else full
