"""N > 1 path on CPU: two gloo ranks score their shards of the seeded stream (the oracle stands in for the GPU
kernels here -- this test is about shard boundaries and the gather, not arithmetic) and rank 0's gathered
vector must equal the single-process result, for the weak-scaling split bench.py uses, a count-balanced strong
split and a cells-balanced ragged split."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, queue):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle
    import stringwars_amd as sw
    from stringwars_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    per_rank, seed = 3000, 42
    # weak scaling (bench.py): rank r generates [r * per_rank, (r + 1) * per_rank) of the global stream
    a, b = sw.generate_pairs("tokens64", per_rank, seed=seed, first=sharding.weak_shard_first(rank, per_rank))
    local = torch.from_numpy(oracle.levenshtein_pairs(a, b, algo="hyyro").astype(np.int32))
    got = sharding.gather_distances(local)
    if rank == 0:
        out["weak"] = got.numpy()
    # strong scaling over one fixed batch, count-balanced and cells-balanced (ragged) splits
    fa, fb = sw.generate_pairs("words16", 5001, seed=seed)
    for name, ranges in (("count", [sharding.shard_range(5001, r, world) for r in range(world)]),
                         ("cells", sharding.shard_ranges_by_cells(fa.lengths, fb.lengths, world))):
        lo, hi = ranges[rank]
        local = torch.from_numpy(oracle.levenshtein_pairs(fa.subview(lo, hi), fb.subview(lo, hi)).astype(np.int32))
        got = sharding.gather_distances(local, counts=[h - l for l, h in ranges])
        if rank == 0:
            out[name] = got.numpy()
            out[name + "_ranges"] = ranges
    # config C5's shape: ranges balanced on per-block cell sums (no rank sees all lengths), results gathered in three
    # pieces straight into their place on the root, every shard's checksum verified there (bench.py --config c5)
    total, block = 7003, 500
    lo, hi = sharding.shard_range(total, rank, world)
    ga, gb = sw.generate_pairs("short_words", hi - lo, seed=seed, first=lo)
    cells = (ga.lengths * gb.lengths).astype(np.int64)
    first_block, last_block = lo // block, (hi - 1) // block
    mine = np.zeros((total + block - 1) // block, dtype=np.int64)
    for k in range(first_block, last_block + 1):
        mine[k] = cells[max(k * block, lo) - lo:min((k + 1) * block, hi) - lo].sum()
    summed = torch.from_numpy(mine)
    dist.all_reduce(summed)
    ranges = sharding.shard_ranges_by_block_cells(summed.numpy(), block, total, world)
    lo, hi = ranges[rank]
    ga, gb = sw.generate_pairs("short_words", hi - lo, seed=seed, first=lo)
    local = torch.from_numpy(oracle.levenshtein_pairs(ga, gb, algo="hyyro").astype(np.int32))
    gather = sharding.ChunkedGather(ranges, 3, torch.int32, "cpu")
    for j in range(3):
        gather.send_chunk(local, j)
    full = gather.wait()
    sums = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sums, local.to(torch.int64).sum().reshape(1))
    # the same gather twice more into ONE root buffer handed in (bench.py keeps one per pipeline slot), the second time with
    # bytes on the wire (no distance of word-sized pairs exceeds 255) widened on arrival
    reused = torch.full((total,), -1, dtype=torch.int32) if rank == 0 else None
    narrow_ok = True
    for transport in (None, torch.uint8):
        if rank == 0:
            reused.fill_(-1)
        gather = sharding.ChunkedGather(ranges, 3, torch.int32, "cpu", full=reused, transport=transport)
        for j in range(3):
            gather.send_chunk(local, j)
        again = gather.wait()
        if rank == 0:
            narrow_ok = narrow_ok and again is reused and bool((again == full).all())
    if rank == 0:
        out["c5_reused_and_narrow_ok"] = narrow_ok
        out["c5"] = full.numpy()
        out["c5_ranges"] = ranges
        out["c5_sums_ok"] = all(int(full[l:h].to(torch.int64).sum()) == int(sums[r]) for r, (l, h) in enumerate(ranges))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        queue.put(out)


def test_two_rank_gloo_gather_matches_single_process(sw, orc):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    queue, port, world = ctx.Queue(), _free_port(), 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    out = queue.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = sw.generate_pairs("tokens64", 6000, seed=42)
    assert (out["weak"] == orc.levenshtein_pairs(a, b, algo="hyyro").astype(np.int32)).all()
    fa, fb = sw.generate_pairs("words16", 5001, seed=42)
    want = orc.levenshtein_pairs(fa, fb).astype(np.int32)
    assert (out["count"] == want).all() and (out["cells"] == want).all()
    (lo0, hi0), (lo1, hi1) = out["cells_ranges"]
    assert lo0 == 0 and hi0 == lo1 and hi1 == 5001
    cells = fa.lengths * fb.lengths
    assert abs(int(cells[lo0:hi0].sum()) - int(cells[lo1:hi1].sum())) <= int(cells.max())
    ga, gb = sw.generate_pairs("short_words", 7003, seed=42)
    assert (out["c5"] == orc.levenshtein_pairs(ga, gb, algo="hyyro").astype(np.int32)).all() and out["c5_sums_ok"]
    assert out["c5_reused_and_narrow_ok"]
    (lo0, hi0), (lo1, hi1) = out["c5_ranges"]
    cells = (ga.lengths * gb.lengths).astype(np.int64)
    assert lo0 == 0 and hi0 == lo1 and hi1 == 7003
    assert abs(int(cells[lo0:hi0].sum()) - int(cells[lo1:hi1].sum())) < 0.02 * int(cells.sum())


def _worker8(rank, world, port, queue):
    """World size 8, the shape of the first real node: ragged cells-balanced shards (cut on all-reduced per-block cell sums), four
    pieces per shard sent while the next is 'scored', u32 on the wire and bytes on the wire, into a root buffer that is reused."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import oracle
    import stringwars_amd as sw
    from stringwars_amd import sharding

    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, block, seed, chunks = 40_013, 1 << 10, 42, 4
    lo, hi = sharding.shard_range(total, rank, world)
    # a stream whose pairs get heavier towards its end: count-balanced and cells-balanced cuts differ by a lot
    def shard(first, count):
        a, b = sw.generate_pairs("short_words", count, seed=seed, first=first)
        keep = (np.arange(first, first + count) * 16 // total + 1).astype(np.int64)          # string i keeps at most 1 .. 16 bytes
        cut = lambda t: sw.Strs([bytes(t[i])[: int(keep[i])] for i in range(count)])
        return cut(a), cut(b)
    a, b = shard(lo, hi - lo)
    cells = (a.lengths.astype(np.int64) * b.lengths.astype(np.int64))
    mine = np.zeros((total + block - 1) // block, dtype=np.int64)
    for k in range(lo // block, (hi - 1) // block + 1):
        mine[k] = cells[max(k * block, lo) - lo:min((k + 1) * block, hi) - lo].sum()
    summed = torch.from_numpy(mine)
    dist.all_reduce(summed)
    ranges = sharding.shard_ranges_by_block_cells(summed.numpy(), block, total, world)
    lo, hi = ranges[rank]
    a, b = shard(lo, hi - lo)
    local = torch.from_numpy(oracle.levenshtein_pairs(a, b, algo="hyyro").astype(np.int32))
    my_cells = int((a.lengths.astype(np.int64) * b.lengths.astype(np.int64)).sum())
    all_cells = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(all_cells, torch.tensor([my_cells]))
    sums = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sums, torch.tensor([int(local.to(torch.int64).sum()), int(np.bitwise_xor.reduce(local.numpy())) if hi > lo else 0]))
    reused = torch.full((total,), -7, dtype=torch.int32) if rank == 0 else None
    results = []
    for transport in (None, torch.uint8, None):
        if rank == 0:
            reused.fill_(-7)
        gather = sharding.ChunkedGather(ranges, chunks, torch.int32, "cpu", full=reused, transport=transport)
        for j in range(chunks):
            gather.send_chunk(local, j)           # piece j leaves while piece j + 1 would be scored
        full = gather.wait()
        if rank == 0:
            ok = all(int(full[l:h].to(torch.int64).sum()) == int(sums[r][0]) and
                     (int(np.bitwise_xor.reduce(full[l:h].numpy())) if h > l else 0) == int(sums[r][1]) for r, (l, h) in enumerate(ranges))
            results.append((ok, full.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        queue.put({"ranges": ranges, "cells": [int(c) for c in all_cells], "results": results})


def test_eight_rank_gloo_chunked_gather_of_ragged_shards(sw, orc):
    """`ChunkedGather` at world size 8 (the node the scaling curve is taken on): cells-balanced RAGGED shards -- the stream's pairs get
    heavier towards its end, so the first rank's shard holds several times the pairs of the last --, four pieces per shard, u32 and
    u8 on the wire, the root's buffer reused; every rank's slice arrives in its place (per-rank sum and xor checked on the root)
    and the gathered vector is the single-process result."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    queue, port, world = ctx.Queue(), _free_port(), 8
    procs = [ctx.Process(target=_worker8, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    out = queue.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total = 40_013
    ranges = out["ranges"]
    assert ranges[0][0] == 0 and ranges[-1][1] == total and all(ranges[r][1] == ranges[r + 1][0] for r in range(7))
    counts = [h - l for l, h in ranges]
    assert max(counts) > 2 * min(counts)                                   # ragged: balanced on cells, not on pairs
    assert max(out["cells"]) - min(out["cells"]) < 0.05 * sum(out["cells"]) / 8
    a, b = sw.generate_pairs("short_words", total, seed=42)
    keep = (np.arange(total) * 16 // total + 1)
    a = sw.Strs([bytes(a[i])[: int(keep[i])] for i in range(total)]); b = sw.Strs([bytes(b[i])[: int(keep[i])] for i in range(total)])
    want = orc.levenshtein_pairs(a, b, algo="hyyro").astype(np.int32)
    assert len(out["results"]) == 3
    for ok, full in out["results"]:
        assert ok and (full == want).all()


def test_shard_helpers():
    from stringwars_amd import sharding
    for total in (0, 1, 7, 1000):
        for world in (1, 2, 3, 8):
            ranges = [sharding.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    assert sharding.chunk_ranges(10, 4) == [(0, 2), (2, 5), (5, 7), (7, 10)] and sharding.chunk_ranges(2, 4) == [(0, 1), (1, 2)]
    assert sharding.chunk_ranges(0, 4) == [(0, 0)]
    assert sharding.shard_ranges_by_block_cells([10, 10, 10, 10], 100, 400, 2) == [(0, 200), (200, 400)]
    assert sharding.shard_ranges_by_block_cells([30, 10], 100, 150, 2) == [(0, 67), (67, 150)]
    la = np.array([1, 100, 1, 1, 100, 1]); lb = np.array([1, 100, 1, 1, 100, 1])
    assert sharding.shard_ranges_by_cells(la, lb, 2) == [(0, 2), (2, 6)] or sharding.shard_ranges_by_cells(la, lb, 2)[0][1] in (2, 3, 4)


def test_c_abi_shard_cuts_match_the_python_partition(sw):
    """`swh_shard_cuts_*` (what `swh_sharded_prepare_*` cuts a batch with inside the library) == the Python reference
    partition, u32 and u64 offsets, degenerate batches included."""
    from stringwars_amd import sharding
    for workload, count in (("words16", 5001), ("tokens64", 777), ("short_words", 20_000)):
        a, b = sw.generate_pairs(workload, count, seed=9)
        for shards in (1, 2, 3, 8, 16):
            want = sharding.shard_ranges_by_cells(a.lengths, b.lengths, shards)
            want = [r[0] for r in want] + [count]
            assert sw.shard_cuts(a, b, shards) == want
            assert sw.shard_cuts(a.with_offsets(np.uint32), b.with_offsets(np.uint32), shards) == want
    empty = sw.Strs([b"", b"", b"", b""])
    assert sw.shard_cuts(empty, empty, 2) == [0, 2, 4]                      # no cells at all: equal counts
    one = sw.Strs([b"abc"])
    assert sw.shard_cuts(one, one, 4) in ([0, 0, 0, 0, 1], [0, 1, 1, 1, 1], [0, 0, 0, 1, 1], [0, 0, 1, 1, 1])
