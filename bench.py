#!/usr/bin/env python3
"""bench.py -- GCUPS of batched Levenshtein on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch held resident in HBM, tapes prepared once outside the timed
region (as the reference builds its tape views once, bench.rs:292-306), plus -- for N > 1 -- the RCCL gather of the
u32 distances to rank 0 that the north-star names.

  --config c2 (default)  BASELINE configs[1]: 1,000,000 printable-ASCII token pairs per GPU, lengths U[32,96],
                         unbounded. Weak scaling: rank r scores pairs [r*P, (r+1)*P) of the seeded stream.
  --config c5            BASELINE configs[4]: 100,000,000 short-word pairs (<= 16 B) in total, strong scaling: the
                         ranks split the stream into cells-balanced contiguous shards, score them in pieces and
                         send each piece to rank 0 (ncclSend / ncclRecv group) while the next one is scored.

CUPS accounting is the reference's (similarities/bench.rs:413-414): cells = sum len(a_i)*len(b_i), whatever the
algorithm skips. `value` is the rate of the K timed steps, enqueued asynchronously (two internal lanes); the
like-for-like figure for the reference's synchronous `compute_into` (bench.rs:478-486, utils.rs:721-799) is
`value_sync_call`. One JSON line is printed by rank 0.

Order of a run: data, tapes, `--prewarm-seconds` (default 0.5) of untimed steps that bring an idle device to its clocks
(`config.device_prewarm_s`; the same count on every rank), the W warm-up steps, the K timed steps between
barrier + synchronize on both sides, then the checks (oracle, checksums), the synchronous calls, the CPU rows.

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Peaks from /opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 32-bit
# integer lane-ops/s (the FP32 vector peak 157.3 TFLOP/s counts an FMA as two); HBM3E 8 TB/s.
PEAK_VALU_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
PEAK_HBM_GBS = 8000.0
NOMINAL_OPS_PER_CELL = 5  # SURVEY.md 8d's scalar-DP model: compare, add-diagonal, min(up,left), +1, min
# Executed VALU instructions and HBM traffic per launch come from rocprofv3 PMC passes over this very command
# (tools/profile_pmc.sh -> tools/pmc_constants.py); they cannot be read from inside the process.
PMC_CONSTANTS = os.path.join(ROOT, "profiles", "r2", "pmc_constants.json")

CONFIGS = {
    "c2": dict(workload="tokens64", pairs=1_000_000, scaling="weak",
               text="C2 tokens64: {pairs} ASCII token pairs per GPU, lengths U[32,96], unbounded Levenshtein"),
    "c5": dict(workload="short_words", pairs=100_000_000, scaling="strong",
               text="C5 short_words: {pairs} word pairs (<= 16 B, mean ~6) in total, split over the GPUs, unbounded Levenshtein"),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--prewarm-seconds", type=float, default=0.5, help="untimed steps before the W warm-up steps, by the clock: brings an idle device to its clocks (0: none)")
    p.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    p.add_argument("--workload", default=None, help="override the config's synthetic workload")
    p.add_argument("--pairs", type=int, default=None, help="pairs per GPU (weak configs) / in total (strong configs)")
    p.add_argument("--chunks", type=int, default=0,
                   help="pieces a shard is scored and gathered in (strong configs); 0 = 4 when there is a gather to overlap, 1 on one GPU")
    p.add_argument("--algorithm", default="auto", choices=["auto", "wavefront", "bitparallel", "tiled"])
    p.add_argument("--seed", type=int, default=int(os.environ.get("STRINGWARS_SEED", "42")))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo + --share-gpu exercises the multi-rank control flow on a one-GPU box (testing only)")
    p.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (testing only; never a result)")
    return p.parse_args()


def load_pmc_constants():
    try:
        with open(PMC_CONSTANTS) as handle:
            return json.load(handle)
    except (OSError, ValueError):
        return {"kernels": {}}


def roofline_of(kernel, kernel_ms, cells, algorithmic_bytes, workload, pairs, constants, extra=None):
    """Roofline object of one kernel launch. `achieved` is EXECUTED work: SQ_INSTS_VALU (wave instructions, PMC) x 64
    lanes / the kernel's duration measured live with hipEvents on its own stream; `frac` = achieved / the integer
    VALU peak. SURVEY 8d's nominal 5-ops-per-cell model is carried separately (a bit-parallel kernel executes ~1.2
    lane-ops per cell, so that figure exceeds the peak by construction and is not a roofline fraction)."""
    seconds = kernel_ms * 1e-3
    entry = constants.get("kernels", {}).get(f"{kernel}|{workload}")
    scale = pairs / entry["pairs_per_launch"] if entry else 1.0   # the counters were taken on launches of pairs_per_launch pairs
    hbm_gbs = algorithmic_bytes / seconds / 1e9 if seconds > 0 else 0.0
    roof = {
        "bound": "valu", "kernel": kernel, "kernel_ms": round(kernel_ms, 4), "unit": "Tint32op/s", "peak": round(PEAK_VALU_TOPS, 1),
        "achieved": None, "frac": None, "traffic": None, "cells_per_launch": cells,
        "nominal_ops_per_cell_equiv": round(NOMINAL_OPS_PER_CELL * cells / seconds / 1e12, 3) if seconds > 0 else None,
        "hbm": {"achieved": round(hbm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(hbm_gbs / PEAK_HBM_GBS, 4),
                "algorithmic_bytes": algorithmic_bytes},
    }
    if entry and seconds > 0:
        lane_ops = entry["valu_insts"] * scale * 64
        roof["achieved"] = round(lane_ops / seconds / 1e12, 3)
        roof["frac"] = round(lane_ops / seconds / 1e12 / PEAK_VALU_TOPS, 4)
        roof["valu_wave_insts_per_launch"] = round(entry["valu_insts"] * scale, 1)
        roof["lane_ops_per_cell"] = round(lane_ops / max(cells, 1), 3)
        if entry.get("fetch_kb") is not None and entry.get("write_kb") is not None:
            # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of the bytes
            # of wide reads (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE as is. Infinity-Cache hits are counted.
            # Calibrated on this box (profiles/r2/fetch_calibration.txt): x2 holds for coalesced streams and for windows in
            # adjacent sectors, a lone 64-byte sector is fetched and tallied as 64 B (x1) -- kernels whose work items read
            # scattered 16-byte windows lie between `vs_algorithmic_lower` (x1) and `vs_algorithmic` (x2).
            roof["traffic"] = int((entry["fetch_kb"] * 1024 * 2 + entry["write_kb"] * 1024) * scale)
            lower = (entry["fetch_kb"] * 1024 + entry["write_kb"] * 1024) * scale
            roof["traffic_detail"] = {"fetch_kb_raw": round(entry["fetch_kb"] * scale, 1), "write_kb_raw": round(entry["write_kb"] * scale, 1),
                                      "fetch_correction": 2, "vs_algorithmic": round(roof["traffic"] / max(algorithmic_bytes, 1), 2),
                                      "vs_algorithmic_lower": round(lower / max(algorithmic_bytes, 1), 2)}
        roof["pmc_source"] = f"{entry.get('source')}; launches of {entry['pairs_per_launch']} pairs"
    else:
        roof["note"] = f"no PMC constants for {kernel}|{workload} in profiles/r2/pmc_constants.json"
    if extra:
        roof.update(extra)
    return roof


def cpu_rows(a, b, budget_s=6.0):
    """The CPU rows BASELINE.md section 3 lists, on the host cores of this box, same inputs, bounded samples:
    hyyro<1cpu> (the row that corresponds to rapidfuzz::levenshtein<Bytes,1cpu>, bench.rs:407), hyyro<Ncpu>,
    wagner_fischer<1cpu> (~ bio::levenshtein, bench.rs:443-459)."""
    import oracle  # checker + reported baseline only; never on the timed GPU path
    rows = []
    pairs = len(a.offsets) - 1

    def timed(name, cores, fn, sample_pairs, what):
        cells = int((a.lengths[:sample_pairs] * b.lengths[:sample_pairs]).sum())
        repeats, start = 0, time.perf_counter()
        while True:
            fn(sample_pairs)
            repeats += 1
            spent = time.perf_counter() - start
            if spent >= budget_s:
                break
        rows.append({"name": name, "value": round(cells * repeats / spent / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": "port",
                     "sample": f"first {sample_pairs} pairs of the same workload x {repeats} repeats, {what}"})

    one = min(pairs, 1_000_000)
    timed("cpu::hyyro<1cpu>", 1, lambda n: oracle.levenshtein_pairs(a, b, algo="hyyro", count=n), one,
          "oracle Hyyro/Myers 64-bit bit-parallel (the algorithm family of rapidfuzz), one pair per call")
    cores = min(os.cpu_count() or 1, 64)
    if cores > 1:
        pool = ThreadPoolExecutor(cores)

        def sharded(n):
            bounds = [n * t // cores for t in range(cores + 1)]
            jobs = [pool.submit(oracle.levenshtein_pairs, a, b, False, "hyyro", None, bounds[t], bounds[t + 1] - bounds[t]) for t in range(cores)]
            for job in jobs:
                job.result()
        timed(f"cpu::hyyro<{cores}cpu>", cores, sharded, one, f"the same over {cores} host threads (contiguous slices)")
        pool.shutdown()
    timed("cpu::wagner_fischer<1cpu>", 1, lambda n: oracle.levenshtein_pairs(a, b, algo="wf", count=n), min(pairs, 100_000),
          "oracle two-row Wagner-Fischer (the algorithm of bio::levenshtein)")
    # cpu::gotoh<1cpu> (~ bio::pairwise::Aligner::global, bench.rs:746-765) has no pairs of this workload to run on: it is
    # timed on config C4's shape (4 KB amino-acid sequences, 256x256 i8 matrix, affine gaps), cells of that sample
    import stringwars_amd as sw
    pa, pb = sw.generate_pairs("protein4k", 4, seed=42)
    matrix = sw.substitution_matrix(42)
    gotoh_cells = int((pa.lengths * pb.lengths).sum())
    repeats, start = 0, time.perf_counter()
    while True:
        oracle.nw_pairs(pa, pb, matrix, -11, -1)
        repeats += 1
        spent = time.perf_counter() - start
        if spent >= budget_s / 2:
            break
    rows.append({"name": "cpu::gotoh<1cpu>", "value": round(gotoh_cells * repeats / spent / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "port",
                 "sample": f"4 pairs of config C4 (protein4k, affine -11/-1) x {repeats} repeats, oracle Gotoh score-only DP "
                           "(bio::pairwise also keeps traceback matrices)"})
    return rows


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import stringwars_amd as sw
    from stringwars_amd import sharding

    cfg = CONFIGS[args.config]
    workload = args.workload or cfg["workload"]
    strong = cfg["scaling"] == "strong"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    comm_device = device if args.backend == "nccl" else torch.device("cpu")

    # ---- this rank's shard of the seeded stream -------------------------------------------------------------------
    if strong:
        total_pairs = args.pairs or cfg["pairs"]
        lo, hi = sharding.shard_range(total_pairs, rank, world)
        if world > 1:
            # cells-balanced cuts without any rank holding all the lengths: per-block cell sums of the count-balanced
            # slices, all-reduced, then cut on their prefix (sharding.shard_ranges_by_block_cells)
            block = 1 << 20
            a, b = sw.generate_pairs(workload, hi - lo, seed=args.seed, first=lo)
            pair_cells = (a.lengths * b.lengths).astype(np.int64)
            sums = np.zeros((total_pairs + block - 1) // block, dtype=np.int64)
            for k in range(lo // block, (hi - 1) // block + 1 if hi > lo else 0):
                sums[k] = pair_cells[max(k * block, lo) - lo:min((k + 1) * block, hi) - lo].sum()
            reduced = torch.from_numpy(sums).to(comm_device)
            dist.all_reduce(reduced)
            ranges = sharding.shard_ranges_by_block_cells(reduced.cpu().numpy(), block, total_pairs, world)
            if ranges[rank] != (lo, hi):
                lo, hi = ranges[rank]
                a, b = sw.generate_pairs(workload, hi - lo, seed=args.seed, first=lo)
        else:
            ranges = [(0, total_pairs)]
            a, b = sw.generate_pairs(workload, total_pairs, seed=args.seed)
        pairs = hi - lo
    else:
        pairs = args.pairs or cfg["pairs"]
        total_pairs = pairs * world
        ranges = [(r * pairs, (r + 1) * pairs) for r in range(world)]
        a, b = sw.generate_pairs(workload, pairs, seed=args.seed, first=sharding.weak_shard_first(rank, pairs))
    cells = int((a.lengths * b.lengths).sum())
    # per-shard u32 offsets (half the offset traffic of u64) whenever the shard's bytes fit them
    if int(a.offsets[-1]) < 2 ** 32 and int(b.offsets[-1]) < 2 ** 32:
        a, b = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
    offsets_dtype = a.offsets.dtype
    as_torch = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else (x.view(np.int32) if x.dtype == np.uint32 else x)).to(device)
    tensors = [as_torch(x) for x in (a.data, a.offsets, b.data, b.offsets)]
    da = sw.DeviceTape(tensors[0].data_ptr(), tensors[1].data_ptr(), a.count, offsets_dtype, keepalive=tensors[:2])
    db = sw.DeviceTape(tensors[2].data_ptr(), tensors[3].data_ptr(), b.count, offsets_dtype, keepalive=tensors[2:])

    scope = sw.DeviceScope(gpu_device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm=args.algorithm)
    # prepared once, outside every timed region: resident, measured (the engine then needs no planning pre-pass)
    pa, pb = sw.PreparedTape(scope, da), sw.PreparedTape(scope, db)

    # ---- the step ------------------------------------------------------------------------------------------------------
    outs = [torch.zeros(max(pairs, 1), dtype=torch.int32, device=device) for _ in range(2)]
    counter = [0]
    if strong:
        if args.chunks <= 0:
            args.chunks = 4 if world > 1 else 1
        pieces = sharding.chunk_ranges(pairs, args.chunks)
        gathers = [None, None]
        # one pre-bound call per (buffer, piece): sub-views of the prepared tapes, the piece's slice of the result buffer
        piece_calls = [[engine.bind_pairs(pa[p_lo:p_hi], pb[p_lo:p_hi], scope, outs[slot][p_lo:p_hi]) if p_hi > p_lo else None
                        for p_lo, p_hi in pieces] for slot in range(2)]

        def step():
            slot = counter[0] & 1
            counter[0] += 1
            if gathers[slot] is not None:
                gathers[slot].wait()            # this buffer's previous gather has left it
            gather = sharding.ChunkedGather(ranges, args.chunks, torch.int32, device) if world > 1 else None
            for j, (p_lo, p_hi) in enumerate(pieces):
                if p_hi > p_lo:
                    piece_calls[slot][j]()
                if gather is not None:
                    scope.join()                # the send is ordered on torch's stream: make it wait for this piece
                    gather.send_chunk(outs[slot], j)
            gathers[slot] = gather
    else:
        gathered = [[torch.zeros(pairs, dtype=torch.int32, device=comm_device) for _ in range(world)] for _ in range(2)] \
            if rank == 0 and world > 1 else [None, None]
        works = [None, None]
        gathers = works
        calls = [engine.bind_pairs(pa, pb, scope, outs[slot]) for slot in range(2)]

        def step():
            slot = counter[0] & 1
            counter[0] += 1
            if works[slot] is not None:
                works[slot].wait()
                works[slot] = None
            calls[slot]()               # engine.pairs(pa, pb, scope, out=outs[slot]) with its arguments bound once
            if world > 1:
                scope.join()            # the gather is ordered on torch's stream: make that stream wait for this call
                if args.backend == "nccl":
                    works[slot] = dist.gather(outs[slot], gathered[slot], dst=0, async_op=True)
                else:
                    dist.gather(outs[slot].cpu(), gathered[slot], dst=0)

    def fence():
        for slot in range(2):
            if gathers[slot] is not None:
                gathers[slot].wait()
                if not strong:
                    gathers[slot] = None
        scope.synchronize()         # both pipeline lanes
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(steps):
        fence()
        start = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        return time.perf_counter() - start

    # `value`: K steps enqueued asynchronously on two internal lanes (host-side work of step i+1 overlaps step i)
    scope.set_async(True)
    scope.set_pipelined(True)
    # An idle MI355X takes a few hundred milliseconds of work to reach its clocks (profiles/r2: the same 20 steps measure
    # 8 % apart right after start-up and after a second of calls). The device is brought there first -- every rank runs the same
    # number of untimed steps, the collective included --, then come the W warm-up steps and the K timed ones.
    if args.prewarm_seconds > 0:
        fence()
        start = time.perf_counter()
        for _ in range(8):                      # what a step costs here
            step()
        fence()
        per_step = (time.perf_counter() - start) / 8
        extra = int(min(max(args.prewarm_seconds / max(per_step, 1e-6) - 8, 0), 100000))
        if world > 1:                           # the same count on every rank: a step contains the collective
            agreed = torch.tensor([extra], dtype=torch.int64, device=comm_device)
            dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
            extra = int(agreed.item())
        for i in range(extra):
            step()
            if i % 64 == 63:
                fence()
    for _ in range(args.warmup):
        step()
    elapsed = timed_region(args.steps)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([cells], dtype=torch.int64, device=comm_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = int(c.item())
    else:
        total_cells = cells
    last_slot = (counter[0] - 1) & 1
    pipelined_result = outs[last_slot][:pairs].cpu().numpy().astype(np.uint32)   # produced by the timed steps

    # ---- every rank's slice of the gathered vector, against a checksum the rank computed locally ---------------------
    gather_ok = None
    if world > 1:
        mine = torch.tensor([int(pipelined_result.astype(np.int64).sum()), int(np.bitwise_xor.reduce(pipelined_result)) if pairs else 0],
                            dtype=torch.int64, device=comm_device)
        sums = [torch.zeros(2, dtype=torch.int64, device=comm_device) for _ in range(world)]
        dist.all_gather(sums, mine)
        if rank == 0:
            if strong:
                full = gathers[last_slot].full.cpu().numpy().astype(np.uint32)
                slices = [full[l:h] for l, h in ranges]
            else:
                slices = [g.cpu().numpy().astype(np.uint32) for g in gathered[last_slot]]
            gather_ok = all(int(s.astype(np.int64).sum()) == int(sums[r][0]) and
                            (int(np.bitwise_xor.reduce(s)) if s.size else 0) == int(sums[r][1]) for r, s in enumerate(slices))
            gather_ok = bool(gather_ok and (slices[0] == pipelined_result).all())

    # ---- the same region with the library's hipEvent pairs on (kernel durations on the kernels' own streams) ---------
    # They cost a few microseconds of bubbles per call, so `value` above comes from the run without them; same
    # steps, same overlap -- the conditions `rocprofv3 --kernel-trace --stats` of this command averages over.
    scope.set_profiling(True)
    timed_region(args.steps)
    totals = scope.timing_totals()
    scope.set_profiling(False)
    scope.set_pipelined(False)
    scope.set_async(False)

    # ---- the reference's own metric: synchronous calls (results visible on return), timed one by one -----------------
    out = outs[0]
    sync_call = engine.bind_pairs(pa, pb, scope, out)   # arguments bound once: the loop below only crosses the FFI
    sync_call()
    torch.cuda.synchronize()
    sync_start = time.perf_counter()
    for _ in range(args.steps):
        sync_call()
    sync_elapsed = time.perf_counter() - sync_start
    scope.set_profiling(True)
    engine.pairs(pa, pb, scope, out=out)
    sync_timing = scope.last_timing()
    scope.set_profiling(False)
    sync_result = out[:pairs].cpu().numpy().astype(np.uint32)

    line = None
    if rank == 0:
        constants = load_pmc_constants()
        calls = max(totals["calls"], 1)
        dominant = sync_timing["dominant_name"]
        kernel_ms = totals["compute_ms"] / calls
        per_call_cells = cells / (len(pieces) if strong else 1)
        per_call_bytes = sync_timing["bytes"] / (len(pieces) if strong else 1)
        roofline = roofline_of(dominant, kernel_ms, int(per_call_cells), int(per_call_bytes), workload,
                               pairs // (len(pieces) if strong else 1), constants,
                               extra={"all_kernels_ms": round(totals["total_ms"] / calls, 4), "launches_timed": totals["calls"],
                                      "measured": "hipEvents inside the library over a repeat of the timed region"})
        if roofline.get("achieved") is not None:
            # two launches overlap in the pipelined region (each then takes about twice as long as alone): per launch the
            # fraction above halves, per DEVICE it is the work of one launch over the step time
            step_s = elapsed / args.steps / (len(pieces) if strong else 1)
            roofline["frac_device"] = round(roofline["valu_wave_insts_per_launch"] * 64 / step_s / 1e12 / PEAK_VALU_TOPS, 4)
            roofline["frac_device_is"] = "executed lane-ops of one launch / time per launch of the timed region (launches of two lanes overlap)"
        roofline["sync_call"] = roofline_of(sync_timing["dominant_name"], sync_timing["compute_ms"], sync_timing["cells"], sync_timing["bytes"],
                                            workload, pairs, constants, extra={"measured": "one synchronous call on an idle GPU"})
        parity = None
        cpu_baseline, cpu_baselines = None, None
        if not args.no_cpu_baseline:
            import oracle  # checker + reported baseline only; never on the timed GPU path
            check = min(pairs, 20_000)   # (rank 0's shard; the other ranks' slices are covered by `gather_ok`)
            want = oracle.levenshtein_pairs(a, b, algo="hyyro", count=check)
            parity = bool((want == pipelined_result[:check]).all() and (want == sync_result[:check]).all()
                          and (pipelined_result == sync_result).all())
            if world == 1:   # the CPU baseline is timed at N = 1 only
                cpu_baselines = cpu_rows(a, b)
                cpu_baseline = {k: v for k, v in cpu_baselines[0].items() if k != "name"}
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "GCUPS (DP cell updates/s) batched Levenshtein", "value": round(total_cells * args.steps / elapsed / 1e9, 2),
            "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "value_sync_call": round(cells * args.steps / sync_elapsed / 1e9, 2),
            "config": {"workload": cfg["text"].format(pairs=total_pairs if strong else pairs) + ", tapes prepared and resident in HBM",
                       "value_is": "rate of the K timed steps, calls enqueued asynchronously on two internal lanes",
                       "value_sync_call_is": "rank 0's shard, K synchronous calls timed by the host (the reference's compute_into metric, utils.rs:721-799)",
                       "pairs_per_gpu": pairs, "pairs_total": total_pairs, "cells_per_gpu": cells, "algorithm": args.algorithm,
                       "offsets": str(offsets_dtype), "pieces_per_step": len(pieces) if strong else 1, "device_prewarm_s": args.prewarm_seconds,
                       "collective": ("ncclSend/ncclRecv group per piece to rank 0 (variable-size gather of u32 distances)" if strong
                                      else "RCCL gather of u32 distances to rank 0") if world > 1 else "none",
                       "seed": args.seed},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "cpu_baselines": cpu_baselines,
            "parity_vs_oracle": parity, "gather_ok": gather_ok,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
