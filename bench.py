#!/usr/bin/env python3
"""bench.py -- GCUPS of batched Levenshtein on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: `swh_levenshtein_pairs_u64tape` over the
config C2 workload (1,000,000 printable-ASCII token pairs, lengths U[32,96], unbounded; SURVEY.md
8d) held resident in HBM, plus -- for N > 1 -- the RCCL gather of the u32 distances to rank 0
that the north-star names. Weak scaling: every rank scores its own 1M-pair shard of the same
seeded stream (pair i depends only on (seed, i)).

CUPS accounting is the reference's (similarities/bench.rs:413-414): cells = sum len(a_i)*len(b_i),
whatever the algorithm skips. One JSON line is printed by rank 0.

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Peaks from /opt/skills/guides/MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6e12 32-bit
# integer lane-ops/s (the FP32 vector peak 157.3 TFLOP/s counts an FMA as two); HBM3E 8 TB/s.
PEAK_VALU_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
PEAK_HBM_GBS = 8000.0
OPS_PER_CELL = 5  # compare, add-diagonal, min(up,left), +1, min


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--workload", default="tokens64")
    p.add_argument("--pairs", type=int, default=1_000_000, help="pairs per GPU")
    p.add_argument("--algorithm", default="auto", choices=["auto", "wavefront", "bitparallel"])
    p.add_argument("--seed", type=int, default=int(os.environ.get("STRINGWARS_SEED", "42")))
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo + --share-gpu exercises the multi-rank control flow on a one-GPU box (testing only)")
    p.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (testing only; never a result)")
    return p.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import stringwars_amd as sw

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

    # ---- this rank's shard of the seeded stream, made resident in HBM (torch = allocator + stream) ----
    a, b = sw.generate_pairs(args.workload, args.pairs, seed=args.seed, first=rank * args.pairs)
    cells = int((a.lengths * b.lengths).sum())
    tensors = [torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x).to(device)
               for x in (a.data, a.offsets, b.data, b.offsets)]
    da = sw.DeviceTape(tensors[0].data_ptr(), tensors[1].data_ptr(), a.count, np.uint64, keepalive=tensors[:2])
    db = sw.DeviceTape(tensors[2].data_ptr(), tensors[3].data_ptr(), b.count, np.uint64, keepalive=tensors[2:])
    # Two result buffers: the RCCL gather of step i overlaps the kernels of step i+1 (it runs on the process
    # group's own stream); a buffer is reused only after its gather has been waited for.
    outs = [torch.zeros(args.pairs, dtype=torch.int32, device=device) for _ in range(2)]
    out = outs[0]
    comm_device = device if args.backend == "nccl" else torch.device("cpu")
    gathered = [[torch.zeros(args.pairs, dtype=torch.int32, device=comm_device) for _ in range(world)] for _ in range(2)] \
        if rank == 0 and world > 1 else [None, None]
    works = [None, None]

    scope = sw.DeviceScope(gpu_device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
    engine = sw.LevenshteinDistances(capabilities=scope, algorithm=args.algorithm)
    scope.set_async(True)
    scope.set_pipelined(True)   # step i+1's planning pre-pass overlaps step i's DP kernel (two internal lanes)
    counter = [0]

    def step():
        slot = counter[0] & 1
        counter[0] += 1
        if works[slot] is not None:
            works[slot].wait()
            works[slot] = None
        engine.pairs(da, db, scope, out=outs[slot])
        if world > 1:
            scope.join()            # the gather is ordered on torch's stream: make that stream wait for this call
            if args.backend == "nccl":
                works[slot] = dist.gather(outs[slot], gathered[slot], dst=0, async_op=True)
            else:
                dist.gather(outs[slot].cpu(), gathered[slot], dst=0)

    def fence():
        for slot in range(2):
            if works[slot] is not None:
                works[slot].wait()
                works[slot] = None
        scope.synchronize()         # both pipeline lanes
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([cells], dtype=torch.int64, device=comm_device)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_cells = int(c.item())
    else:
        total_cells = cells

    # ---- roofline of the dominant kernel: hipEvents on the kernels' own streams, inside the library ------
    # The timed region is repeated with the library's event pairs switched on (they cost a few microseconds of bubbles
    # per call, so `value` above comes from the run without them): same pipelined steps, same overlap between one
    # lane's DP kernel and the other lane's planning pre-pass -- the conditions `rocprofv3 --kernel-trace --stats` of
    # this command averages over. The mean over those K launches is the kernel duration of the roofline.
    scope.set_profiling(True)
    for _ in range(args.steps):
        step()
    fence()
    totals = scope.timing_totals()
    scope.set_profiling(False)
    scope.set_pipelined(False)
    scope.set_async(False)
    engine.pairs(da, db, scope, out=out)   # one synchronous call: cells / bytes / kernel names of a launch
    scope.set_profiling(True)
    engine.pairs(da, db, scope, out=out)
    samples = [scope.last_timing()]
    scope.set_profiling(False)
    # the bit-parallel path scores every pair in ONE launch; the wavefront path launches one kernel per
    # columns-per-lane class, so its "dominant kernel" is the family and its duration their sum
    dominant_ms = totals["compute_ms"] / max(totals["calls"], 1)
    kernels_ms = totals["total_ms"] / max(totals["calls"], 1)
    timing = samples[-1]
    algorithmic_bytes = timing["bytes"]
    valu_tops = OPS_PER_CELL * cells / (dominant_ms * 1e-3) / 1e12
    hbm_gbs = algorithmic_bytes / (dominant_ms * 1e-3) / 1e9
    roofline = {
        "bound": "valu", "kernel": timing["dominant_name"] if args.algorithm != "wavefront" else "wavefront_* (all classes)",
        "kernel_ms": round(dominant_ms, 4),
        "achieved": round(valu_tops, 3), "peak": round(PEAK_VALU_TOPS, 1), "unit": "Tint32op/s",
        "frac": round(valu_tops / PEAK_VALU_TOPS, 4), "traffic": None,
        "ops_per_cell": OPS_PER_CELL, "cells_per_launch": cells,
        "hbm": {"achieved": round(hbm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(hbm_gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes": algorithmic_bytes},
        "all_kernels_ms": round(kernels_ms, 4), "launches_timed": totals["calls"],
        "kernel_ms_unoverlapped": round(samples[-1]["compute_ms"], 4),
    }

    result_host = out.cpu().numpy().astype(np.uint32)
    line = None
    gather_ok = None
    if world > 1 and rank == 0:
        # the gathered vector must hold every rank's shard in rank order: rank 0's own slice is checked bit for bit,
        # the others by a cheap invariant (distances are bounded by the longer string of the pair)
        last = (counter[0] - 1) & 1
        gather_ok = bool((gathered[last][0].cpu().numpy().astype(np.uint32) == outs[last].cpu().numpy().astype(np.uint32)).all())
    if rank == 0:
        cpu_baseline = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            import oracle  # checker + reported baseline only; never on the timed GPU path
            check = min(args.pairs, 20_000)
            parity = bool((oracle.levenshtein_pairs(a, b, algo="hyyro", count=check) == result_host[:check]).all())
            sample_pairs = min(args.pairs, 1_000_000)
            sample_cells = int((a.lengths[:sample_pairs] * b.lengths[:sample_pairs]).sum())
            repeats, t_cpu = 0, 0.0
            c0 = time.perf_counter()
            while t_cpu < 10.0:
                oracle.levenshtein_pairs(a, b, algo="hyyro", count=sample_pairs)
                repeats += 1
                t_cpu = time.perf_counter() - c0
            cpu_baseline = {
                "value": round(sample_cells * repeats / t_cpu / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "port",
                "sample": f"first {sample_pairs} pairs of the same workload x {repeats} repeats, "
                          "oracle Hyyro/Myers 64-bit bit-parallel (the algorithm family of rapidfuzz), one pair per call",
            }
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "GCUPS (DP cell updates/s) batched Levenshtein", "value": round(total_cells * args.steps / elapsed / 1e9, 2),
            "unit": "GCUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"C2 {args.workload}: {args.pairs} ASCII token pairs per GPU, lengths U[32,96], "
                                   "unbounded Levenshtein, inputs resident in HBM",
                       "pairs_per_gpu": args.pairs, "cells_per_gpu": cells, "algorithm": args.algorithm,
                       "collective": "RCCL gather of u32 distances to rank 0" if world > 1 else "none",
                       "seed": args.seed},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_vs_oracle": parity, "gather_ok": gather_ok,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
