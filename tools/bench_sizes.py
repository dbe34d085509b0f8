#!/usr/bin/env python3
"""Measuring tool: config C2 (ASCII token pairs ~64 B, unbounded Levenshtein, tapes prepared) at 1 M / 4 M / 16 M pairs -- 128 MB / 512 MB /
2 GB of tapes, i.e. inside, around and far beyond the 256 MiB Infinity Cache -- per synchronous call. The tiled kernel reads a tile's
strings about twice (the affix windows of its planning step, then the work items' own windows): on 1 M pairs the second read comes out
of the Infinity Cache; this table says what it costs when it cannot.
    python tools/bench_sizes.py                       # one JSON line per size
    python tools/bench_sizes.py --pairs 16000000 --calls 3   # exactly 3 calls of one size (rocprofv3 --pmc FETCH_SIZE passes)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stringwars_amd as sw

p = argparse.ArgumentParser()
p.add_argument("--pairs", type=int, default=0)
p.add_argument("--calls", type=int, default=0)
args = p.parse_args()
scope = sw.DeviceScope(gpu_device=0)
engine = sw.LevenshteinDistances(capabilities=scope)
for pairs in ([args.pairs] if args.pairs else [1_000_000, 4_000_000, 16_000_000]):
    a, b = sw.generate_pairs("tokens64", pairs, seed=42)
    a, b = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
    cells = int((a.lengths.astype(np.int64) * b.lengths.astype(np.int64)).sum())
    da, db = a.to_device(scope), b.to_device(scope)
    pa, pb = sw.PreparedTape(scope, da), sw.PreparedTape(scope, db)
    out = torch.zeros(pairs, dtype=torch.int32, device="cuda")
    call = engine.bind_pairs(pa, pb, scope, out)
    if args.calls:
        for _ in range(args.calls):
            call()
        torch.cuda.synchronize()
        print(json.dumps({"pairs": pairs, "calls": args.calls}))
        continue
    call()
    until = time.perf_counter() + 0.4
    while time.perf_counter() < until:
        call()
    walls = []
    for _ in range(9):
        t0 = time.perf_counter(); call(); walls.append(time.perf_counter() - t0)
    scope.set_profiling(True); call(); timing = scope.last_timing(); scope.set_profiling(False)
    sample = min(pairs, 200_000)
    import oracle   # the checker: a sample of the last call's results
    ok = bool((out[:sample].cpu().numpy().astype(np.uint32) == oracle.levenshtein_pairs(a, b, algo="hyyro", count=sample)).all())
    print(json.dumps({"pairs": pairs, "tape_bytes": int(a.data.nbytes + b.data.nbytes), "cells": cells,
                      "tcups_best_call": round(cells / min(walls) / 1e12, 3), "tcups_median_call": round(cells / sorted(walls)[len(walls) // 2] / 1e12, 3),
                      "call_ms_best": round(min(walls) * 1e3, 4), "kernel": timing["dominant_name"], "kernel_ms": round(timing["compute_ms"], 4),
                      "parity_sample_ok": ok}), flush=True)
    del call, pa, pb, da, db, out
    torch.cuda.empty_cache()
