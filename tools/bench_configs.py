#!/usr/bin/env python3
"""Per-config throughput table (BASELINE.json configs C1..C5 at full or stated size), with the kernel
breakdown the library's hipEvent stamps give. Not the driver contract (that is bench.py): a measuring tool.

    python tools/bench_configs.py [--configs c1,c2,c3,c4,c5] [--repeats 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringwars_amd as sw  # noqa: E402

CONFIGS = {
    "c1": dict(workload="words16", pairs=10_000, kind="lev"),
    "c2": dict(workload="tokens64", pairs=1_000_000, kind="lev"),
    "c3": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8", bound=32),
    "c3u": dict(workload="utf8_lines", pairs=100_000, kind="lev_utf8"),
    "c3b": dict(workload="utf8_lines", pairs=100_000, kind="lev"),
    "c4": dict(workload="protein4k", pairs=10_000, kind="nw", gaps=(-4, -4)),
    "c4a": dict(workload="protein4k", pairs=10_000, kind="nw", gaps=(-11, -1)),
    "c4b": dict(workload="bytes4k", pairs=2_000, kind="nw", gaps=(-4, -4)),
    "c4l": dict(workload="protein4k", pairs=10_000, kind="lev"),
    "c5": dict(workload="short_words", pairs=20_000_000, kind="lev"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,c2,c3,c4,c5")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--algorithm", default="auto")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--warm-seconds", type=float, default=0.3, help="untimed calls before the measured ones (clock ramp)")
    ap.add_argument("--offsets", default="u64", choices=["u32", "u64"])
    ap.add_argument("--prepared", action="store_true", help="prepare the tapes once (swh_tape_prepare_*) outside the timed calls")
    args = ap.parse_args()
    scope = sw.DeviceScope(gpu_device=0)
    for name in args.configs.split(","):
        cfg = CONFIGS[name]
        pairs = max(1, int(cfg["pairs"] * args.scale))
        t0 = time.perf_counter()
        a, b = sw.generate_pairs(cfg["workload"], pairs, seed=42)
        gen_s = time.perf_counter() - t0
        if args.offsets == "u32":
            a, b = a.with_offsets(np.uint32), b.with_offsets(np.uint32)
        da, db = a.to_device(scope), b.to_device(scope)
        if args.prepared:
            utf8 = cfg["kind"] == "lev_utf8"
            da, db = sw.PreparedTape(scope, da, utf8=utf8), sw.PreparedTape(scope, db, utf8=utf8)
        # results stay on the device, like `UnifiedMat` in the reference (bench.rs:466-476): no D2H in the timed call
        import ctypes as C
        from stringwars_amd import _native as N
        out_ptr, err = C.c_void_p(), C.c_char_p()
        N.check(N.lib.swh_device_alloc(scope.handle, pairs * 4 + 16, C.byref(out_ptr), C.byref(err)), err)
        out = int(out_ptr.value)
        if cfg["kind"] == "nw":
            alphabet = None if cfg["workload"] == "bytes4k" else sw.synth.AMINO_ACIDS
            engine = sw.NeedlemanWunschScores(substitution_matrix=sw.substitution_matrix(42, alphabet), open=cfg["gaps"][0],
                                              extend=cfg["gaps"][1], capabilities=scope)
            call = lambda: engine.pairs(da, db, scope, out=out)
        else:
            cls = sw.LevenshteinDistancesUTF8 if cfg["kind"] == "lev_utf8" else sw.LevenshteinDistances
            engine = cls(capabilities=scope, algorithm=args.algorithm)
            call = lambda: engine.pairs(da, db, scope, bound=cfg.get("bound"), out=out)
        call()
        # the device takes a few hundred milliseconds of work to reach its clocks: warm up by the clock, not by a count
        warm_until = time.perf_counter() + args.warm_seconds
        while time.perf_counter() < warm_until:
            call()
        scope.set_profiling(True)
        walls, timings = [], []
        for _ in range(args.repeats):
            t0 = time.perf_counter()
            call()
            walls.append(time.perf_counter() - t0)
            timings.append(scope.last_timing())
        scope.set_profiling(False)
        cells = timings[-1]["cells"]
        wall, comp = min(walls), min(t["compute_ms"] for t in timings) * 1e-3
        print(json.dumps({
            "config": name, **{k: v for k, v in cfg.items() if k != "gaps"}, "prepared": args.prepared, "algorithm": args.algorithm, "offsets": args.offsets,
            "pairs": pairs, "cells": cells,
            "gcups_call": round(cells / wall / 1e9, 1), "gcups_kernels": round(cells / comp / 1e9, 1),
            "call_ms": round(wall * 1e3, 3), "compute_ms": round(comp * 1e3, 3),
            "all_kernels_ms": round(min(t["total_ms"] for t in timings), 3), "kernels": timings[-1]["kernels"],
            "dominant": timings[-1]["dominant_name"], "generate_s": round(gen_s, 2),
        }), flush=True)
        da.free(); db.free()
        del da, db
        N.lib.swh_device_free(scope.handle, out_ptr)


if __name__ == "__main__":
    main()
