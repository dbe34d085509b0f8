#!/usr/bin/env python3
"""Measuring tool: UNRELATED lines through the UTF-8 engine, unbounded -- what the reference's cross-product of article lines is made
of (similarities/README.md:39-40, :56: XLSum lines, one language per line), and what the two-stage schedule cannot settle.
Lines of ~1000 code points, each in ONE script (its letters + ASCII punctuation / digits / spaces), pair i = (line i, line i + 1):
    python tools/bench_unrelated.py [--pairs 50000] [--scripts cyrillic,latin,...] [--cps 700,1300] [--mixed]
--mixed: C3's synthetic lines instead (four scripts in every line, ~340 distinct symbols: beyond a 255-slot dictionary).
Rows: code points on prepared tapes, the same on raw tapes, and the same tapes as bytes. STRINGWARS_AMD_DOUBLING=0 is set: the first
stage would be tried once and then sit out (nothing to settle); the rows are the block kernels' own."""
import argparse, json, os, sys, time
os.environ.setdefault("STRINGWARS_AMD_DOUBLING", "0")
# (A / B switches are test hooks since round 6: the TEST library reads them, the shipped one does not)
os.environ.setdefault("STRINGWARS_AMD_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stringwars_amd", "libstringwars_amd_test.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stringwars_amd as sw

SCRIPTS = {
    "latin": list(range(0x61, 0x7B)) + list(range(0x41, 0x5B)) + [0xE9, 0xE8, 0xE0, 0xF1, 0xFC, 0xF6, 0xE4, 0xDF],
    "cyrillic": list(range(0x410, 0x450)) + [0x401, 0x451],
    "greek": list(range(0x391, 0x3AA)) + list(range(0x3B1, 0x3CA)),
    "arabic": list(range(0x621, 0x64B)),
    "devanagari": list(range(0x905, 0x93A)) + list(range(0x93E, 0x94E)),
    "kana": list(range(0x3041, 0x3097)) + list(range(0x30A1, 0x30FB)),
    "cjk": list(range(0x4E00, 0x4E00 + 3000)),
}
COMMON = [0x20] * 12 + list(b".,;:!?-()\"'") + list(range(0x30, 0x3A))


def lines(count, scripts, rng, lo=700, hi=1300):
    out = []
    for i in range(count):
        letters = SCRIPTS[scripts[i % len(scripts)]]
        n = int(rng.integers(lo, hi))
        pick = np.where(rng.random(n) < 0.8, rng.choice(letters, n), rng.choice(COMMON, n))
        out.append("".join(map(chr, pick.tolist())).encode())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=50_000)
    ap.add_argument("--scripts", default="latin,cyrillic,greek,arabic,devanagari")
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--cps", default="700,1300", help="code points per line, lo,hi (beyond 2048: the kernel of several passes)")
    args = ap.parse_args()
    import torch
    scope = sw.DeviceScope(gpu_device=0)
    rng = np.random.default_rng(42)
    if args.mixed:
        a, _ = sw.generate_pairs("utf8_lines", args.pairs + 1, seed=42)
        label = "C3's synthetic lines (four scripts per line)"
    else:
        lo, hi = (int(x) for x in args.cps.split(","))
        a = sw.Strs(lines(args.pairs + 1, args.scripts.split(","), rng, lo, hi))
        label = f"{args.scripts} {lo}-{hi} code points"
    left, right = a.subview(0, args.pairs), a.subview(1, args.pairs + 1)
    leads = np.concatenate([[0], np.cumsum((a.data & 0xC0) != 0x80)])
    cps = leads[a.offsets[1:].astype(np.int64)] - leads[a.offsets[:-1].astype(np.int64)]
    cp_cells = int((cps[:-1].astype(np.int64) * cps[1:]).sum())
    byte_cells = int((a.lengths[:-1].astype(np.int64) * a.lengths[1:]).sum())
    out = torch.zeros(args.pairs, dtype=torch.int32, device="cuda")
    rows = (("code points, prepared", sw.LevenshteinDistancesUTF8, lambda t: sw.PreparedTape(scope, t, utf8=True), cp_cells),
            ("code points, raw device tapes", sw.LevenshteinDistancesUTF8, lambda t: t.to_device(scope), cp_cells),
            ("bytes, prepared", sw.LevenshteinDistances, lambda t: sw.PreparedTape(scope, t, utf8=False), byte_cells))
    for name, Engine, stage, cells in rows:
        engine = Engine(capabilities=scope)
        ta, tb = stage(left), stage(right)
        call = engine.bind_pairs(ta, tb, scope, out) if isinstance(ta, sw.PreparedTape) else (lambda: engine.pairs(ta, tb, scope, out=out))
        call()
        until = time.perf_counter() + 0.5
        while time.perf_counter() < until:
            call()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
        scope.set_profiling(True); call(); timing = scope.last_timing(); scope.set_profiling(False)
        print(json.dumps({"lines": label, "pairs": args.pairs, "row": name, "tcups": round(cells / best / 1e12, 2), "call_ms": round(best * 1e3, 3),
                          "kernel": timing["dominant_name"], "kernel_ms": round(timing["compute_ms"], 3), "cells": cells,
                          "library": os.path.basename(os.environ.get("STRINGWARS_AMD_LIBRARY", "libstringwars_amd.so")),
                          "dense": os.environ.get("STRINGWARS_AMD_BP_DENSE", "1")}), flush=True)


if __name__ == "__main__":
    main()
