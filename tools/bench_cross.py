#!/usr/bin/env python3
"""Cross-product rows in the shape of the reference's published tables (similarities/README.md:30-136): queries [0, side)
x candidates [side, 2*side), side = round(sqrt(BATCH_PER_CORE * compute units)) (bench.rs:113-117), synthetic stand-ins
for its datasets (ACGT 100 B / 1 KB, ~5 B words, ~3.2 KB lines; `ulines`: lines of one non-Latin or Latin script each, ~1000 code points). Matrix stays on the device (UnifiedMat role)."""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stringwars_amd as sw
from stringwars_amd import _native as N


def tokens(kind, count, rng):
    if kind == "acgt100":
        lens = np.full(count, 100)
    elif kind == "acgt1k":
        lens = np.full(count, 1000)
    elif kind.startswith("acgt"):   # acgt<N>: any other fixed length, for route experiments
        lens = np.full(count, int(kind[4:]))
    elif kind.startswith("text"):   # text<N>: 26 letters, lengths around N (the mid-length shapes between words and lines)
        lens = np.clip(rng.normal(int(kind[4:]), int(kind[4:]) / 5, count).astype(int), 1, None)
    elif kind in ("words", "uwords", "twords"):
        lens = np.clip(rng.poisson(4.0, count) + 1, 1, 24)
    elif kind == "lines":
        lens = np.clip(rng.normal(3200, 1200, count).astype(int), 200, 9000)
    elif kind == "ulines":   # article lines of 700 ... 1300 code points, one script each (~1.8 KB): XLSum lines as the UTF-8 engine sees them
        return sw.generate_pairs("script_lines", count, seed=42)[0]
    if kind in ("uwords", "twords"):   # multilingual word-sized tokens (what XLSum words look like to the UTF-8 engine): ~5 code points of four scripts
        cps = np.array([0x61, 0x65, 0x6F, 0x74, 0xE9, 0xFC, 0x430, 0x435, 0x43E, 0x442, 0x4E2D, 0x6587, 0x65E5, 0x672C], dtype=np.uint32)
        lens = np.clip(rng.poisson(4.0, count) + 1, 1, 24)
        if kind == "uwords":
            lens = np.minimum(lens, 12)            # (~11 bytes on average, no token beyond 36: the lane kernels' own range)
        else:
            lens[rng.integers(0, count, 6)] = 60   # "twords": six tokens of ~130 bytes among them (URLs, unsegmented sentences)
        return sw.Strs(["".join(chr(int(c)) for c in cps[rng.integers(0, len(cps), int(n))]).encode() for n in lens])
    alphabet = np.frombuffer(b"ACGT" if kind.startswith("acgt") else bytes(range(97, 123)), dtype=np.uint8)
    offsets = np.zeros(count + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    data = alphabet[rng.integers(0, len(alphabet), int(offsets[-1]))]
    return sw.Strs(data=data, offsets=offsets)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="acgt100,acgt1k,words,uwords,lines,ulines")
    ap.add_argument("--batch-per-core", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args()
    scope = sw.DeviceScope(gpu_device=0)
    rng = np.random.default_rng(42)
    classes, costs = sw.unary_class_costs(2, -1)
    for kind in args.kinds.split(","):
        per_core = args.batch_per_core or (256 if kind in ("lines", "ulines") else 16384)   # similarities/README.md:22-23
        side = max(1, round((per_core * scope.compute_units) ** 0.5))
        tape = tokens(kind, 2 * side, rng)
        q, c = tape.subview(0, side).to_device(scope), tape.subview(side, 2 * side).to_device(scope)
        lens = tape.lengths
        cells = int(lens[:side].sum()) * int(lens[side:].sum())
        # the UTF-8 engine's cells are counted in code points (bench.rs:230-247): lead bytes per string
        leads = np.concatenate([[0], np.cumsum((tape.data & 0xC0) != 0x80)])
        cp_lens = leads[tape.offsets[1:].astype(np.int64)] - leads[tape.offsets[:-1].astype(np.int64)]
        cp_cells = int(cp_lens[:side].sum()) * int(cp_lens[side:].sum())
        out_ptr, err = C.c_void_p(), C.c_char_p()
        N.check(N.lib.swh_device_alloc(scope.handle, side * side * 8, C.byref(out_ptr), C.byref(err)), err)
        engines = {
            "uniform/LevenshteinDistances": sw.LevenshteinDistances(capabilities=scope),
            "uniform/LevenshteinDistancesUtf8": sw.LevenshteinDistancesUTF8(capabilities=scope),
            "linear/NeedlemanWunschScores": sw.NeedlemanWunschScores(classes, costs, open=-2, extend=-2, capabilities=scope),
            "affine/NeedlemanWunschScores": sw.NeedlemanWunschScores(classes, costs, open=-5, extend=-1, capabilities=scope),
            "linear/SmithWatermanScores": sw.SmithWatermanScores(classes, costs, open=-2, extend=-2, capabilities=scope),
            "affine/SmithWatermanScores": sw.SmithWatermanScores(classes, costs, open=-5, extend=-1, capabilities=scope),
        }
        for name, engine in engines.items():
            call = lambda: engine(q, c, scope, out=int(out_ptr.value))
            call()
            warm_until = time.perf_counter() + 0.3   # the device needs a few hundred milliseconds of work to reach its clocks
            while time.perf_counter() < warm_until:
                call()
            best = 1e9
            for _ in range(args.repeats):
                t0 = time.perf_counter(); call(); best = min(best, time.perf_counter() - t0)
            row_cells = cp_cells if "Utf8" in name else cells
            print(json.dumps({"dataset": kind, "side": side, "row": name + "<1gpu>", "mcups": round(row_cells / best / 1e6),
                              "call_ms": round(best * 1e3, 3), "cells": row_cells}), flush=True)
        N.lib.swh_device_free(scope.handle, out_ptr)
        q.free(); c.free()


if __name__ == "__main__":
    main()
