#!/usr/bin/env python3
"""Counterpart of the reference's ``similarities/bench.py`` for the rows this backend provides.

    STRINGWARS_DATASET=README.md python -m stringwars_amd.bench_similarities --time-limit 1 [-k regex]

Same flow (reference file:line): load + tokenize in ``words`` mode, seeded shuffle (bench.py:847-867), per-device
batch ``256 * compute units`` -> ``side = round(sqrt(batch))`` clamped to ``tokens / 2`` (bench.py:152-161,
:330-368), disjoint query/candidate slices, a preallocated ``side x side`` matrix reused by
``engine(queries, candidates, scope, out=matrix)`` (bench.py:399-422), ``category/library.Engine<1gpu>`` names,
``SKIPPED (<reason>)`` instead of a crash (bench.py:402-431). Adds the pairwise batch row, shaped like
``cudf ... str.edit_distance`` (bench.py:584-622).
"""
from __future__ import annotations

import argparse
import re
import sys

import numpy as np

import stringwars_amd as swa
from stringwars_amd import harness as H

DEFAULT_BATCH_PER_CORE = 256  # bench.py / bench.rs:93


def main(argv=None) -> int:
    parser = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    parser.add_argument("--dataset")
    parser.add_argument("--tokens", choices=["lines", "words", "file"], default=None)
    parser.add_argument("-k", "--filter", metavar="REGEX")
    parser.add_argument("--time-limit", type=float, default=10.0)
    parser.add_argument("--dataset-limit", type=str, default="128mb",
                        help="Maximum dataset size (default: 128mb). Supports formats like '1gb', '500mb', '10kb'")     # utils.py:489-494
    parser.add_argument("--bio", action="store_true",
                        help="accepted so that the reference's command lines work (similarities/bench.py:824-828 gates its NW / SW groups on it); "
                             "this backend's linear / affine rows are always measured")
    parser.add_argument("--batch-size", type=int, default=None, help="pairs per core (overrides STRINGWARS_BATCH_PER_CORE)")
    args = parser.parse_args(argv)
    pattern = re.compile(args.filter) if args.filter else None
    time_limit = H.get_env_parsed("STRINGWARS_TIME", args.time_limit, parser=float)
    warmup = H.get_env_parsed("STRINGWARS_WARMUP", 0.0, parser=float)

    try:
        tokens = H.load_tokens(args.dataset, tokens_mode=args.tokens or "words", size_limit=args.dataset_limit)
    except ValueError as problem:
        parser.error(str(problem))
    if len(tokens) < 2:
        parser.error("Dataset must contain at least two tokens for the cross-product")
    codepoints = np.fromiter((len(t) for t in tokens), dtype=np.int64, count=len(tokens))
    nbytes = np.fromiter((len(t.encode("utf-8")) for t in tokens), dtype=np.int64, count=len(tokens))
    print(f"stringwars_amd {swa.__version__} ({swa.capabilities()})")

    try:
        scope = swa.DeviceScope(gpu_device=0)
        scope_error = None
    except Exception as creation_error:  # the reference drops the <1gpu> variant when the scope cannot be built
        scope, scope_error = None, creation_error
    cores = scope.compute_units if scope is not None else 64
    side = H.crossproduct_side(H.auto_batch_size(cores, base=args.batch_size, default_base=DEFAULT_BATCH_PER_CORE), len(tokens))
    print(f"- GPU batch: {side}x{side} cross-product, tokens available: {len(tokens)}\n")
    queries, candidates = swa.Strs(tokens[:side]), swa.Strs(tokens[side:2 * side])

    def cross_row(category, label, make_engine, lengths, dtype):
        name = f"{label}<1gpu>"
        if not H.should_run(f"{category}/{name}", pattern):
            return
        if scope is None:
            print(f"{name}: SKIPPED ({scope_error})")
            return
        try:
            engine = make_engine()
            matrix = np.zeros((side, side), dtype=dtype)
            compute = lambda: engine(queries, candidates, scope, out=matrix)
            compute()
        except Exception as error:
            print(f"{name}: SKIPPED ({error})")
            return
        cells = int(lengths[:side].sum()) * int(lengths[side:2 * side].sum())       # bench.py:177-181
        total_bytes = int(nbytes[:side].sum()) + int(nbytes[side:2 * side].sum())
        H.measure(name, compute, cells, total_bytes, warmup, time_limit)

    print("# uniform")
    cross_row("uniform", "stringwars_amd.LevenshteinDistances", lambda: swa.LevenshteinDistances(capabilities=scope), nbytes, np.uint64)
    cross_row("uniform", "stringwars_amd.LevenshteinDistancesUTF8", lambda: swa.LevenshteinDistancesUTF8(capabilities=scope),
              codepoints, np.uint64)
    name = "stringwars_amd.edit_distance<1gpu>"
    if H.should_run(f"uniform/{name}", pattern):
        if scope is None:
            print(f"{name}: SKIPPED ({scope_error})")
        else:
            half = len(tokens) // 2
            col_a, col_b = swa.Strs(tokens[:half]), swa.Strs(tokens[half:2 * half])
            engine = swa.LevenshteinDistancesUTF8(capabilities=scope)
            out = np.zeros(half, dtype=np.uint32)
            cells = int((codepoints[:half] * codepoints[half:2 * half]).sum())      # bench.py:599
            total_bytes = int(nbytes[:2 * half].sum())
            H.measure(name, lambda: engine.pairs(col_a, col_b, scope, out=out), cells, total_bytes, warmup, time_limit)
            print(f"  {name} checksum={int(out.sum())}", file=sys.stderr)                  # bench.py:305

    # Bounded Levenshtein: out[i] = min(d, k + 1); k from STRINGWARS_ERROR_BOUND (the reference's only trace of
    # a bound, README.md:311; STRINGWARS_BOUND is an alias), default 32. Bytes and code points, every pair checked against
    # the unbounded distance of the same engine.
    bound = H.get_env_parsed("STRINGWARS_ERROR_BOUND", H.get_env_parsed("STRINGWARS_BOUND", 32))
    for label, cls, lengths in ((f"stringwars_amd.levenshtein_pairs<k={bound},1gpu>", swa.LevenshteinDistances, nbytes),
                                (f"stringwars_amd.edit_distance<k={bound},1gpu>", swa.LevenshteinDistancesUTF8, codepoints)):
        if not H.should_run(f"uniform/{label}", pattern):
            continue
        if scope is None:
            print(f"{label}: SKIPPED ({scope_error})")
            continue
        half = len(tokens) // 2
        col_a, col_b = swa.Strs(tokens[:half]), swa.Strs(tokens[half:2 * half])
        engine = cls(capabilities=scope)
        out = np.zeros(half, dtype=np.uint32)
        cells = int((lengths[:half] * lengths[half:2 * half]).sum())
        H.measure(label, lambda: engine.pairs(col_a, col_b, scope, bound=bound, out=out), cells, int(nbytes[:2 * half].sum()), warmup, time_limit)
        unbounded = engine.pairs(col_a, col_b, scope)
        if not (out == np.minimum(unbounded, bound + 1)).all():
            print(f"error: {label}: bounded != min(unbounded, k + 1)", file=sys.stderr)
            return 2
        print(f"  {label} exceeded={int((unbounded > bound).sum())} of {half}", file=sys.stderr)

    byte_to_class, class_costs = swa.unary_class_costs(2, -1)                          # bench.py:742
    for header, category, gap_open, gap_extend in (("# linear", "linear", -2, -2), ("# affine", "affine", -5, -1)):
        print(header)
        for label, cls in (("stringwars_amd.NeedlemanWunschScores", swa.NeedlemanWunschScores),
                           ("stringwars_amd.SmithWatermanScores", swa.SmithWatermanScores)):
            cross_row(category, label, lambda cls=cls: cls(byte_to_class, class_costs, open=gap_open, extend=gap_extend,
                                                           capabilities=scope), nbytes, np.int64)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
