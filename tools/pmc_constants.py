#!/usr/bin/env python3
"""Folds rocprofv3 PMC passes (tools/refresh_profiles.sh) into profiles/<round>/pmc_constants.json: per kernel family,
workload and variant, the PER-CALL totals bench.py's roofline objects need (SQ_INSTS_VALU for executed lane-ops,
FETCH_SIZE / WRITE_SIZE for HBM traffic) plus the counters DESIGN.md quotes.

    tools/pmc_constants.py <pmc_outdir> --workload tokens64 --pairs 1000000 --calls 4 [--variant linear] [--out ...]

`--calls` is the number of engine calls the profiled command made (`bench.py --only-config NAME --calls N` makes exactly
N): a call may launch several kernels of one family (the wavefront classes of an NW call), so the figures are sums over
every dispatch of the family divided by the calls, not per-dispatch averages. Every entry is stamped with a digest of
the kernel's sources (tools/kernel_sources.py); bench.py flags entries whose digest no longer matches the tree.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_sources import KERNELS, ROOT, source_digest  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--pairs", type=int, required=True)
    ap.add_argument("--calls", type=int, required=True, help="engine calls made by the profiled command")
    ap.add_argument("--variant", default="", help="distinguishes entries of one kernel and workload (linear / affine / k32 / raw ...)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r6", "pmc_constants.json"))
    ap.add_argument("--source", default=None, help="what to record as the origin of the numbers")
    args = ap.parse_args()
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    dispatches = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in sorted(glob.glob(os.path.join(args.outdir, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(path)):
            sums[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
            dispatches[row["Kernel_Name"]][row["Counter_Name"]].add(row["Dispatch_Id"])
    try:
        book = json.load(open(args.out))
    except (OSError, ValueError):
        book = {"about": "per-call totals of rocprofv3 --pmc passes (tools/refresh_profiles.sh) over `bench.py --only-config NAME --calls N`: "
                         "sums over every dispatch of the kernel family / N; FETCH_SIZE / WRITE_SIZE in KB, uncorrected",
                "kernels": {}}
    for stamp, (needle, _) in KERNELS.items():
        merged, counts = collections.defaultdict(float), collections.defaultdict(int)
        for kernel, counters in sums.items():
            if needle not in kernel:
                continue
            for counter, total in counters.items():
                merged[counter] += total
                counts[counter] += len(dispatches[kernel][counter])
        if not merged or "SQ_INSTS_VALU" not in merged:
            continue
        per = {c: merged[c] / args.calls for c in merged}
        entry = {"kernel_symbol": needle, "calls": args.calls, "dispatches_per_call": round(counts["SQ_INSTS_VALU"] / args.calls, 2),
                 "pairs_per_call": args.pairs,
                 "valu_insts": round(per["SQ_INSTS_VALU"], 1), "valu_insts_per_pair": per["SQ_INSTS_VALU"] / args.pairs,
                 "fetch_kb": round(per["FETCH_SIZE"], 1) if "FETCH_SIZE" in per else None,
                 "write_kb": round(per["WRITE_SIZE"], 1) if "WRITE_SIZE" in per else None,
                 "counters": {c: round(v, 1) for c, v in sorted(per.items())},
                 "source_digest": source_digest(stamp),
                 "source": args.source or f"tools/refresh_profiles.sh -> {os.path.basename(os.path.normpath(args.outdir))}"}
        key = f"{stamp}|{args.workload}" + (f"|{args.variant}" if args.variant else "")
        book["kernels"][key] = entry
        print(key, args.pairs, "VALU wave-insts/call", entry["valu_insts"], "fetch KB", entry["fetch_kb"], "write KB", entry["write_kb"],
              "dispatches/call", entry["dispatches_per_call"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(book, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
