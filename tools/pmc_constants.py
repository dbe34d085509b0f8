#!/usr/bin/env python3
"""Folds the rocprofv3 PMC passes of tools/profile_pmc.sh into profiles/<round>/pmc_constants.json: per kernel and
workload, the per-dispatch averages bench.py's roofline object needs (SQ_INSTS_VALU for executed lane-ops,
FETCH_SIZE / WRITE_SIZE for HBM traffic) plus the counters DESIGN.md quotes.

    tools/pmc_constants.py <pmc_outdir> --workload tokens64 --pairs 1000000 [--out profiles/r2/pmc_constants.json]
"""
import argparse
import collections
import csv
import glob
import json
import os

# stamp name used by the library's timing (swh_timing_t::dominant_name) -> substring of the kernel symbol
KERNELS = {
    "bitparallel": "swh::k_bitparallel<unsigned char,",
    "bitparallel_u32": "swh::k_bitparallel<unsigned int,",
    "bitparallel_tiled": "swh::k_bitparallel_tiled<unsigned char,",
    "bitparallel_tiled_u32": "swh::k_bitparallel_tiled<unsigned int,",
    "bitparallel_long": "swh::k_bitparallel_long<unsigned char",
    "direct_short": "swh::k_direct_short<",
    "short_tiled": "swh::k_short_tiled<",
    "banded": "swh::k_banded<",
    "wavefront": "swh::k_wavefront<",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--pairs", type=int, required=True)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r2", "pmc_constants.json"))
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
        book = {"about": "per-dispatch averages of rocprofv3 --pmc passes (tools/profile_pmc.sh); FETCH_SIZE / WRITE_SIZE in KB, uncorrected",
                "kernels": {}}
    for stamp, needle in KERNELS.items():
        merged, counts = collections.defaultdict(float), collections.defaultdict(int)
        for kernel, counters in sums.items():
            if needle not in kernel:
                continue
            for counter, total in counters.items():
                merged[counter] += total
                counts[counter] += len(dispatches[kernel][counter])
        if not merged or "SQ_INSTS_VALU" not in merged:
            continue
        per = {c: merged[c] / counts[c] for c in merged}
        # Every dispatch of the profiled command scores `--pairs` pairs, so the averages are per launch of that size; the
        # per-pair figures let bench.py price launches of another size of the same workload (work is linear in pairs).
        entry = {"kernel_symbol": needle, "dispatches": int(counts["SQ_INSTS_VALU"]), "pairs_per_launch": args.pairs,
                 "valu_insts": round(per["SQ_INSTS_VALU"], 1), "valu_insts_per_pair": per["SQ_INSTS_VALU"] / args.pairs,
                 "fetch_kb": round(per["FETCH_SIZE"], 1) if "FETCH_SIZE" in per else None,
                 "write_kb": round(per["WRITE_SIZE"], 1) if "WRITE_SIZE" in per else None,
                 "counters": {c: round(v, 1) for c, v in sorted(per.items())},
                 "source": args.source or f"tools/profile_pmc.sh -> {os.path.basename(os.path.normpath(args.outdir))}"}
        book["kernels"][f"{stamp}|{args.workload}"] = entry
        print(stamp, args.workload, args.pairs, "VALU wave-insts", entry["valu_insts"], "fetch KB", entry["fetch_kb"], "write KB", entry["write_kb"])
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(book, open(args.out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
