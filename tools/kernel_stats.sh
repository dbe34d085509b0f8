#!/bin/bash
# usage: tools/kernel_stats.sh <outdir> <python script + args...>   -- rocprofv3 kernel-trace stats of one command
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ks -- python3 "$@" > "$OUT/ks.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1]+"/ks_kernel_stats.csv")):
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
PY
