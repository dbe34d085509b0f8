#!/bin/bash
# usage: tools/profile_pmc_sets.sh <outdir> "<set1 counters>" "<set2 counters>" ... -- <python script + args...>
# One rocprofv3 pass per counter set (--pmc with --kernel-trace only, as the pool requires); prints per-kernel averages.
set -u
OUT=$1; shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT" -o "set$i" -- python3 "$@" > "$OUT/set$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys,json
out={}
for f in sorted(glob.glob(sys.argv[1]+"/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if "swh::" in k: out.setdefault(k,{"dispatches":len(disp[k])}).update({c:round(val/len(disp[k]),1) for c,val in v.items()})
json.dump(out,open(sys.argv[1]+"/summary.json","w"),indent=1)
for k,v in out.items():
    if v.get("SQ_INSTS_VALU",0)>1e6 or v.get("SQ_WAVE_CYCLES",0)>1e7: print(k[:60], json.dumps(v))
PY
