#!/bin/bash
# usage: tools/profile_pmc_cmd.sh <outdir> <python script + args...>   (PMC passes, --kernel-trace only)
set -u
OUT=$1; shift
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; local counters=("$@"); timeout 900 rocprofv3 --pmc "${counters[@]}" --kernel-trace --output-format csv -d "$OUT" -o "$name" -- python3 $CMD > "$OUT/$name.log" 2>&1; }
CMD="$*"
run pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run pmc_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv,glob,collections,sys,json
out={}
for f in sorted(glob.glob(sys.argv[1]+"/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        if "swh::" in k: out.setdefault(k,{"dispatches":len(disp[k])}).update({c:round(val/len(disp[k]),1) for c,val in v.items()})
for k,v in out.items():
    if "GRBM_GUI_ACTIVE" in v and v.get("SQ_INSTS_VALU",0)>1e6:
        cyc=v["GRBM_GUI_ACTIVE"]/8
        print(k[:70], "| cycles",round(cyc),"| VALU insts",v["SQ_INSTS_VALU"],"| valu busy",round(v.get("SQ_ACTIVE_INST_VALU",0)/(cyc/4*1024),3),
              "| wait_inst/wave_cyc", round(v.get("SQ_WAIT_INST_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),3), "| wait_any/wave_cyc", round(v.get("SQ_WAIT_ANY",0)/max(v.get("SQ_WAVE_CYCLES",1),1),3),
              "| waves", v.get("SQ_WAVES"), "| wave_cyc/(cyc/4*4096slots)", round(v.get("SQ_WAVE_CYCLES",0)/(cyc/4*1024*8),3))
json.dump(out,open(sys.argv[1]+"/summary.json","w"),indent=1)
PY
