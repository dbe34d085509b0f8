#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> -- registers / LDS / scratch of every kernel in one translation unit
# (compiles the device side only; no GPU needed)
set -e
SRC=$1
DIR=$(cd "$(dirname "$SRC")" && pwd)
OUT=/tmp/$(basename "$SRC" .hip).gfx950.co
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only --no-gpu-bundle-output -c "$SRC" -o "$OUT" -I"$DIR" ${EXTRA:-}
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT" | python3 -c '
import sys,re
name=None; rec={}
for line in sys.stdin:
    m=re.search(r"\.(name|vgpr_count|agpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size|vgpr_spill_count):\s+(\S+)", line)
    if not m: continue
    k,v=m.groups()
    if k=="name":
        if rec.get("name","").endswith(".kd"): pass
        rec["name"]=v
    else: rec[k]=v
    if k=="vgpr_spill_count":
        print("%-90s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill %s" % (rec.get("name","?")[:90], rec.get("vgpr_count"), rec.get("agpr_count","0"), rec.get("sgpr_count"), rec.get("group_segment_fixed_size"), rec.get("private_segment_fixed_size"), v)); rec={}
'
