#!/bin/bash
# FETCH_SIZE and time per launch of k_bitparallel_tiled against the tile size (STRINGWARS_AMD_TILE): how much of C2's fetch
# traffic is tiles pushing each other out of the XCD's L2. Writes gpurun_out/tile_fetch/summary.txt.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/tile_fetch
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export STRINGWARS_AMD_LIBRARY=$REPO/stringwars_amd/libstringwars_amd_test.so   # (A / B switches are test hooks since round 6)
for tile in 0 256 512; do
    if [ "$tile" = 0 ]; then unset STRINGWARS_AMD_TILE; else export STRINGWARS_AMD_TILE=$tile; fi
    timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/t$tile" -o fetch -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 3 --warmup 1 --prewarm-seconds 0 > "$OUT/t$tile.log" 2>&1
    timeout 200 python3 "$REPO/bench.py" --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/t$tile.json"
    python3 - "$OUT" "$tile" <<'PY' >> "$OUT/summary.txt"
import csv, glob, json, sys
out, tile = sys.argv[1], sys.argv[2]
vals = [float(r["Counter_Value"]) for f in glob.glob(f"{out}/t{tile}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
        if "k_bitparallel_tiled" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
line = json.loads(open(f"{out}/t{tile}.json").read() or "{}")
print(f"tile {tile or 'default'}: FETCH_SIZE {sum(vals) / max(len(vals), 1) / 1024:.1f} MB raw per launch ({len(vals)} launches) | pipelined {line.get('value')} GCUPS, sync {line.get('value_sync_call')} GCUPS")
PY
done
rm -rf "$OUT"/t*/
cat "$OUT/summary.txt"
