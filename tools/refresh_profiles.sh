#!/bin/bash
# Regenerates the measurement files kept under profiles/<round>/ on a GPU box (writes into gpurun_out/refresh/).
# usage (from the repo root on the box): tools/refresh_profiles.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/refresh
mkdir -p "$OUT"
cd "$REPO"
python3 bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python3 tools/bench_configs.py --configs c1,c2,c3,c3u,c3b,c4,c4a,c4b,c4l,c5 --repeats 5 > "$OUT/configs_table.jsonl" 2> "$OUT/configs.err"
python3 tools/bench_cross.py > "$OUT/crossproduct_table.jsonl" 2> "$OUT/cross.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rp_bench" -o bench -- python3 "$REPO/bench.py" --no-cpu-baseline > "$OUT/bench_c2_under_rocprof.json" 2> "$OUT/rp_bench.err"
cp "$OUT/rp_bench/bench_kernel_stats.csv" "$OUT/bench_c2_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rp_c3" -o c3 -- python3 "$REPO/tools/bench_configs.py" --configs c3 --repeats 10 > /dev/null 2> "$OUT/rp_c3.err"
cp "$OUT/rp_c3/c3_kernel_stats.csv" "$OUT/config_c3_kernel_stats.csv"
cd "$REPO"
tools/profile_pmc.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
hipcc --offload-arch=gfx950 -O3 -w tools/valu_ops.hip -o /tmp/valu_ops && /tmp/valu_ops > "$OUT/valu_ops.txt"
hipcc --offload-arch=gfx950 -O3 -w tools/valu_chain.hip -o /tmp/valu_chain && /tmp/valu_chain > "$OUT/valu_chain.txt"
rm -rf "$OUT/rp_bench" "$OUT/rp_c3"
ls -la "$OUT"
