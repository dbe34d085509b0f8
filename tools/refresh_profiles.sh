#!/bin/bash
# Regenerates the measurement files kept under profiles/<round>/ on a GPU box (writes into gpurun_out/refresh/).
# usage (from the repo root on the box): tools/refresh_profiles.sh
# Every profiler run is bounded by `timeout`; PMC passes use --kernel-trace only (pool rule).
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/refresh
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
# 1. the bench lines (C2 headline; C5 at 20 M pairs, the single-GPU slice of 100 M / 8 rounded up, and at the full 100 M)
timeout 300 python3 bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
timeout 300 python3 bench.py --config c5 --pairs 20000000 > "$OUT/bench_c5_20m.json" 2> "$OUT/bench_c5_20m.err"
timeout 400 python3 bench.py --config c5 --steps 20 --warmup 2 --no-cpu-baseline > "$OUT/bench_c5_100m.json" 2> "$OUT/bench_c5_100m.err"
# 2. per-config tables: raw tapes (hint route), prepared tapes
timeout 300 python3 tools/bench_configs.py --configs c1,c2,c3,c3u,c3b,c4,c4a,c4b,c4l,c5 --repeats 9 > "$OUT/configs_table.jsonl" 2> "$OUT/configs.err"
timeout 300 python3 tools/bench_configs.py --configs c1,c2,c3,c3u,c3b,c4,c4a,c4b,c4l,c5 --repeats 9 --prepared --offsets u32 >> "$OUT/configs_table.jsonl" 2>> "$OUT/configs.err"
timeout 300 python3 tools/bench_configs.py --configs c2 --repeats 9 --prepared --offsets u32 --algorithm bitparallel >> "$OUT/configs_table.jsonl" 2>> "$OUT/configs.err"
timeout 300 python3 tools/bench_cross.py > "$OUT/crossproduct_table.jsonl" 2> "$OUT/cross.err"
# 3. rocprofv3 --kernel-trace --stats of the bench command and of the other configs
cd /tmp && export TMPDIR=/tmp
stats() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rp_$name" -o "$name" -- python3 "$@" > "$OUT/rp_$name.log" 2>&1; cp "$OUT/rp_$name/${name}_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/rp_$name"; }
stats bench_c2 "$REPO/bench.py" --no-cpu-baseline
stats bench_c5 "$REPO/bench.py" --config c5 --pairs 20000000 --no-cpu-baseline
stats config_c3 "$REPO/tools/bench_configs.py" --configs c3 --repeats 10 --prepared
stats config_c4_linear "$REPO/tools/bench_configs.py" --configs c4 --repeats 5
stats config_c4_affine "$REPO/tools/bench_configs.py" --configs c4a --repeats 5
stats config_c4_256class "$REPO/tools/bench_configs.py" --configs c4b --repeats 5
# 4. PMC passes (SQ_*, FETCH_SIZE, WRITE_SIZE in runs of their own) -> pmc_constants.json
cd "$REPO"
pmc() { name=$1; shift; mkdir -p "$OUT/pmc_$name"; ( cd /tmp; for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do tag=$(echo $set | cut -d' ' -f1); timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o "$tag" -- python3 "$@" > "$OUT/pmc_$name/$tag.log" 2>&1; done ); }
pmc c2 "$REPO/bench.py" --no-cpu-baseline --steps 3 --warmup 1 --prewarm-seconds 0
python3 tools/pmc_constants.py "$OUT/pmc_c2" --workload tokens64 --pairs 1000000 --out "$OUT/pmc_constants.json" --source "rocprofv3 --pmc passes over 'bench.py --no-cpu-baseline --steps 3 --warmup 1 --prewarm-seconds 0' (tools/refresh_profiles.sh)"
pmc c2_planned "$REPO/bench.py" --no-cpu-baseline --steps 3 --warmup 1 --prewarm-seconds 0 --algorithm bitparallel
python3 tools/pmc_constants.py "$OUT/pmc_c2_planned" --workload tokens64 --pairs 1000000 --out "$OUT/pmc_constants.json" --source "the same with --algorithm bitparallel"
pmc c5 "$REPO/bench.py" --config c5 --pairs 20000000 --chunks 1 --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline
python3 tools/pmc_constants.py "$OUT/pmc_c5" --workload short_words --pairs 20000000 --out "$OUT/pmc_constants.json" --source "rocprofv3 --pmc passes over 'bench.py --config c5 --pairs 20000000 --chunks 1 --steps 3 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline'"
pmc c4_linear "$REPO/tools/bench_configs.py" --configs c4 --repeats 2
pmc c4_affine "$REPO/tools/bench_configs.py" --configs c4a --repeats 2
pmc c4_256class "$REPO/tools/bench_configs.py" --configs c4b --repeats 2
pmc c3 "$REPO/tools/bench_configs.py" --configs c3 --repeats 3 --prepared
python3 - "$OUT" <<'PY'
# per-kernel averages of the C3 / C4 PMC passes -> one JSON per config
import collections, csv, glob, json, os, sys
out = sys.argv[1]
for name in ("c4_linear", "c4_affine", "c4_256class", "c3"):
    sums = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in glob.glob(os.path.join(out, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "swh::" not in row["Kernel_Name"]: continue
            sums[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"]); disp[row["Kernel_Name"]][row["Counter_Name"]].add(row["Dispatch_Id"])
    book = {k[:160]: {c: round(v / len(disp[k][c]), 1) for c, v in cs.items()} | {"dispatches": max(len(d) for d in disp[k].values())} for k, cs in sums.items()}
    json.dump(book, open(os.path.join(out, f"config_{name}_pmc.json"), "w"), indent=1, sort_keys=True)
PY
find "$OUT" -name "*_agent_info.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
for d in "$OUT"/pmc_*; do rm -rf "$d"/*/ 2>/dev/null; done
du -sh "$OUT"; ls "$OUT"
