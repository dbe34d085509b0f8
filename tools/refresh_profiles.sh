#!/bin/bash
# Regenerates the measurement files kept under profiles/<round>/ on a GPU box (writes into gpurun_out/refresh/).
# usage (from the repo root on the box): tools/refresh_profiles.sh [legs...]      (default: every leg of bench.py)
# Every profiler run is bounded by `timeout`; PMC passes use --kernel-trace only (pool rule), one pass per counter set.
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/refresh
mkdir -p "$OUT"
cd "$REPO"
# entries of legs that are not re-profiled in this run stay as committed (a fresh box has no gpurun_out/)
[ -f "$OUT/pmc_constants.json" ] || cp profiles/r6/pmc_constants.json "$OUT/pmc_constants.json" 2>/dev/null || cp profiles/r5/pmc_constants.json "$OUT/pmc_constants.json" 2>/dev/null
LEGS=${*:-c2 c1 c3 c3_raw c3_raw_cold c3_raw_forget utf8_unbounded_raw utf8_unrelated_raw c3_k100 c4_linear c4_affine c4_bytes c4_letters52 c5 nw_words sw_linear sw_affine cross_lev cross_nw}
declare -A WORKLOAD=([c1]=words16 [c2]=tokens64 [c3]=utf8_lines [c3_raw]=utf8_lines [c3_raw_cold]=utf8_lines [c3_raw_forget]=utf8_lines [utf8_unbounded_raw]=utf8_lines [utf8_unrelated_raw]=script_lines [c3_k100]=utf8_lines [nw_words]=words16 [c4_linear]=protein4k [c4_affine]=protein4k [c4_bytes]=bytes4k [c4_letters52]=bytes4k [c5]=short_words [sw_linear]=protein4k [sw_affine]=protein4k [cross_lev]=acgt100 [cross_nw]=acgt100)
declare -A PAIRS=([c1]=10000 [c2]=1000000 [c3]=100000 [c3_raw]=100000 [c3_raw_cold]=100000 [c3_raw_forget]=100000 [utf8_unbounded_raw]=100000 [utf8_unrelated_raw]=50000 [c3_k100]=100000 [nw_words]=4000000 [c4_linear]=10000 [c4_affine]=10000 [c4_bytes]=10000 [c4_letters52]=10000 [c5]=20000000 [sw_linear]=10000 [sw_affine]=10000 [cross_lev]=4194304 [cross_nw]=4194304)
declare -A VARIANT=([c1]="" [c2]="" [c3]=k32 [c3_raw]=k32 [c3_raw_cold]=k32 [c3_raw_forget]=k32 [utf8_unbounded_raw]=unbounded [utf8_unrelated_raw]=unbounded [c3_k100]=k100 [nw_words]=unary_linear [c4_linear]=linear [c4_affine]=affine [c4_bytes]=linear [c4_letters52]=letters52 [c5]="" [sw_linear]=sw_linear [sw_affine]=sw_affine [cross_lev]=cross [cross_nw]=cross_linear)
CALLS=3
declare -A LEG_CALLS=([utf8_unbounded_raw]=9 [utf8_unrelated_raw]=9 [c3_k100]=9)   # two-stage calls: the first call of a scope runs in one stage (no lengths, no record yet) -- diluted
# 1. PMC passes per config: exactly $CALLS engine calls each -> per-call totals in pmc_constants.json (stamped with a digest
#    of the kernel's sources). c3_raw shares c3's dominant kernel and key; its pass is kept as a JSON summary only.
cd /tmp && export TMPDIR=/tmp
for leg in $LEGS; do
  CALLS=${LEG_CALLS[$leg]:-3}
  dir="$OUT/pmc_$leg"; rm -rf "$dir"; mkdir -p "$dir"
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$dir" -o "$tag" -- python3 "$REPO/bench.py" --only-config $leg --calls $CALLS --no-cpu-baseline > "$dir/$tag.log" 2>&1
  done
  if [ "$leg" != c3_raw ] && [ "$leg" != c3_raw_cold ] && [ "$leg" != c3_raw_forget ]; then   # (they share c3's dominant kernel and key: their passes are kept as JSON summaries only)
    python3 "$REPO/tools/pmc_constants.py" "$dir" --workload ${WORKLOAD[$leg]} --pairs ${PAIRS[$leg]} --calls $CALLS --variant "${VARIANT[$leg]}" --out "$OUT/pmc_constants.json" \
      --source "rocprofv3 --pmc passes over 'bench.py --only-config $leg --calls $CALLS --no-cpu-baseline' (tools/refresh_profiles.sh)"
  fi
  python3 - "$dir" "$OUT/config_${leg}_pmc.json" $CALLS <<'PY'
# per-kernel, per-call totals of every swh:: kernel of the config's passes -> one JSON per config
import collections, csv, glob, json, os, sys
sums = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "swh::" not in row["Kernel_Name"]: continue
        sums[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"]); disp[row["Kernel_Name"] + "|" + row["Counter_Name"]].add(row["Dispatch_Id"])
calls = int(sys.argv[3])
book = {k[:160]: {c: round(v / calls, 1) for c, v in cs.items()} | {"dispatches_per_call": round(max(len(disp[k + "|" + c]) for c in cs) / calls, 2)} for k, cs in sums.items()}
json.dump({"per": f"engine call ({calls} calls profiled); FETCH_SIZE / WRITE_SIZE in KB, uncorrected", "kernels": book}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
  find "$dir" -name "*_agent_info.csv" -delete; find "$dir" -name "*kernel_trace.csv" -delete; find "$dir" -name "*counter_collection.csv" -delete
done
if [ "${REFRESH_PMC_ONLY:-0}" = 1 ]; then rm -rf "$OUT"/pmc_*/; ls "$OUT"; exit 0; fi
# 2. rocprofv3 --kernel-trace --stats of the bench command (synchronous steps only: what `roofline.kernel_ms` averages over) and of every leg
stats() { name=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rp_$name" -o "$name" -- python3 "$@" > "$OUT/rp_$name.log" 2>&1; cp "$OUT/rp_$name/${name}_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/rp_$name"; }
stats bench_c2_sync "$REPO/bench.py" --no-cpu-baseline --no-configs --no-pipelined
stats bench_c2_full "$REPO/bench.py" --no-cpu-baseline --no-configs
for leg in $LEGS; do stats leg_$leg "$REPO/bench.py" --only-config $leg --no-cpu-baseline; done
# 3. the bench lines themselves, with the fresh constants in place
cd "$REPO"
mkdir -p profiles/r6 && cp "$OUT/pmc_constants.json" profiles/r6/pmc_constants.json
timeout -k 10 900 python3 bench.py --details-out "$OUT/bench_configs.json" > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
timeout -k 10 400 python3 bench.py --config c5 --steps 20 --warmup 2 --no-cpu-baseline --no-configs > "$OUT/bench_c5_100m.json" 2> "$OUT/bench_c5_100m.err"
timeout -k 10 300 python3 tools/bench_cross.py > "$OUT/crossproduct_table.jsonl" 2> "$OUT/cross.err"
timeout -k 10 300 python3 tools/bench_bounds.py > "$OUT/bounds_table.jsonl" 2> "$OUT/bounds.err"
timeout -k 10 300 python3 tools/bench_bounds.py --bytes >> "$OUT/bounds_table.jsonl" 2>> "$OUT/bounds.err"
timeout -k 10 600 python3 tools/bench_sizes.py > "$OUT/c2_sizes.jsonl" 2> "$OUT/c2_sizes.err"
rm -rf "$OUT"/pmc_*/
du -sh "$OUT"; ls "$OUT"
