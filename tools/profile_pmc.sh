#!/bin/bash
# PMC counter passes over bench.py (one rocprofv3 run per counter set; --kernel-trace only, as the pool requires).
# usage: tools/profile_pmc.sh <outdir> [bench args...]
set -u
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT" -o "$name" -- python3 "$REPO/bench.py" --no-cpu-baseline --steps 3 --warmup 1 ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1; }
run pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run pmc_sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run pmc_fetch FETCH_SIZE
run pmc_write WRITE_SIZE
ls "$OUT"
