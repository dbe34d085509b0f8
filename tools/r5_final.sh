set -x
mkdir -p gpurun_out/r5z
timeout 2700 python -m pytest tests/ -q -m gpu > gpurun_out/r5z/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5z/tests.log
tail -6 gpurun_out/r5z/tests.log
cp profiles/r5/pmc_constants.json gpurun_out/refresh/pmc_constants.json 2>/dev/null || (mkdir -p gpurun_out/refresh && cp profiles/r5/pmc_constants.json gpurun_out/refresh/pmc_constants.json)
REFRESH_PMC_ONLY=1 timeout 1500 bash tools/refresh_profiles.sh utf8_unbounded_raw c3_k100 c3_raw_cold > gpurun_out/r5z/refresh_pmc.log 2>&1
cp gpurun_out/refresh/pmc_constants.json profiles/r5/pmc_constants.json
timeout 900 python bench.py > gpurun_out/r5z/bench_c2.json 2> gpurun_out/r5z/bench_c2.err
timeout 600 python bench.py --config c5 --steps 20 --warmup 2 --no-cpu-baseline --no-configs > gpurun_out/r5z/bench_c5_100m.json 2> gpurun_out/r5z/bench_c5_100m.err
cp gpurun_out/refresh/pmc_constants.json gpurun_out/r5z/pmc_constants.json
cp gpurun_out/refresh/config_utf8_unbounded_raw_pmc.json gpurun_out/refresh/config_c3_k100_pmc.json gpurun_out/refresh/config_c3_raw_cold_pmc.json gpurun_out/r5z/
