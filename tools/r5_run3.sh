set -x
mkdir -p gpurun_out/r5c
timeout 2400 python -m pytest tests/ -q -m gpu --deselect tests/test_gpu_parity.py::test_bench_line_carries_every_config > gpurun_out/r5c/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c/tests.log
tail -15 gpurun_out/r5c/tests.log
for leg in utf8_unbounded_raw c3_k100 c3 c3_raw; do
  timeout 300 python bench.py --only-config $leg > gpurun_out/r5c/bench_${leg}.json 2> gpurun_out/r5c/bench_${leg}.err
done
STRINGWARS_AMD_DOUBLING=0 timeout 300 python bench.py --only-config utf8_unbounded_raw > gpurun_out/r5c/bench_utf8_unbounded_raw_nodoubling.json 2>/dev/null
timeout 300 python tools/bench_bounds.py > gpurun_out/r5c/bounds_table.jsonl 2> gpurun_out/r5c/bounds.err
timeout 300 python tools/bench_bounds.py --bytes >> gpurun_out/r5c/bounds_table.jsonl 2>> gpurun_out/r5c/bounds.err
