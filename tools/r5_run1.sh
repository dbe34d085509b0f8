set -x
mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "utf8_lines_are_staged or string_too_long or string_by_string" > gpurun_out/r5a/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5a/tests.log
for mode in auto tiles; do
  for leg in c3_raw c3_raw_cold utf8_unbounded_raw; do
    STRINGWARS_AMD_UTF8_STAGING=$mode timeout 300 python bench.py --only-config $leg > gpurun_out/r5a/bench_${leg}_${mode}.json 2> gpurun_out/r5a/bench_${leg}_${mode}.err
  done
  STRINGWARS_AMD_UTF8_STAGING=$mode STRINGWARS_AMD_STAMPS=1 timeout 300 python bench.py --only-config c3_raw --calls 3 > /dev/null 2> gpurun_out/r5a/stamps_${mode}.txt
done
tail -5 gpurun_out/r5a/tests.log
