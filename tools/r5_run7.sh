set -x
mkdir -p gpurun_out/r5g
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "string_by_string or string_too_long or utf8_lines_are_staged" > gpurun_out/r5g/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5g/tests.log
tail -4 gpurun_out/r5g/tests.log
for leg in c3_raw c3_raw_cold; do
  timeout 300 python bench.py --only-config $leg > gpurun_out/r5g/bench_${leg}.json 2> gpurun_out/r5g/bench_${leg}.err
done
bash tools/profile_pmc_cmd.sh $PWD/gpurun_out/r5g/pmc_c3raw $PWD/bench.py --only-config c3_raw --calls 3 --no-cpu-baseline > gpurun_out/r5g/pmc_c3raw.txt 2>&1
