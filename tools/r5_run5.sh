set -x
mkdir -p gpurun_out/r5e
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "utf8 or string_by_string or string_too_long or doubling or believed or golden" > gpurun_out/r5e/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5e/tests.log
tail -8 gpurun_out/r5e/tests.log
for leg in c3_raw c3_raw_cold utf8_unbounded_raw; do
  timeout 300 python bench.py --only-config $leg > gpurun_out/r5e/bench_${leg}.json 2> gpurun_out/r5e/bench_${leg}.err
done
STRINGWARS_AMD_STAMPS=1 timeout 300 python bench.py --only-config c3_raw --calls 3 > /dev/null 2> gpurun_out/r5e/stamps.txt
bash tools/profile_pmc_cmd.sh $PWD/gpurun_out/r5e/pmc_c3raw $PWD/bench.py --only-config c3_raw --calls 3 --no-cpu-baseline > gpurun_out/r5e/pmc_c3raw.txt 2>&1
