set -x
mkdir -p gpurun_out/r5f
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "sketch or prepared or config3" > gpurun_out/r5f/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5f/tests.log
tail -8 gpurun_out/r5f/tests.log
for leg in c3 c3_raw; do
  timeout 300 python bench.py --only-config $leg > gpurun_out/r5f/bench_${leg}.json 2> gpurun_out/r5f/bench_${leg}.err
done
STRINGWARS_AMD_SKETCH=0 timeout 300 python bench.py --only-config c3 > gpurun_out/r5f/bench_c3_nosketch.json 2>/dev/null
STRINGWARS_AMD_SKETCH=2 timeout 300 python bench.py --only-config c3_raw > gpurun_out/r5f/bench_c3_raw_sketch2.json 2>/dev/null
STRINGWARS_AMD_STAMPS=1 timeout 300 python bench.py --only-config c3 --calls 3 > /dev/null 2> gpurun_out/r5f/stamps.txt
