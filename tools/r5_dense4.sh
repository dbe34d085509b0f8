set -x
mkdir -p gpurun_out/dense4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense_alphabet or longer_than_64_blocks or golden_script_lines" > gpurun_out/dense4/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/dense4/tests.log
tail -25 gpurun_out/dense4/tests.log
export STRINGWARS_AMD_LIBRARY=$PWD/stringwars_amd/libstringwars_amd_test.so
for dense in 1 0; do
  STRINGWARS_AMD_BP_DENSE=$dense timeout 600 python tools/bench_unrelated.py --pairs 8000 --cps 2500,3600 >> gpurun_out/dense4/long.jsonl 2>> gpurun_out/dense4/long.err
done
cat gpurun_out/dense4/long.jsonl; tail -3 gpurun_out/dense4/long.err
