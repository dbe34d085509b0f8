set -x
mkdir -p gpurun_out/r5d
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide_class_table or doubling or bench_failures or bench_starts_two_ranks or class_table_needleman or column_profile or bench_line_carries" > gpurun_out/r5d/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5d/tests.log
tail -15 gpurun_out/r5d/tests.log
for leg in c4_letters52 c4_bytes c3_k100 utf8_unbounded_raw; do
  timeout 300 python bench.py --only-config $leg > gpurun_out/r5d/bench_${leg}.json 2> gpurun_out/r5d/bench_${leg}.err
done
for strip in 4 8; do STRINGWARS_AMD_NWP_STRIP=$strip timeout 300 python bench.py --only-config c4_letters52 > gpurun_out/r5d/bench_c4_letters52_w${strip}.json 2>/dev/null; done
timeout 300 python tools/bench_bounds.py > gpurun_out/r5d/bounds_table.jsonl 2> gpurun_out/r5d/bounds.err
timeout 600 python tools/bench_sizes.py > gpurun_out/r5d/c2_sizes.jsonl 2> gpurun_out/r5d/c2_sizes.err
