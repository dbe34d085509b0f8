# What a round ends with, in one gpurun call: the whole -m gpu suite, the bench line (C2, with every configuration's leg) and C5 at its full
# 100 M pairs. Outputs under gpurun_out/round_end/; copy bench_c2.json / bench_c5_100m.json into profiles/<round>/ afterwards.
# (The PMC passes are tools/refresh_profiles.sh's; run that first when a kernel's sources changed.)
set -x
mkdir -p gpurun_out/round_end
timeout 2700 python -m pytest tests/ -q -m gpu > gpurun_out/round_end/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/round_end/tests.log
tail -6 gpurun_out/round_end/tests.log
timeout 900 python bench.py > gpurun_out/round_end/bench_c2.json 2> gpurun_out/round_end/bench_c2.err
timeout 600 python bench.py --config c5 --steps 20 --warmup 2 --no-cpu-baseline --no-configs > gpurun_out/round_end/bench_c5_100m.json 2> gpurun_out/round_end/bench_c5_100m.err
