set -x
mkdir -p gpurun_out/r5b
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5b/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5b/tests.log
tail -15 gpurun_out/r5b/tests.log
