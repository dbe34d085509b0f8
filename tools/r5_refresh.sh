set -x
timeout 3000 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
ls gpurun_out/refresh | head -80
