set -x
mkdir -p gpurun_out/r5s
for seed in 301 302 303; do
  timeout 400 python tests/soak.py --seconds 150 --seed $seed >> gpurun_out/r5s/soak.txt 2>&1; echo "seed $seed rc $?" >> gpurun_out/r5s/soak.txt
done
tail -20 gpurun_out/r5s/soak.txt
