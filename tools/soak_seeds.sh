# Three fresh soak seeds (tests/soak.py: random batches through every route against the oracle), 150 s each. usage: soak_seeds.sh [first-seed]
set -x
mkdir -p gpurun_out/soak
first=${1:-301}
for seed in $first $((first+1)) $((first+2)); do
  timeout 400 python tests/soak.py --seconds 150 --seed $seed >> gpurun_out/soak/soak.txt 2>&1; echo "seed $seed rc $?" >> gpurun_out/soak/soak.txt
done
tail -20 gpurun_out/soak/soak.txt
