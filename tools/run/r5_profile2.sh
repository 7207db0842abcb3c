set -o pipefail
bash tools/profile_round.sh r05 x3 > gpurun_out/prof_x3.log 2>&1; echo "x3 rc $?"
bash tools/profile_round.sh r05 cae > gpurun_out/prof_cae.log 2>&1; echo "cae rc $?"
head -12 gpurun_out/r05_bench_bf16x3_layers.txt | cut -c1-120
tail -1 gpurun_out/r05_bench_bf16x3_layers.txt | cut -c1-300
