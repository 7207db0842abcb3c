set -o pipefail
bash tools/profile_round.sh r05 unet > gpurun_out/prof_unet.log 2>&1; echo "unet rc $?"
tail -3 gpurun_out/prof_unet.log
head -40 gpurun_out/r05_bench_layers.txt | cut -c1-140
