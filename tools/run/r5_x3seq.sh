set -o pipefail
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/prof_x -o x -- python bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary --no-parity > gpurun_out/x.log 2>&1
python tools/rocpd_sequence.py $(find gpurun_out/prof_x -name "*.db" | head -1) > gpurun_out/r05_bf16x3_step_sequence.txt
rm -rf gpurun_out/prof_x
grep -n "copyBuffer" -B2 -A1 gpurun_out/r05_bf16x3_step_sequence.txt | cut -c1-120 | head -60
grep -c . gpurun_out/r05_bf16x3_step_sequence.txt
