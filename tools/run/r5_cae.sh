set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_cae.py tests/test_gpu_round4.py tests/test_gpu_phase2.py -x -q > gpurun_out/tcae.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/tcae.log | cut -c1-200
for i in 1 2; do timeout -k 10 300 python bench.py --workload cae --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c -o c -- python bench.py --workload cae --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/c.log 2>&1 && python tools/rocpd_stats.py $(find gpurun_out/prof_c -name "*.db" | head -1) gpurun_out/r05_cae_kernel_stats.csv > gpurun_out/r05_cae_kernel_stats.txt; python tools/rocpd_sequence.py $(find gpurun_out/prof_c -name "*.db" | head -1) > gpurun_out/r05_cae_step_sequence.txt; rm -rf gpurun_out/prof_c
head -30 gpurun_out/r05_cae_kernel_stats.txt | cut -c1-150
