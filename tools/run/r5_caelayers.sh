set -o pipefail
SP_WGRAD_DEBUG=1 timeout -k 10 300 python bench.py --workload cae --steps 5 --warmup 2 --no-cpu-baseline --layers > gpurun_out/c.out 2> gpurun_out/c.err
grep -E "^wgrad " gpurun_out/c.err | sort -u | head -30
grep -E "conv_wgrad" gpurun_out/c.err | cut -c1-110 | sort -t@ -k2 | head -40
