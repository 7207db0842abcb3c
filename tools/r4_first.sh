set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
python bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-parity --layers > gpurun_out/r4a/f32_layers.json 2> gpurun_out/r4a/f32_layers.txt && \
SP_OVERLAP=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r4a/p -o s -- python bench.py --dtype f32 --steps 5 --warmup 2 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing > gpurun_out/r4a/f32_prof.log 2>&1 && \
python tools/rocpd_stats.py $(find gpurun_out/r4a/p -name "*.db" | head -1) gpurun_out/r4a/f32_kernel_stats_serial.csv > gpurun_out/r4a/f32_kernel_stats_serial.txt && rm -rf gpurun_out/r4a/p && \
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r4a/bf16.json 2> gpurun_out/r4a/bf16.err
tail -3 gpurun_out/r4a/f32_layers.txt
