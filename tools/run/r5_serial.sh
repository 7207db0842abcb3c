set -o pipefail
MIN_US=12 bash tools/probes/serial_traffic.sh bf16 unet > gpurun_out/r05_bench_serial_traffic.txt 2>&1
tail -60 gpurun_out/r05_bench_serial_traffic.txt | cut -c1-135
