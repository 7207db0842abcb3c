set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_round2.py -k "z_marching or headline" tests/test_gpu_round4.py tests/test_gpu_bf16x3.py -x -q > gpurun_out/t1.log 2>&1; echo "pytest rc $?" ; tail -5 gpurun_out/t1.log
SP_ZM_TILE=16 timeout -k 10 200 python tools/bench_conv.py fwd dgrad > gpurun_out/conv_t16.txt 2>&1
timeout -k 10 200 python tools/bench_conv.py fwd dgrad > gpurun_out/conv_auto.txt 2>&1
paste -d'\n' gpurun_out/conv_t16.txt gpurun_out/conv_auto.txt | grep -v DICE | cut -c1-200
