set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_dp_graph.py tests/test_gpu_parallel_exact.py tests/test_gpu_dropin.py -x -q > gpurun_out/t5c.log 2>&1; echo "tests rc $?"; tail -14 gpurun_out/t5c.log | cut -c1-220
