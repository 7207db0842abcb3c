set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -k "plane_serial" -x -q > gpurun_out/t5f.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/t5f.log | cut -c1-220
