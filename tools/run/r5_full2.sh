set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/tfull.log 2>&1; echo "gpu tests rc $?"; tail -6 gpurun_out/tfull.log | cut -c1-200
