set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -k "plane_serial" -x -q 2>&1 | tail -2
for v in "SP_ZM_PSER=" "SP_ZM_PSER=all"; do
  echo "== $v"
  env "$v" ONLY=b4c1,b5c1 timeout -k 10 200 python tools/bench_conv.py fwd 2>&1 | grep -v "DICE\|amdgpu" | cut -c1-110
done
