set -o pipefail
for v in "" "SP_ZM_SPLIT=1,3" "SP_ZM_SPLIT=2,2"; do
  echo "== $v"
  env $v ONLY=b5c1,b4c1,b2c2,b4c2 timeout -k 10 200 python tools/bench_conv.py fwd dgrad 2>&1 | grep -v "DICE\|amdgpu" | cut -c1-150
done
