set -o pipefail
for v in "" "SP_ZM_VARIANT=000001" "SP_ZM_NW=4"; do
  echo "== $v"
  env $v ONLY=b1c2,b2c1,b2c2,b4c1,b4c2,b5c1,b5c2 timeout -k 10 200 python tools/bench_conv.py fwd dgrad 2>&1 | grep -v "DICE\|amdgpu" | cut -c1-150
done
