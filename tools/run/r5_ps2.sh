set -o pipefail
for v in "SP_ZM_PSER=" "SP_ZM_PSER=all"; do
  echo "== $v"
  env "$v" ONLY=b4c1,b5c1,b2c2 timeout -k 10 200 python tools/bench_conv.py fwd 2>&1 | grep -v "DICE\|amdgpu" | cut -c1-110
done
for v in "SP_ZM_PSER=" "SP_ZM_PSER=3,1,0;3,1,2" "SP_ZM_PSER=" "SP_ZM_PSER=3,1,0;3,1,2"; do
  echo "== $v"; env "$v" timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
for v in "SP_ZM_PSER=" "SP_ZM_PSER=3,1,0;3,1,2"; do
  echo "== x3 $v"; env "$v" timeout -k 10 200 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers 2>&1 | grep -E "48->16 @92|96->32 @50x50x50 |ms_per_step" | cut -c1-100 | sed -e 's/"value.*//'
done
