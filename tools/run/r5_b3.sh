set -o pipefail
for v in "" "SP_PLAN_NT_SMALL_VOX=20000"; do
  echo "== $v"
  env $v ONLY=b3c1,b3c2 timeout -k 10 200 python tools/bench_conv.py fwd dgrad 2>&1 | grep -v "DICE\|amdgpu" | cut -c1-150
  env $v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('loss'))"
done
