set -o pipefail
for v in "SP_ZM_PSER_FALLBACK=1" "SP_ZM_PSER_FALLBACK=0" "SP_ZM_PSER_FALLBACK=1" "SP_ZM_PSER_FALLBACK=0"; do
  echo "== $v"; env "$v" timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers > gpurun_out/b.out 2> gpurun_out/b.err; grep -E "96->32 @50" gpurun_out/b.err | cut -c1-100; python -c "import json; d=json.loads(open('gpurun_out/b.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
