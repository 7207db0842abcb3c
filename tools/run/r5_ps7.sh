set -o pipefail
V=$PWD/stroke-prediction_amd/lib/variants/nofront.so
for v in "X=1" "SP_LIB_PATH=$V" "X=1" "SP_LIB_PATH=$V"; do
  echo "== x3 ${v:0:12}"; env "$v" timeout -k 10 200 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary --layers > gpurun_out/x3.out 2> gpurun_out/x3.err; grep -E "48->16 @92x92x92 zm" gpurun_out/x3.err | grep -v wgrad | cut -c1-100; python -c "import json; d=json.loads(open('gpurun_out/x3.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
