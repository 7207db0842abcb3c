set -o pipefail
for v in "SP_WGRAD_BIG_BLOCKS=0" "SP_WGRAD_BIG_BLOCKS=1" "SP_WGRAD_BIG_BLOCKS=2" "SP_WGRAD_BIG_BLOCKS=4" "SP_WGRAD_BIG_BLOCKS=0"; do
  echo "== $v"; env "$v" timeout -k 10 300 python bench.py --workload cae --steps 20 --warmup 5 --no-cpu-baseline --layers > gpurun_out/c.out 2> gpurun_out/c.err; grep -E "conv_wgrad  100->800" gpurun_out/c.err | cut -c1-100; python -c "import json; d=json.loads(open('gpurun_out/c.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
timeout -k 10 600 python -m pytest tests/test_gpu_cae.py -x -q 2>&1 | tail -1
