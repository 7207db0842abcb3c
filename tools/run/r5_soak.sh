set -o pipefail
for dt in bf16 bf16x3 f16x3; do
  timeout -k 10 300 python bench.py --dtype $dt --steps 1500 --warmup 10 --no-parity --no-cpu-baseline --no-secondary > gpurun_out/b.out 2> gpurun_out/b.err; python -c "import json; d=json.loads(open('gpurun_out/b.out').read().strip().splitlines()[-1]); print('$dt', 'ms', d['ms_per_step'], 'loss after 1510 steps', d['config'].get('loss'))"
done
timeout -k 10 300 python bench.py --workload cae --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/b.out 2> gpurun_out/b.err; python -c "import json; d=json.loads(open('gpurun_out/b.out').read().strip().splitlines()[-1]); print('cae ms', d['ms_per_step'], d['config'].get('loss'))"
