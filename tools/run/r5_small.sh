set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tfull.log 2>&1; echo "gpu tests rc $?"; tail -2 gpurun_out/tfull.log | cut -c1-200
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-parity --no-cpu-baseline --no-secondary > gpurun_out/b.out 2> gpurun_out/b.err; python -c "import json; d=json.loads(open('gpurun_out/b.out').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], d['config'].get('loss'))"
done
