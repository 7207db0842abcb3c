set -o pipefail
SP_BENCH_BACKEND=gloo SP_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-secondary > gpurun_out/n2.json 2> gpurun_out/n2.err; echo "rc $?"
tail -3 gpurun_out/n2.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/n2.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["ms_per_step"], d["value"], {k:v for k,v in d["config"].items() if k.startswith("rccl") or k in ("parallelism","dp_mode","grad_exchange","launch")})
PY
