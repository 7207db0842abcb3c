set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tfull.log 2>&1; echo "gpu tests rc $?"; tail -5 gpurun_out/tfull.log | cut -c1-200
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_bench.json").read().strip().splitlines()[-1])
print("ms", d["ms_per_step"], "value", d["value"])
print("roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline"].items() if not isinstance(v,(dict,str))})
print("config", {k:(round(v,6) if isinstance(v,float) else v) for k,v in d["config"].items()})
print("secondary", {k:(round(v.get("ms_per_step",0),3) if isinstance(v,dict) else v) for k,v in d["secondary"].items()})
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k in ("value","cores","kind")})
PY
