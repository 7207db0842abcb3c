# usage: serial_traffic.sh [dtype [workload]] -- serial (SP_OVERLAP=0) kernel durations of a training step next to the HBM traffic of every kernel (separate --pmc passes)
set -o pipefail
export TMPDIR=/tmp
DT=${1:-fp8}; WL=${2:-unet4}
OUT=gpurun_out/prof_fp8serial; rm -rf $OUT; mkdir -p $OUT
SP_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $OUT/s -o s -- python bench.py --workload $WL --dtype $DT --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary > $OUT/s.log 2>&1 &&
python tools/rocpd_sequence.py $(find $OUT/s -name "*.db" | head -1) > gpurun_out/${DT}_serial_sequence.txt &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o f -- python bench.py --workload $WL --dtype $DT --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-secondary > $OUT/f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o w -- python bench.py --workload $WL --dtype $DT --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing --no-secondary > $OUT/w.log 2>&1
python - $OUT $DT <<'PY'
import csv, glob, sys, collections
out, dt = sys.argv[1], sys.argv[2]
def seq(sub, cname):
    path = glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == cname]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])) for r in rows]
f, w = seq("f", "FETCH_SIZE"), seq("w", "WRITE_SIZE")
# last step = the dispatches after the second-to-last adam_hyp_kernel
def last_step(s):
    idx = [i for i, (k, _) in enumerate(s) if "adam_hyp" in k]
    return s[idx[-2] + 1: idx[-1] + 1] if len(idx) >= 2 else s
f, w = last_step(f), last_step(w)
assert [k for k, _ in f] == [k for k, _ in w], (len(f), len(w))
lines = [l.split(None, 2) for l in open("gpurun_out/%s_serial_sequence.txt" % dt) if not l.startswith("#")]
dur = [(l[2].split("(")[0].strip(), float(l[1])) for l in lines if len(l) == 3]
print("# %d kernels in the eager step (counters), %d in the replayed serial step (durations)" % (len(f), len(dur)))
# match by order of kernel name occurrences
from collections import defaultdict, deque
dq = defaultdict(deque)
for k, d in dur:
    dq[k].append(d)
tot = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
print("%-84s %9s %10s %10s %8s" % ("kernel", "us", "fetch MB", "write MB", "TB/s"))
for (k, fv), (_, wv) in zip(f, w):
    d = dq[k].popleft() if dq[k] else float("nan")
    fb, wb = 2 * fv / 1024, wv / 1024          # KB -> MB; gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md)
    t = tot[k]; t[0] += 1; t[1] += d; t[2] += fb; t[3] += wb
    if d == d and d > float(__import__("os").environ.get("MIN_US", "100")):
        print("%-84s %9.1f %10.1f %10.1f %8.2f" % (k[:84], d, fb, wb, (fb + wb) / d))
print()
print("%-84s %4s %9s %10s %10s %8s" % ("kernel (sum over the step)", "n", "us", "fetch MB", "write MB", "TB/s"))
for k, t in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-84s %4d %9.1f %10.1f %10.1f %8.2f" % (k[:84], t[0], t[1], t[2], t[3], (t[2] + t[3]) / max(t[1], 1e-9)))
PY
rm -rf $OUT
