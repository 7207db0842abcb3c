"""Per-layer table of the headline step's convolution kernels (VERDICT r4 "next" 1 / missing 3): for every conv / weight-gradient
launch of ONE step, in launch order -- layer label, serial duration (us), algorithmic TFLOP/s, FETCH_SIZE x2 / WRITE_SIZE MB
(gfx950 correction, MI355X_MICROARCH.md) and the MFMA pipe's busy share (SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES)).

usage: layer_table.py serial_sequence.txt fetch.csv write.csv sq.csv layer_order.json
  serial_sequence.txt  tools/rocpd_sequence.py of a SP_OVERLAP=0 replayed step (durations)
  *.csv                rocprofv3 --pmc counter_collection.csv of eager steps (separate passes)
  layer_order.json     bench.py --layers: [[tag, detail, flops], ...] of one eager step in launch order
"""
import csv
import json
import sys
from collections import defaultdict, deque

seqf, ff, wf, qf, of = sys.argv[1:6]
CONV = ("conv_zm3_kernel", "conv_igemm", "first_fwd_kernel", "wgrad_zr_kernel", "wgrad_dma_kernel", "wgrad_zs_kernel", "wgrad_kernel", "first_wgrad_kernel", "conv_fc_")


def last_step(rows):
    idx = [i for i, r in enumerate(rows) if "adam_hyp" in r[0]]
    return rows[idx[-2] + 1: idx[-1] + 1] if len(idx) >= 2 else rows


def counter(path, names):
    per = defaultdict(dict)
    kn = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in names:
            d = int(r["Dispatch_Id"])
            per[d][r["Counter_Name"]] = float(r["Counter_Value"])
            kn[d] = r["Kernel_Name"].split("(")[0]
    rows = [(kn[d], per[d]) for d in sorted(per)]
    return last_step(rows)


f = counter(ff, ("FETCH_SIZE",))
w = counter(wf, ("WRITE_SIZE",))
q = counter(qf, ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"))
dur = defaultdict(deque)
for l in open(seqf):
    if l.startswith("#"):
        continue
    p = l.split(None, 2)
    if len(p) == 3:
        dur[p[2].split("(")[0].strip()].append(float(p[1]))
order = json.load(open(of))
labels = deque(order)
assert [k for k, _ in f] == [k for k, _ in w], "FETCH / WRITE passes saw different launch sequences"
qd = defaultdict(deque)
for k, v in q:
    qd[k].append(v)
print("# one training step of the headline workload (U-Net 2 16 32 64 32 16 32 2, B=4, 2x128^3, bf16), convolution and weight-gradient launches in order")
print("%-12s %-44s %8s %9s %10s %10s %7s %9s" % ("pass", "layer", "us", "TFLOP/s", "fetch MB", "write MB", "TB/s", "MFMA busy"))
tot = defaultdict(float)
for (k, fv), (_, wv) in zip(f, w):
    if not any(c in k for c in CONV):
        continue
    tag, detail, flops = labels.popleft() if labels else ("?", k, 0.0)
    d = dur[k].popleft() if dur[k] else float("nan")
    qv = qd[k].popleft() if qd[k] else {}
    fb, wb = 2 * fv["FETCH_SIZE"] / 1024, wv["WRITE_SIZE"] / 1024
    busy = 100 * qv["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * qv["SQ_BUSY_CU_CYCLES"]) if qv.get("SQ_BUSY_CU_CYCLES") else float("nan")
    print("%-12s %-44s %8.1f %9.1f %10.1f %10.1f %7.2f %8.1f%%" % (tag, detail[:44], d, flops / d / 1e6 if d == d and d > 0 else 0.0, fb, wb, (fb + wb) / d, busy))
    tot["us"] += d; tot["flops"] += flops; tot["mb"] += fb + wb
print("%-12s %-44s %8.1f %9.1f %21.1f" % ("sum", "%d launches" % len(order), tot["us"], tot["flops"] / tot["us"] / 1e6, tot["mb"]))
if labels:
    print("# %d labels left over: the launch sequence and bench.py --layers disagree" % len(labels))
