"""Summarise rocprofv3 --pmc SQ_* counter CSVs (one or more passes) per kernel: averages per launch, plus the derived shares
the CDNA4 guide names (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES
(quad-cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles.  usage: pmc_sq.py pass1.csv [pass2.csv ...] [--match substr]"""
import collections
import csv
import sys

args = sys.argv[1:]
match = ""
if "--match" in args:
    i = args.index("--match")
    match = args[i + 1]
    del args[i:i + 2]
files = args
vals = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in files:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0]
            if match and match not in k:
                continue
            v = vals[k][r["Counter_Name"]]
            v[0] += float(r["Counter_Value"])
            v[1] += 1
for k, cs in sorted(vals.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    avg = {c: v[0] / max(1, v[1]) for c, v in cs.items()}
    n = max(v[1] for v in cs.values())
    print("## `%s`  (%d launches)" % (k[:110], n))
    wc = avg.get("SQ_WAVE_CYCLES")
    for c in sorted(avg):
        extra = ""
        if wc and c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                        "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR"):
            extra = "  = %.1f %% of wave cycles" % (100 * avg[c] / wc)
        if c == "SQ_VALU_MFMA_BUSY_CYCLES" and "SQ_BUSY_CU_CYCLES" in avg:
            extra = "  = %.1f %% of busy CU cycles x 4 SIMDs" % (100 * avg[c] / (4 * avg["SQ_BUSY_CU_CYCLES"]))
        print("  %-28s %16.0f%s" % (c, avg[c], extra))
