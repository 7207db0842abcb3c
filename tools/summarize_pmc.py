"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel -> markdown.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by exactly 2x, WRITE_SIZE
is exact for 16-byte streaming stores; both columns are KB in the CSV.  Usage: summarize_pmc.py fetch.csv write.csv"""
import collections
import csv
import sys


def agg(path, cname):
    d = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == cname:
                k = r["Kernel_Name"].split("(")[0]
                d[k][0] += float(r["Counter_Value"])
                d[k][1] += 1
    return d


f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | FETCH_SIZE x2 MB/launch | WRITE_SIZE MB/launch |\n|---|---|---|---|")
for k in sorted(f, key=lambda k: -f[k][0])[:20]:
    n = f[k][1]
    wv = w.get(k, [0.0, 1])
    print("| `%s` | %d | %.1f | %.1f |" % (k[:70], n, 2 * f[k][0] / n / 1024, wv[0] / max(1, wv[1]) / 1024))
