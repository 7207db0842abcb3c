"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes, CSV output) -> profiles/<tag>_pmc_traffic.md and
profiles/traffic.json (HBM bytes per launch of the conv_igemm / conv_wgrad kernel families, read by bench.py).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-reports wide coalesced reads by exactly 2x,
WRITE_SIZE is exact for 16-byte streaming stores; both are KB in the CSV.
usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv TAG [KEY_PREFIX [WORKLOAD_TEXT]]
KEY_PREFIX (e.g. "cae_"): the families are stored as <prefix>conv_igemm / <prefix>conv_wgrad next to the existing keys."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(path, cname):
    d = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == cname:
                k = r["Kernel_Name"].split("(")[0]
                d[k][0] += float(r["Counter_Value"]); d[k][1] += 1
    return d


f, w, tag = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
prefix = sys.argv[4] if len(sys.argv) > 4 else ""
workload = sys.argv[5] if len(sys.argv) > 5 else "bench.py --steps 2 --warmup 1 --no-graph, U-Net B=4 2x128^3 bf16"
lines = ["# %s HBM traffic per kernel (%s; separate --pmc passes)" % (tag, workload), "",
         "| kernel | launches | FETCH_SIZE x2 MB/launch | WRITE_SIZE MB/launch |", "|---|---|---|---|"]
for k in sorted(f, key=lambda k: -f[k][0])[:24]:
    n = f[k][1]; wv = w.get(k, [0.0, 1])
    lines.append("| `%s` | %d | %.1f | %.1f |" % (k[:70], n, 2 * f[k][0] / n / 1024, wv[0] / max(1, wv[1]) / 1024))
open(os.path.join(ROOT, "profiles", "%s_pmc_traffic.md" % tag), "w").write("\n".join(lines) + "\n")
fam = {"conv_igemm": ("conv_igemm", "conv_zm3_kernel", "conv_fc_", "first_fwd_kernel"),
       "conv_wgrad": ("wgrad_dma_kernel", "wgrad_zs_kernel", "wgrad_zr_kernel", "wgrad_kernel", "first_wgrad_kernel")}
tj = os.path.join(ROOT, "profiles", "traffic.json")
out = json.load(open(tj)) if (prefix and os.path.exists(tj)) else {}
for name, pats in fam.items():
    fb = sum(v[0] for k, v in f.items() if any(p in k for p in pats)) * 2 * 1024
    wb = sum(v[0] for k, v in w.items() if any(p in k for p in pats)) * 1024
    n = sum(v[1] for k, v in f.items() if any(p in k for p in pats))
    if n == 0:
        continue
    out[prefix + name] = {"launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n, "bytes_per_launch": (fb + wb) / n}
# the whole step: every kernel of the pass divided by the steps it ran (one fused Adam launch per step)
nstep = sum(v[1] for k, v in f.items() if "adam_hyp_kernel" in k or "adam_kernel" in k)
if nstep:
    out[prefix + "step"] = {"steps": nstep, "fetch_bytes_per_step": sum(v[0] for v in f.values()) * 2 * 1024 / nstep,
                            "write_bytes_per_step": sum(v[0] for v in w.values()) * 1024 / nstep}
    out[prefix + "step"]["bytes_per_step"] = out[prefix + "step"]["fetch_bytes_per_step"] + out[prefix + "step"]["write_bytes_per_step"]
out["_source" + ("_" + prefix.rstrip("_") if prefix else "")] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python %s`; FETCH_SIZE doubled per the gfx950 correction; see profiles/%s_pmc_traffic.md" % (workload, tag)
json.dump(out, open(tj, "w"), indent=1)
# the GPU box returns gpurun_out/ only: leave copies there (copy them into profiles/ after the call)
go = os.path.join(ROOT, "gpurun_out")
if os.path.isdir(go):
    open(os.path.join(go, "%s_pmc_traffic.md" % tag), "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(go, "traffic_%s.json" % tag), "w"), indent=1)
print("\n".join(lines[:16])); print(json.dumps(out, indent=1))
