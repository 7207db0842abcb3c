"""Compact per-kernel resource table of one HIP source: python tools/kres.py csrc/file.hip [-DSP_HALF_F16] [filter]
(VGPRs, AGPRs, spills, scratch, LDS, occupancy from -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
extra = [a for a in sys.argv[2:] if a.startswith("-")]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                      "-c", src, "-o", "/tmp/kres.o"] + extra, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs|VGPRs Spill|SGPRs Spill): (.*?) \[-R", line)
    if "error" in line:
        print(line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
for r in rows:
    n = subprocess.run(["/usr/bin/c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("unsigned short", "u16")
    if flt and not any(f in n for f in flt):
        continue
    print("%-78s v%-4s a%-4s vspill %-3s sspill %-3s scratch %-4s occ %s" % (n[:78], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("Occupancy")))
