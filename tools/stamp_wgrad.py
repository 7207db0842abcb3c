"""Diagnostic: phase shares of the DMA weight-gradient kernel (issue / MFMA loop / wait+barrier), stamp build only."""
import sys, os, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L
src = [os.path.join(L.CSRC_DIR, s) for s in L.SOURCES]
dbg = "/tmp/libstroke_amd_stamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSP_CONV_STAMPS", "-fgpu-rdc"] + os.environ.get("XDEF", "").split() + ["-o", dbg] + src, check=True)
L.LIB_PATH = dbg
from stroke_prediction_amd.runtime import ops as O
ci, co, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 4
dims, od = (d,) * 3, (d - 2,) * 3
cpi, cpo = O.cpad(ci), O.cpad(co)
x = torch.randn((B,) + dims + (cpi,), device="cuda").bfloat16()
dz = torch.randn((B,) + od + (cpo,), device="cuda").bfloat16()
w = torch.zeros(co, ci, 3, 3, 3, device="cuda")
wg = O.WgradRunner(ci, co, 3, 1, 0, dims, od, cpi, cpo, ci * 27, 27, L.SP_BF16, "cuda")
fn = lambda: wg.run(x, dz, B, w)
for _ in range(2):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
buf = np.zeros((256, 6), dtype=np.uint64)
assert L.load().sp_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), 256) == 0
b = buf.astype(np.float64)
n = b[:, 3].mean()
print("kernel %.1f us; tiles/block %.1f; per tile cycles: issue %.0f  compute %.0f  wait+barrier %.0f" % (
    e0.elapsed_time(e1) * 1e3, n, (b[:, 0] / b[:, 3]).mean(), (b[:, 1] / b[:, 3]).mean(), (b[:, 2] / b[:, 3]).mean()))
