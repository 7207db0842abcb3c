"""Diagnostic: build a -DSP_CONV_STAMPS copy of the library, run one conv layer, print the per-phase cycle shares."""
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
from stroke_prediction_amd.runtime import ops as O, plan as P
ci, co, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 4
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
dt = L.SP_BF16
dims = (d, d, d)
cpi, cpo = O.cpad(ci), O.cpad(co)
x = torch.randn((B,) + dims + (cpi,), device="cuda").bfloat16()
w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
if mode == "fwd":
    op = P.conv_fwd_op(ci, co, 3, 1, 0, dims, cpi, cpo, dt)
    r = O.ConvRunner(op, "cuda"); r.prep(w, torch.zeros(co, device="cuda"))
    y = O.alloc_cl(B, op.y_dims, cpo, dt, "cuda")
    st = torch.zeros(cpo, 2, dtype=torch.float64, device="cuda")
    sc, sh = torch.rand(cpi, device="cuda") + 0.5, torch.randn(cpi, device="cuda") * 0.1
    fn = lambda: r.run(x, y, B, sc, sh, L.ACT_LEAKY, 0.01, st if os.environ.get("STATS", "1") == "1" else None)
else:
    op0 = P.conv_fwd_op(ci, co, 3, 1, 0, dims, cpi, cpo, dt)
    op = P.conv_dgrad_op(ci, co, 3, 1, 0, dims, cpo, cpi, dt)
    r = O.ConvRunner(op, "cuda"); r.prep(w)
    dz = torch.randn((B,) + tuple(op0.y_dims) + (cpo,), device="cuda").bfloat16()
    g = O.alloc_cl(B, dims, cpi, dt, "cuda")
    fn = lambda: r.run(dz, g, B)
for _ in range(3):
    fn()
torch.cuda.synchronize()
n = 16384
buf = np.zeros((n, 6), dtype=np.uint64)
rc = L.load().sp_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), n)
assert rc == 0
dtot = buf[:, 5].astype(np.float64) - buf[:, 0].astype(np.float64)
ok = dtot > 0
names = ["stage issue", "wait+barrier", "K loop (MFMA)", "epilogue store", "stats reduce+atomics"]
if os.environ.get("FOLD", "1") == "1" and mode == "fwd":
    r.prep(w, torch.zeros(co, device="cuda"), sc, sh)
    fn = lambda: r.run(x, y, B, None, None, L.ACT_LEAKY, 0.01, st if os.environ.get("STATS", "1") == "1" else None)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    rc = L.load().sp_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), n)
    buf = buf.astype(np.float64) if buf.dtype != np.float64 else buf
    dtot = buf[:, 5] - buf[:, 0]
    ok = dtot > 0
print("blocks sampled", int(ok.sum()), "median total cycles/block %.0f" % np.median(dtot[ok]))
buf = buf.astype(np.float64)
for k in range(5):
    dd = (buf[:, k + 1] - buf[:, k])[ok]
    print("  %-30s median %8.0f cyc  (%.1f%%)" % (names[k], np.median(dd), 100 * np.median(dd) / np.median(dtot[ok])))
t0, t1 = buf[ok, 0].min(), buf[ok, 5].max()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print("kernel time %.1f us ; sum(block cycles)/256 CUs = %.0f cycles" % (e0.elapsed_time(e1) * 1e3, dtot[ok].sum() / 256))
print("span of sampled blocks: %.0f cycles; tile lds %d B, MT %d" % (t1 - t0, op.subs[0].tile["lds_bytes"], op.subs[0].tile["MT"]))
