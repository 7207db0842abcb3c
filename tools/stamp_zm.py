"""Diagnostic: per-step cycle shares of the z-marching convolution kernel (csrc/sp_conv_zm.hip built with -DSP_ZM_STAMPS
into lib/variants/zmstamps.so by tools/build_variant.sh).  usage: SP_LIB_PATH=.../zmstamps.so python tools/stamp_zm.py CIN COUT N [fwd|dgrad]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L, ops as O, plan as P

ci, co, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
B, dt, dims = 4, L.SP_BF16, (d, d, d)
w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
if mode == "fwd":
    op = P.conv_fwd_op(ci, co, 3, 1, 0, dims, ci, co, dt)
    r = O.ConvRunner(op, "cuda", zm_batch=B)
    r.prep(w, torch.zeros(co, device="cuda"))
    x = torch.randn((B,) + dims + (ci,), device="cuda").bfloat16()
    y = O.alloc_cl(B, op.y_dims, co, dt, "cuda")
    st = torch.zeros(64 * co * 2, dtype=torch.float64, device="cuda")
    fn = lambda: r.run(x, y, B, None, None, L.ACT_LEAKY, 0.01, st, stats_nrep=64)
else:
    op0 = P.conv_fwd_op(ci, co, 3, 1, 0, dims, ci, co, dt)
    op = P.conv_dgrad_op(ci, co, 3, 1, 0, dims, co, ci, dt)
    r = O.ConvRunner(op, "cuda", zm_batch=B)
    r.prep(w)
    dz = torch.randn((B,) + tuple(op0.y_dims) + (co,), device="cuda").bfloat16()
    g = O.alloc_cl(B, dims, ci, dt, "cuda")
    fn = lambda: r.run(dz, g, B)
assert r.uses_zm()
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
buf = np.zeros((1024, 8, 8), dtype=np.uint64)
lib = L.load()
lib.sp_debug_zm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.sp_debug_zm_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
b = buf.astype(np.float64)
ok = b[:, 0, 4] > 0
b = b[ok][:, :r.zm["NW"]]
z = r.zm
print("%d->%d @%d %s: P=%d NT=%d MT=%d NW=%d KS=%d  kernel %.1f us, %d workgroups stamped" % (ci, co, d, mode, z["P"], z["NT"], z["MT"], z["NW"], z["KS"], e0.elapsed_time(e1) * 1e3, len(b)))
steps = b[:, :, 3].mean()
tot = b[:, :, 4].mean()
clock = b[:, :, 4].mean() / (b[:, :, 7].mean() * 10.0) if b[:, :, 7].mean() > 0 else float("nan")      # cycles per ns
print("  steps per workgroup %.1f; total cycles per wave %.0f (= %.1f us at %.2f GHz in-kernel clock)" % (steps, tot, tot / clock / 1e3, clock))
names = ["wait + barrier", "DMA issue", "K loop + epilogue"]
for k in range(3):
    print("  %-20s %8.0f cycles per step  (%.1f %% of the kernel)" % (names[k], b[:, :, k].mean() / steps, 100 * b[:, :, k].mean() / tot))
fs = b[:, :, 6].mean()
mfma = z["KS"] * z["MT"] * z["NT"] * 3
print("  fast-path steps %.1f per workgroup: %.0f cycles each for %d MFMAs (%d cycles of matrix pipe)" % (fs, b[:, :, 5].mean() / max(fs, 1), mfma, mfma * 16))
print("  outside the steps (prologue, weights, statistics flush): %.1f %%" % (100 * (tot - b[:, :, 0:3].sum(axis=2).mean()) / tot))
