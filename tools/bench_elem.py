"""Time the HBM-bound elementwise kernels at the headline shapes (B=4): bn_act_bwd, pool_skip_act_bwd, upsample2_act_bwd,
upsample2_crop_cat_fwd, maxpool2_fwd; prints us and effective TB/s of algorithmic traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stroke_prediction_amd.runtime import lib as L, ops as O
dev, dt, B = "cuda:0", L.SP_BF16, 4
R = L.SP_REDUCE_ROWS
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def cl(d, c): return torch.randn(B, d, d, d, c, device=dev).bfloat16()
def nbytes(*ts): return sum(x.numel() * x.element_size() for x in ts)
# bn_act_bwd @126^3 x16
g, y = cl(126, 16), cl(126, 16); dz = torch.empty_like(g); coef = torch.randn(3, 16, device=dev); db = torch.zeros(R * 16, dtype=torch.float64, device=dev)
us = t(lambda: O.bn_act_bwd(g, y, coef, dt, L.ACT_LEAKY, 0.01, dz, db)); print("bn_act_bwd 126^3x16      %7.1f us  %.2f TB/s" % (us, nbytes(g, y, dz) / us / 1e6))
# pool_skip_act_bwd: y 124^3x16, gp 62^3x16, gs = cat-grad 92^3x48 (skip channels 32..47)
y = cl(124, 16); gp = cl(62, 16); gs = cl(92, 48); cat = cl(92, 48); dz = torch.empty_like(y)
cp, cs = torch.randn(3, 16, device=dev), torch.randn(3, 48, device=dev)
us = t(lambda: O.pool_skip_act_bwd(y, gp, cp, cat, gs, cs, 32, dt, L.ACT_LEAKY, 0.01, dz, db)); print("pool_skip_act_bwd 124^3    %7.1f us  %.2f TB/s" % (us, (nbytes(y, gp, dz) + gs.numel() * 2 // 3) / us / 1e6))
# upsample2_act_bwd: low 46^3x32, cat-grad 92^3x48
low = cl(46, 32); dzl = torch.empty_like(low); db32 = torch.zeros(R * 32, dtype=torch.float64, device=dev)
us = t(lambda: O.upsample2_act_bwd(low, cat, gs, cs, dt, L.ACT_LEAKY, 0.01, dzl, db32)); print("upsample2_act_bwd 46->92   %7.1f us  %.2f TB/s" % (us, (nbytes(low, dzl) + gs.numel() * 2 * 2 // 3) / us / 1e6))
# upcat fwd
skip = cl(124, 16); st = torch.zeros(R * 48, 2, dtype=torch.float64, device=dev)
us = t(lambda: O.upsample2_crop_cat_fwd(low, skip, cat, dt, st)); print("upsample2_crop_cat_fwd 92  %7.1f us  %.2f TB/s" % (us, (nbytes(low, cat) + cat.numel() * 2 // 3) / us / 1e6))
# maxpool
p = cl(62, 16); st16 = torch.zeros(R * 16, 2, dtype=torch.float64, device=dev)
us = t(lambda: O.maxpool2_fwd(y, p, dt, st16)); print("maxpool2_fwd 124^3x16      %7.1f us  %.2f TB/s" % (us, nbytes(y, p) / us / 1e6))
# ---- small shapes (block 3/4 side) and the cost of the statistics atomics (stats=None skips them)
low2 = cl(25, 64); skip2 = cl(58, 32); cat2 = cl(50, 96); st96 = torch.zeros(R * 96, 2, dtype=torch.float64, device=dev)
for s_, nm in ((st96, "stats"), (None, "no stats")):
    us = t(lambda: O.upsample2_crop_cat_fwd(low2, skip2, cat2, dt, s_)); print("upcat 25->50 x96 %-9s %7.1f us" % (nm, us))
for s_, nm in ((st, "stats"), (None, "no stats")):
    us = t(lambda: O.upsample2_crop_cat_fwd(low, skip, cat, dt, s_)); print("upcat 46->92 x48 %-9s %7.1f us" % (nm, us))
y2 = cl(58, 32); p2 = cl(29, 32); st32 = torch.zeros(R * 32, 2, dtype=torch.float64, device=dev)
for s_, nm in ((st32, "stats"), (None, "no stats")):
    us = t(lambda: O.maxpool2_fwd(y2, p2, dt, s_)); print("maxpool 58^3x32 %-9s %7.1f us" % (nm, us))
for s_, nm in ((st16, "stats"), (None, "no stats")):
    us = t(lambda: O.maxpool2_fwd(y, p, dt, s_)); print("maxpool 124^3x16 %-9s %7.1f us" % (nm, us))
gs2 = cl(50, 96); cs2 = torch.randn(3, 96, device=dev); dzl2 = torch.empty_like(low2); db64 = torch.zeros(R * 64, dtype=torch.float64, device=dev)
for s_, nm in ((db64, "dbias"), (None, "no dbias")):
    us = t(lambda: O.upsample2_act_bwd(low2, cat2, gs2, cs2, dt, L.ACT_LEAKY, 0.01, dzl2, s_)); print("up_bwd 25->50 x64 %-9s %7.1f us" % (nm, us))
dz2 = torch.empty_like(y2); cp2 = torch.randn(3, 32, device=dev)
for s_, nm in ((db32, "dbias"), (None, "no dbias")):
    us = t(lambda: O.pool_skip_act_bwd(y2, p2, cp2, cat2, gs2, cs2, 64, dt, L.ACT_LEAKY, 0.01, dz2, s_)); print("pool_skip_bwd 58^3x32 %-9s %7.1f us" % (nm, us))
g3, y3 = cl(27, 64), cl(27, 64); dz3 = torch.empty_like(g3); coef3 = torch.randn(3, 64, device=dev)
for s_, nm in ((db64, "dbias"), (None, "no dbias")):
    us = t(lambda: O.bn_act_bwd(g3, y3, coef3, dt, L.ACT_LEAKY, 0.01, dz3, s_)); print("bn_act_bwd 27^3x64 %-9s %7.1f us" % (nm, us))
