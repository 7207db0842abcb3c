"""Elementwise kernels of the U-Net at the calls the engine makes for the headline shape (B=4, 2x128^3, bf16; plane-major
concat buffers, dense per-part concat gradients): us and TB/s of algorithmic traffic per launch.  GPU box only."""
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
def nb(*ts): return sum(x.numel() * x.element_size() for x in ts)
def show(name, us, byts): print("%-34s %7.1f us  %5.2f TB/s  (floor at 8 TB/s %5.1f us)" % (name, us, byts / us / 1e6, byts / 8e6))


tot = 0.0
for (ld, lc, sd, sc) in ((46, 32, 124, 16), (25, 64, 58, 32)):
    low, skip = cl(ld, lc), cl(sd, sc)
    cat = cl(2 * ld, lc + sc)
    st = torch.zeros(R * (lc + sc), 2, dtype=torch.float64, device=dev)
    us = t(lambda: O.upsample2_crop_cat_fwd(low, skip, cat, dt, st, planar=True)); tot += us
    show("upcat %d->%d x%d+%d planar" % (ld, 2 * ld, lc, sc), us, nb(low, cat) + cat.numel() * 2 * sc // (lc + sc))
    g0, g1 = cl(2 * ld, lc), cl(2 * ld, sc)
    coef = torch.randn(3, lc + sc, device=dev)
    dzl = torch.empty_like(low); db = torch.zeros(R * lc, dtype=torch.float64, device=dev)
    us = t(lambda: O.upsample2_act_bwd(low, None, g0, coef, dt, L.ACT_LEAKY, 0.01, dzl, db, coef_stride=lc + sc)); tot += us
    show("upsample2_act_bwd %d<-%d x%d" % (ld, 2 * ld, lc), us, nb(low, dzl, g0))
    y, gp = cl(sd, sc), cl(sd // 2, sc)
    dz = torch.empty_like(y); cp = torch.randn(3, sc, device=dev); db2 = torch.zeros(R * sc, dtype=torch.float64, device=dev)
    us = t(lambda: O.pool_skip_act_bwd(y, gp, cp, None, g1, coef, 0, dt, L.ACT_LEAKY, 0.01, dz, db2, coef_c0=lc, coef_stride=lc + sc)); tot += us
    show("pool_skip_act_bwd %d x%d" % (sd, sc), us, nb(y, gp, dz, g1))
    p = cl(sd // 2, sc); st2 = torch.zeros(R * sc, 2, dtype=torch.float64, device=dev)
    us = t(lambda: O.maxpool2_fwd(y, p, dt, st2)); tot += us
    show("maxpool2_fwd %d x%d" % (sd, sc), us, nb(y, p))
for d, c in ((126, 16), (90, 16), (60, 32), (48, 32), (27, 64)):
    g, y = cl(d, c), cl(d, c); dz = torch.empty_like(g); coef = torch.randn(3, c, device=dev); db = torch.zeros(R * c, dtype=torch.float64, device=dev)
    us = t(lambda: O.bn_act_bwd(g, y, coef, dt, L.ACT_LEAKY, 0.01, dz, db)); tot += us if d != 126 else 0
    show("bn_act_bwd %d^3 x%d" % (d, c), us, nb(g, y, dz))
print("sum (one step's launches) %.0f us" % tot)
