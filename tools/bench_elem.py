"""Time the HBM-bound elementwise kernels at the headline shapes (B=4): bn_act_bwd, pool_skip_act_bwd, upsample2_act_bwd,
upsample2_crop_cat_fwd, maxpool2_fwd; prints us and effective TB/s of algorithmic traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stroke_prediction_amd.runtime import lib as L, ops as O
dev, dt, B = "cuda:0", L.SP_BF16, 4
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
g, y = cl(126, 16), cl(126, 16); dz = torch.empty_like(g); coef = torch.randn(3, 16, device=dev); db = torch.zeros(16, dtype=torch.float64, device=dev)
us = t(lambda: O.bn_act_bwd(g, y, coef, dt, L.ACT_LEAKY, 0.01, dz, db)); print("bn_act_bwd 126^3x16      %7.1f us  %.2f TB/s" % (us, nbytes(g, y, dz) / us / 1e6))
# pool_skip_act_bwd: y 124^3x16, gp 62^3x16, gs = cat-grad 92^3x48 (skip channels 32..47)
y = cl(124, 16); gp = cl(62, 16); gs = cl(92, 48); cat = cl(92, 48); dz = torch.empty_like(y)
cp, cs = torch.randn(3, 16, device=dev), torch.randn(3, 48, device=dev)
us = t(lambda: O.pool_skip_act_bwd(y, gp, cp, cat, gs, cs, 32, dt, L.ACT_LEAKY, 0.01, dz, db)); print("pool_skip_act_bwd 124^3    %7.1f us  %.2f TB/s" % (us, (nbytes(y, gp, dz) + gs.numel() * 2 // 3) / us / 1e6))
# upsample2_act_bwd: low 46^3x32, cat-grad 92^3x48
low = cl(46, 32); dzl = torch.empty_like(low); db32 = torch.zeros(32, dtype=torch.float64, device=dev)
us = t(lambda: O.upsample2_act_bwd(low, cat, gs, cs, dt, L.ACT_LEAKY, 0.01, dzl, db32)); print("upsample2_act_bwd 46->92   %7.1f us  %.2f TB/s" % (us, (nbytes(low, dzl) + gs.numel() * 2 * 2 // 3) / us / 1e6))
# upcat fwd
skip = cl(124, 16); st = torch.zeros(48, 2, dtype=torch.float64, device=dev)
us = t(lambda: O.upsample2_crop_cat_fwd(low, skip, cat, dt, st)); print("upsample2_crop_cat_fwd 92  %7.1f us  %.2f TB/s" % (us, (nbytes(low, cat) + cat.numel() * 2 // 3) / us / 1e6))
# maxpool
p = cl(62, 16); st16 = torch.zeros(16, 2, dtype=torch.float64, device=dev)
us = t(lambda: O.maxpool2_fwd(y, p, dt, st16)); print("maxpool2_fwd 124^3x16      %7.1f us  %.2f TB/s" % (us, nbytes(y, p) / us / 1e6))
