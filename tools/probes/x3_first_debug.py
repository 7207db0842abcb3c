"""debug probe: pair first-layer kernel vs float64, error map per (sample, z, y-tile, x-tile)"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import stroke_prediction_amd
from stroke_prediction_amd.runtime import lib as L, ops as O
DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
for dims in [(10, 13, 70), (10, 13, 40), (12, 16, 66), (12, 16, 130)]:
    B = 2
    x = torch.randn(B, 2, *dims, generator=g)
    w = torch.randn(16, 2, 3, 3, 3, generator=g) / math.sqrt(54)
    b = torch.randn(16, generator=g) * 0.1
    scale, shift = torch.tensor([1.3, 0.7]), torch.tensor([0.2, -0.4])
    wf_h = torch.zeros(3 * 64 * 8, dtype=torch.bfloat16, device=DEV); wf_l = torch.zeros_like(wf_h)
    bias_f = torch.zeros(16, device=DEV)
    sc = torch.zeros(16, device=DEV); sc[:2] = scale
    sh = torch.zeros(16, device=DEV); sh[:2] = shift
    wd, bd, xd = w.to(DEV), b.to(DEV), x.to(DEV)
    st = O.stream()
    L.call("sp_first_prep_hl", O.ptr(wd), O.ptr(bd), O.ptr(sc), O.ptr(sh), O.ptr(wf_h), O.ptr(wf_l), O.ptr(bias_f), 16, st)
    od = tuple(d - 2 for d in dims)
    pair = torch.full((2, B) + od + (16,), 7.0, dtype=torch.bfloat16, device=DEV)
    y_h, y_l = pair[0], pair[1]
    stats = torch.zeros(4 * 32, dtype=torch.float64, device=DEV)
    L.call("sp_first_conv_fwd_hl", O.ptr(xd), B, *dims, O.ptr(wf_h), O.ptr(wf_l), O.ptr(bias_f), L.ACT_LEAKY, 0.01, O.ptr(y_h), O.ptr(y_l), O.ptr(stats), 4, 16, st)
    yb = torch.full((B,) + od + (16,), 7.0, dtype=torch.bfloat16, device=DEV)
    L.call("sp_first_prep_n", O.ptr(wd), O.ptr(bd), O.ptr(sc), O.ptr(sh), O.ptr(wf_h), O.ptr(bias_f), 16, st)
    L.call("sp_first_conv_fwd_n", O.ptr(xd), B, *dims, O.ptr(wf_h), O.ptr(bias_f), L.ACT_LEAKY, 0.01, O.ptr(yb), None, 4, 16, None, 0, st)
    xn = x.double() * scale.double().view(1, 2, 1, 1, 1) + shift.double().view(1, 2, 1, 1, 1)
    ref = F.leaky_relu(F.conv3d(xn, w.double(), b.double()), 0.01)
    got = (y_h.double() + y_l.double()).cpu().permute(0, 4, 1, 2, 3)
    gb = yb.double().cpu().permute(0, 4, 1, 2, 3)
    e = (got - ref).abs().amax(dim=1)
    eb = (gb - ref).abs().amax(dim=1)
    print(dims, "pair max err %.3g  bf16 max err %.3g  ref max %.3g" % (e.max(), eb.max(), ref.abs().max()))
    print("  pair err by (b, z):", [["%.1e" % e[bi, z].max() for z in range(od[0])] for bi in range(B)])
    print("  bf16 err by (b, z):", [["%.1e" % eb[bi, z].max() for z in range(od[0])] for bi in range(B)])
