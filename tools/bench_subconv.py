"""Time a transposed convolution's forward (parity classes in one launch or one by one: SP_CONV_MULTI=1/0).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L, ops as O, plan as P
DEV = "cuda:0"
B = 4
for cin, cout, k, s, dims in ((100, 32, 3, 2, (3, 12, 12)), (16, 16, 2, 2, (14, 62, 62)), (24, 24, 2, 2, (7, 29, 29))):
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    op = P.convT_fwd_op(cin, cout, k, s, 0, dims, cpi, cpo, L.SP_BF16)
    run = O.ConvRunner(op, DEV)
    w = torch.randn(cin, cout, k, k, k, device=DEV) * 0.05
    run.prep(w, torch.zeros(cout, device=DEV))
    x = torch.randn((B,) + dims + (cpi,), device=DEV).bfloat16()
    y = O.alloc_cl(B, op.y_dims, cpo, L.SP_BF16, DEV)
    sc, sh = torch.rand(cpi, device=DEV) + 0.5, torch.randn(cpi, device=DEV) * 0.1
    st = torch.zeros(64 * cpo * 2, dtype=torch.float64, device=DEV)
    for with_scale in (True, False):
        fn = lambda: run.run(x, y, B, sc if with_scale else None, sh if with_scale else None, L.ACT_ELU, 1.0, st, stats_nrep=64)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print("%d->%d k%d s%d @%s  scale=%s  %.1f us  (%d classes, tiles %s)" % (cin, cout, k, s, dims, with_scale, e0.elapsed_time(e1) * 100,
              len(op.subs), sorted({(sb.tile["MT"], sb.tile["dma"], sb.tile["steps_per_group"], sb.tile["ngroups"]) for sb in op.subs})))
