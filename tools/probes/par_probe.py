"""timing of the parity-class kernel (csrc/sp_conv_par.hip) on the CAE's two large class ops, with and without the statistics epilogues"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L, ops as O, plan as P
DEV = "cuda:0"

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for kind, cin, cout, k, s, pad, dims, B, gb in [("dgrad", 16, 24, 3, 2, (1, 1, 1), (28, 124, 124), 12, 4), ("convT", 16, 16, 2, 2, 0, (14, 62, 62), 16, 4)]:
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    if kind == "convT":
        op = P.convT_fwd_op(cin, cout, k, s, pad, dims, cpi, cpo, L.SP_BF16)
        w = torch.randn(cin, cout, k, k, k, device=DEV); src = torch.randn((B,) + tuple(dims) + (cpi,), device=DEV).bfloat16(); cd = cpo
    else:
        op = P.conv_dgrad_op(cin, cout, k, s, pad, dims, cpo, cpi, L.SP_BF16)
        out = tuple((dims[a] + 2 * pad[a] - k) // s + 1 for a in range(3))
        w = torch.randn(cout, cin, k, k, k, device=DEV); src = torch.randn((B,) + out + (cpo,), device=DEV).bfloat16(); cd = cpi
    y = O.alloc_cl(B, op.y_dims, cd, L.SP_BF16, DEV, zero=True)
    aux = torch.randn_like(y.float()).bfloat16()
    st = torch.zeros((B // gb) * 64 * cd * 2, dtype=torch.float64, device=DEV)
    for on in (True, False):
        O.USE_PAR = on
        run = O.ConvRunner(op, DEV)
        run.prep(w, None)
        t0 = timeit(lambda: run.run(src, y, B))
        t1 = timeit(lambda: run.run(src, y, B, stats=st, stats_nrep=64, group_batch=gb))
        t2 = timeit(lambda: run.run(src, y, B, stats=st, stats_nrep=64, stats_mode=1, aux=aux, group_batch=gb))
        print("%s %d->%d par=%d: plain %.1f us, +stats %.1f us, +bn-backward sums %.1f us   (out %.0f MB, in %.0f MB)" % (
            kind, cin, cout, on, t0, t1, t2, y.numel() * 2 / 1e6, src.numel() * 2 / 1e6), flush=True)
