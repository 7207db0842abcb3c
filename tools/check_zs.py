"""z-marching conv variant (SP_CONV_ZS) against the standard DMA kernel: identical results expected (same K order)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stroke_prediction_amd.runtime import lib as L, ops as O, plan as P
dev, dt = "cuda:0", L.SP_BF16
torch.manual_seed(0)
def run(op, x, w, b, zs, stats, zr=False):
    O.USE_ZS = zs
    O.USE_ZR = zr
    r = O.ConvRunner(op, dev); r.prep(w, b)
    y = O.alloc_cl(x.shape[0], op.y_dims, CPO, dt, dev); y.fill_(7.0)
    st = torch.zeros(64, CPO, 2, dtype=torch.float64, device=dev) if stats else None
    r.run(x, y, x.shape[0], None, None, L.ACT_LEAKY, 0.01, st, stats_nrep=64)
    torch.cuda.synchronize()
    return y, (st.sum(0) if stats else None)
CO = int(os.environ.get("CO", "16"))
CPO = -(-CO // 16) * 16
for dims, B in (((40, 70, 50), 2), ((37, 45, 17), 1), ((int(os.environ.get("D", "126")),) * 3, 4)):
    for mode in ("fwd", "dgrad"):
        if mode == "fwd":
            op = P.conv_fwd_op(16, CO, 3, 1, 0, dims, 16, CPO, dt)
            xin = dims
        else:
            op = P.conv_dgrad_op(CO, 16, 3, 1, 0, dims, 16, CPO, dt)      # gradient wrt a CO-channel input, dz has 16
            xin = tuple(d - 2 for d in dims)
        if getattr(op.subs[0], "ktab_zs", None) is None:
            print(dims, mode, "not eligible"); continue
        x = torch.randn((B,) + xin + (16,), device=dev).bfloat16()
        w = (torch.randn(CO, 16, 3, 3, 3, device=dev) if mode == "fwd" else torch.randn(16, CO, 3, 3, 3, device=dev)) * 0.1
        b = torch.randn(CO, device=dev) * 0.1 if mode == "fwd" else None
        y0, s0 = run(op, x, w, b, False, mode == "fwd")
        y1, s1 = run(op, x, w, b, True, mode == "fwd")
        ok = torch.equal(y0, y1)
        print(dims, B, mode, "equal" if ok else "MISMATCH max %.3e" % float((y0.float() - y1.float()).abs().max()),
              "" if s0 is None else "stats diff %.2e" % float((s0 - s1).abs().max()))
        if getattr(op.subs[0], "ktab_zr", None) is not None:      # row-reuse order: other summation order, bf16 rounding
            y2, s2 = run(op, x, w, b, True, mode == "fwd", zr=True)
            d = (y0.float() - y2.float()).abs()
            tol = 2.0 ** -7 * y0.float().abs().clamp_min(1.0)
            print(dims, B, mode, "zr max diff %.3e, beyond 1 bf16 ulp: %d of %d" % (float(d.max()), int((d > tol).sum()), d.numel()),
                  "" if s0 is None else "stats rel diff %.2e" % float(((s0 - s2).abs() / s0.abs().clamp_min(1.0)).max()))
        if dims[0] >= 60:
            for zs, zr in ((False, False), (True, False), (True, True)):
                O.USE_ZS, O.USE_ZR = zs, zr
                r = O.ConvRunner(op, dev); r.prep(w, b)
                y = O.alloc_cl(B, op.y_dims, CPO, dt, dev)
                st = torch.zeros(64, CPO, 2, dtype=torch.float64, device=dev) if mode == "fwd" else None
                f = lambda: r.run(x, y, B, None, None, L.ACT_LEAKY, 0.01, st, stats_nrep=64)
                for _ in range(3): f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                print("   %s zs=%d zr=%d: %.1f us" % (mode, zs, zr, e0.elapsed_time(e1) * 100))
