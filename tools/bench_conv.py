"""Per-layer kernel timing of the U-Net convolutions (fwd / dgrad / wgrad) at the BASELINE configs[1] shapes
(B=4, 128^3): HIP-event timed, prints us and TFLOP/s per launch.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L, ops as O, plan as P

DEV = "cuda:0"
B = int(os.environ.get("B", "4"))
dt = L.SP_F32 if os.environ.get("DT", "bf16") == "f32" else L.SP_BF16
which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
LAYERS = [("b1c1", 2, 16, 128), ("b1c2", 16, 16, 126), ("b2c1", 16, 32, 62), ("b2c2", 32, 32, 60), ("b3c1", 32, 64, 29),
          ("b3c2", 64, 64, 27), ("b4c1", 96, 32, 50), ("b4c2", 32, 32, 48), ("b5c1", 48, 16, 92), ("b5c2", 16, 16, 90)]
if os.environ.get("ONLY"):
    LAYERS = [l for l in LAYERS if l[0] in os.environ["ONLY"].split(",")]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = {}
for name, ci, co, d in LAYERS:
    dims = (d, d, d)
    cpi, cpo = O.cpad(ci), O.cpad(co)
    x = torch.randn((B,) + dims + (cpi,), device=DEV).to(O.TORCH_DT[dt])
    w = torch.randn(co, ci, 3, 3, 3, device=DEV) * 0.05
    sc, sh = torch.rand(cpi, device=DEV) + 0.5, torch.randn(cpi, device=DEV) * 0.1
    op = P.conv_fwd_op(ci, co, 3, 1, 0, dims, cpi, cpo, dt)
    gf = op.flops(B) / 1e9
    line = "%-5s %3d->%-3d @%3d  %6.1f GF |" % (name, ci, co, d, gf)
    y = O.alloc_cl(B, op.y_dims, cpo, dt, DEV)
    if "fwd" in which:
        r = O.ConvRunner(op, DEV, zm_batch=B if os.environ.get("ZM", "1") == "1" else None)
        st = torch.zeros(64, cpo, 2, dtype=torch.float64, device=DEV)
        if dt == L.SP_BF16 and os.environ.get("FOLD", "1") == "1":
            r.prep(w, torch.zeros(co, device=DEV), sc, sh)
            t = timeit(lambda: r.run(x, y, B, None, None, L.ACT_LEAKY, 0.01, st, stats_nrep=64))
        else:
            r.prep(w, torch.zeros(co, device=DEV))
            t = timeit(lambda: r.run(x, y, B, sc, sh, L.ACT_LEAKY, 0.01, st, stats_nrep=64))
        line += " fwd%s %7.1f us %6.1f TF/s (MT%d NT%d g%d lds%dK) |" % ("[zm]" if r.uses_zm() else "    ", t, gf / t * 1e3, op.subs[0].tile["MT"], op.nt, op.subs[0].tile["ngroups"], op.subs[0].tile["lds_bytes"] // 1024)
        tot["fwd"] = tot.get("fwd", 0) + t
    dz = torch.randn_like(y)
    if "dgrad" in which:
        dop = P.conv_dgrad_op(ci, co, 3, 1, 0, dims, cpo, cpi, dt)
        dr = O.ConvRunner(dop, DEV, zm_batch=B if os.environ.get("ZM", "1") == "1" else None); dr.prep(w)
        g = O.alloc_cl(B, dims, cpi, dt, DEV)
        t = timeit(lambda: dr.run(dz, g, B))
        line += " dgrad%s %7.1f us %6.1f TF/s |" % ("[zm]" if dr.uses_zm() else "    ", t, gf / t * 1e3)
        tot["dgrad"] = tot.get("dgrad", 0) + t
    if "wgrad" in which:
        wg = O.WgradRunner(ci, co, 3, 1, 0, dims, op.y_dims, cpi, cpo, ci * 27, 27, dt, DEV)
        dw = torch.zeros_like(w)
        dbs = torch.zeros(cpo, dtype=torch.float64, device=DEV)
        t = timeit(lambda: wg.run(x, dz, B, dw, sc, sh, dbias_sums=dbs))
        line += " wgrad %7.1f us %6.1f TF/s" % (t, gf / t * 1e3)
        tot["wgrad"] = tot.get("wgrad", 0) + t
    print(line, flush=True)
print("totals us:", {k: round(v, 1) for k, v in tot.items()})
