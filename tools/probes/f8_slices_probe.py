"""Probe: does the fp8 data gradient of a 96-channel input (three 32-channel slices written into 192-byte rows) lose time to its
partial-row stores?  Times it against three dense 32 -> 32 data gradients of the same volume."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L, ops as O, plan as P, f8 as F8
DEV = "cuda:0"
B, dims = 2, (168, 168, 168)
od = tuple(d - 2 for d in dims)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin in (96, 32, 64):
    cout = 32
    w = (torch.randn(cout, cin, 3, 3, 3) / math.sqrt(27 * cin)).to(DEV)
    dop = P.conv_dgrad_op(cin, cout, 3, 1, 0, dims, cout, cin, L.SP_BF16)
    run = F8.ConvRunnerF8(dop, DEV, B, F8.E5M2)
    dz8 = torch.randint(0, 100, (cout // 16, B) + od + (16,), dtype=torch.uint8, device=DEV)
    run.prep(w, out_scale=1.0)
    g = torch.empty((B,) + dims + (cin,), dtype=torch.bfloat16, device=DEV)
    t = timed(lambda: run.run(dz8, g))
    fl = 2 * 27 * cin * cout * B * od[0] * od[1] * od[2]
    print("dgrad of %d->%d @%s: %d slices, %.1f us, %.0f TFLOP/s" % (cin, cout, dims, len(run.slices), t, fl / t / 1e6))
