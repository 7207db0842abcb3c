"""Time one weight-gradient layer (ci co d [d_h d_w]) under the env knobs SP_WGRAD_CIB / SP_WGRAD_DMA_MAXCIT / SP_WGRAD_BLOCKS."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.runtime import lib as L, ops as O
ci, co, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 4
dims, od = (d,) * 3, (d - 2,) * 3
cpi, cpo = O.cpad(ci, 16), O.cpad(co, 16)
x = torch.randn((B,) + dims + (cpi,), device="cuda").bfloat16()
dz = torch.randn((B,) + od + (cpo,), device="cuda").bfloat16()
w = torch.zeros(co, ci, 3, 3, 3, device="cuda")
wg = O.WgradRunner(ci, co, 3, 1, 0, dims, od, cpi, cpo, ci * 27, 27, L.SP_BF16, "cuda")
fn = lambda: wg.run(x, dz, B, w)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 100
fl = 2.0 * B * od[0] * od[1] * od[2] * 27 * ci * co
print("wgrad %d->%d @%d dma=%d cib=%s nblocks=%d: %.1f us (incl. finish)  %.0f TFLOP/s" % (ci, co, d, wg.dma, os.environ.get("SP_WGRAD_CIB", "auto"), wg.args.nblocks, t, fl / t / 1e6))
