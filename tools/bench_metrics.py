"""batch metrics of one training step (metrics.py:31-62) -- Dice / precision / sensitivity /
specificity from one HIP reduction on the device, against the reference's route (device -> host copy + numpy counts),
and the cost of the Hausdorff / ASSD distance transforms that stay on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stroke_prediction_amd.common import metrics as M
g = torch.Generator(device="cuda").manual_seed(0)
res = torch.rand((4, 1, 88, 88, 88), generator=g, device="cuda")
tgt = (torch.rand((4, 1, 88, 88, 88), generator=g, device="cuda") > 0.7).float()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
dev = t(lambda: M.binary_measures_torch(res, tgt, True, distances=False), 20)
host = t(lambda: M.binary_measures_numpy(res.cpu().numpy(), tgt.cpu().numpy(), distances=False), 5)
dist = t(lambda: M.binary_measures_torch(res, tgt, True, distances=True), 20)
a, b = M.binary_measures_torch(res, tgt, True, distances=False), M.binary_measures_numpy(res.cpu().numpy(), tgt.cpu().numpy(), distances=False)
assert abs(a.dc - b.dc) < 1e-9 and abs(a.precision - b.precision) < 1e-9 and abs(a.sensitivity - b.sensitivity) < 1e-9
M.DEVICE_DISTANCES = False
dist_host = t(lambda: M.binary_measures_torch(res, tgt, True, distances=True), 1)
M.DEVICE_DISTANCES = True
from stroke_prediction_amd.runtime import lib as L, ops as O
dims = torch.tensor(list(res.shape), dtype=torch.int32); ws = torch.empty(4 * res.numel(), device="cuda"); out = torch.zeros(6, dtype=torch.float64, device="cuda")
kern = t(lambda: L.call("sp_surface_distances", O.ptr(res), O.ptr(tgt), 0.5, 5, dims.data_ptr(), O.ptr(ws), O.ptr(out), O.stream()), 10)
print("batch metrics 4x1x88^3: counts on the device %.3f ms | host copy + numpy counts %.1f ms | + HD/ASSD: device %.1f ms "
      "(sp_surface_distances alone %.2f ms), scipy on the host %.0f ms" % (dev, host, dist, kern, dist_host))
