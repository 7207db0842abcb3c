"""ElasticDeform (common/data.py:313-351) at the CAE sample size, 3 label channels + 0 image channels: device pipeline
(sp_gaussian_filter3d + sp_map_coordinates_linear; host RandomState noise or device noise) against scipy on the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stroke_prediction_amd.common import data as D
from oracle import transforms as T
rs = np.random.RandomState(0)
s = {"case_id": 1, "clinical_idx": 0, "images": rs.rand(128, 128, 28, 2).astype(np.float32),
     "labels": (rs.rand(128, 128, 28, 3) > 0.5).astype(np.float32), "clinical": rs.rand(1, 1, 1, 5).astype(np.float32)}
t0 = time.perf_counter()
T.elastic_deform({k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in s.items()}, 100, 4, False, np.random.RandomState(1))
cpu = time.perf_counter() - t0
for noise in (False, True):
    ed = D.ElasticDeform(100, 4, device_noise=noise)
    dev = D.to_device(s)
    ed(dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ed(dev)
    torch.cuda.synchronize()
    print("ElasticDeform 3 x 128x128x28, %s noise: %.2f ms per sample (scipy on the host: %.1f ms)" % ("device" if noise else "host RandomState", (time.perf_counter() - t0) * 100, cpu * 1e3))
