"""Isolated timing of sp_upsample2_act_bwd_q8 on the three up-path shapes of the 4-scale step at 2 x 2 x 256^3 (bf16 in,
e5m2 copy out).  usage: python tools/probes/upbwd_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stroke_prediction_amd  # noqa: F401,E402
from stroke_prediction_amd.runtime import lib as L, ops as O, f8 as F8  # noqa: E402

DEV = "cuda"
SHAPES = [(256, 24, 128), (128, 44, 64), (64, 84, 32), (64, 84, 0)]      # (low channels, low extent, skip channels; 0: a DENSE gradient tensor); B = 2
if len(sys.argv) > 1 and sys.argv[1] == "headline":          # the 3-scale step at 4 x 2 x 128^3, and the same with a DENSE gradient tensor
    SHAPES = [(32, 46, 16), (32, 46, 0), (64, 23, 32), (64, 23, 0)]


def main():
    torch.manual_seed(0)
    for cu, d, cs in SHAPES:
        B = 4 if len(sys.argv) > 1 and sys.argv[1] == "headline" else 2
        y = torch.randn(B, d, d, d, cu, device=DEV).bfloat16()
        g = (torch.randn(B, 2 * d, 2 * d, 2 * d, cu + cs, device=DEV) * 1e-6).bfloat16()
        coef = torch.randn(3, cu + cs, device=DEV) * 0.5
        dz = torch.empty_like(y)
        dz8 = F8.alloc_f8(B, (d,) * 3, cu, DEV)
        db = O.reduce_rows(cu, 1, DEV)

        def run():
            O.upsample2_act_bwd(y, None, g, coef, L.SP_BF16, L.ACT_LEAKY, 0.01, dz, db, q8=(dz8, F8.E5M2, 2.0 ** 20))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        mb = (g.numel() // (cu + cs) * cu * 2 + y.numel() * 2 * 2 + y.numel()) / 1e6
        print("low %3d ch @%3d (g pitch %3d)  %8.1f us  %7.1f MB algorithmic  %6.1f GB/s  checksum %d %.6e" % (
            cu, d, cu + cs, us, mb, mb / us * 1e3, int(dz8.long().sum()), float(dz.float().abs().sum())), flush=True)


if __name__ == "__main__":
    main()
