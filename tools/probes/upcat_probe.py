"""Isolated timing of the fp8 mode's concatenation kernel (sp_upsample2_crop_cat_fwd_q8, 16-bit output not stored) on the three
concat shapes of the 4-scale step at 2 x 2 x 256^3.  usage: python tools/probes/upcat_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stroke_prediction_amd  # noqa: F401,E402
from stroke_prediction_amd.runtime import lib as L, ops as O, f8 as F8  # noqa: E402

DEV = "cuda"
# (low channels, low dims, skip channels, skip dims)
SHAPES = [(256, 24, 128, 57), (128, 44, 64, 122), (64, 84, 32, 252)]


def main():
    for cu, d, cs, dsk in SHAPES:
        B = 2
        low = torch.randn(B, d, d, d, cu, device=DEV).bfloat16()
        skip = torch.randn(B, dsk, dsk, dsk, cs, device=DEV).bfloat16()
        cat = torch.empty(B, 2 * d, 2 * d, 2 * d, cu + cs, dtype=torch.bfloat16, device=DEV)
        x8 = F8.alloc_f8(B, (2 * d,) * 3, cu + cs, DEV)
        stats = torch.zeros(L.SP_REDUCE_ROWS, cu + cs, 2, dtype=torch.float64, device=DEV)
        for store in (False, True):
            def run():
                O.upsample2_crop_cat_fwd(low, skip, cat, L.SP_BF16, stats, planar=True, q8=(x8, F8.E4M3, 1.0), store=store)
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
            nv = B * (2 * d) ** 3
            mb = (low.numel() * 2 + nv * cs * 2 + nv * (cu + cs) * (3 if store else 1)) / 1e6
            print("up %3d @%3d + skip %3d @%3d  store16=%d  %8.1f us  %7.1f MB algorithmic  %6.1f GB/s" % (cu, d, cs, dsk, store, us, mb, mb / us * 1e3), flush=True)
        print("checksum", int(x8.view(torch.uint8).long().sum()), float(stats.sum()))


if __name__ == "__main__":
    main()
