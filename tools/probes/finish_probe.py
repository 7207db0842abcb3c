"""Isolated timing of sp_wgrad_finish_folded_scaled on the partial-block shapes of the fp8 4-scale step (B=2, 2x256^3):
(cout, cin, nparts) as runtime/f8.py:WgradRunnerF8._alloc picks them.  usage: python tools/probes/finish_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import stroke_prediction_amd  # noqa: F401,E402
from stroke_prediction_amd.runtime import lib as L, ops as O  # noqa: E402

DEV = "cuda"
SHAPES = [(32, 32, 512), (32, 96, 168), (64, 192, 40), (128, 384, 8), (64, 64, 128), (64, 32, 256), (128, 128, 32), (128, 64, 64),
          (256, 256, 8), (256, 128, 16)]


def main():
    tot = 0.0
    for cout, cin, nparts in SHAPES:
        acc = torch.randn(nparts, 27, cout, cin, device=DEV)
        scale = torch.rand(cin, device=DEV) + 0.5
        shift = torch.randn(cin, device=DEV) * 0.1
        dbias = torch.randn(L.SP_REDUCE_ROWS, cout, device=DEV).double()
        w = torch.randn(cout, cin, 27, device=DEV)
        dw = torch.zeros(cout, cin, 27, device=DEV)
        db = torch.zeros(cout, device=DEV)
        bn = torch.zeros(64, cin, 2, dtype=torch.float64, device=DEV)
        tapsrc = torch.arange(27, dtype=torch.int32, device=DEV)

        def run():
            L.call("sp_wgrad_finish_folded_scaled", O.ptr(acc), nparts, O.ptr(tapsrc), 27, cout, cin, cout, cin, cin * 27, 27, O.ptr(scale),
                   O.ptr(shift), O.ptr(dbias), O.ptr(dw), O.ptr(db), O.ptr(w), O.ptr(bn), 64, 0, cout, 0.5, O.stream())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50.0
        mb = acc.numel() * 4 / 1e6
        tot += us
        print("%4d->%-4d nparts %4d  %7.1f MB  %8.1f us  %7.1f GB/s" % (cin, cout, nparts, mb, us, mb / us * 1e3), flush=True)
    print("total %.1f us" % tot)


if __name__ == "__main__":
    main()
