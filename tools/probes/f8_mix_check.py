"""Robustness probe of the fp8 mode: which layers run in fp8 depends on thresholds and volume; every mix must train like the bf16
mode.  Prints loss / gradient-norm ratios of one training step for a few input sizes and thresholds."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import stroke_prediction_amd  # noqa
from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Unet3D import LargeUnet3D
import stroke_prediction_amd.common.dto.UnetDto as U
from stroke_prediction_amd.runtime import f8 as F8
CH4 = [2, 32, 64, 128, 256, 128, 64, 32, 32, 2]
DEV = "cuda:0"


def step(dtype, size, seed=3):
    x, y = W.unet_inputs(2, size, seed, scales=4)
    model = LargeUnet3D(CH4, dtype=dtype)
    model.load_state_dict(W.make_state_dict(W.unet_spec(CH4), seed))
    model = model.to(DEV).train()
    dto = model(U.init_dto(x.to(DEV), y[:, 0:1].to(DEV), y[:, 1:2].to(DEV)))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    loss = nets.unet_loss(seg, y.to(DEV))
    loss.backward()
    eng = model._engine(x.to(DEV))
    kinds = [("F" if l.f8_fwd is not None else "-") + ("D" if l.f8_dgrad is not None else "-") + ("W" if l.f8_wgrad is not None else "-")
             + ("s" if isinstance(l.f8_fwd, F8.ConvRunnerF8Split) else "") for l in eng.layers]
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    return float(loss), g, seg.detach(), kinds


for size in [(92, 92, 92), (108, 100, 92), (124, 116, 132), (156, 156, 156)]:
    lb, gb, sb, _ = step("bf16", size)
    for mp in (128, 8, 4000, 10 ** 9):
        F8.F8_MIN_PLANES = mp
        lf, gf, sf, kinds = step("fp8", size)
        ok = torch.isfinite(gf).all() and torch.isfinite(sf).all()
        print("size %-16s min_planes %-10d loss bf16 %.5f fp8 %.5f  |g| ratio %.3f  seg max diff %.3e  finite %s  %s"
              % (size, mp, lb, lf, float(gf.norm() / gb.norm()), float((sf - sb).abs().max()), bool(ok), " ".join(kinds)))
