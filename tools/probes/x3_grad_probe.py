"""probe: parameter-gradient distance from the fp32 CPU oracle by precision mode and volume size (argv: sizes...)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import stroke_prediction_amd
from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Unet3D import Unet3D
import stroke_prediction_amd.common.dto.UnetDto as U
CH = [2, 16, 32, 64, 32, 16, 32, 2]
DEV = "cuda:0"
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for size in [int(v) for v in sys.argv[1:]]:
    seed = 11
    x, y = W.unet_inputs(2, (size,) * 3, seed)
    sd = W.make_state_dict(W.unet_spec(CH), seed)
    names = nets.trainable(sd)
    for k in names:
        sd[k].requires_grad_(True)
    seg = nets.unet_forward(sd, x, training=True)
    loss = nets.unet_loss(seg, y)
    gref = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    for mode in ("f32", "f16x3", "bf16x3", "f16", "bf16"):
        m = Unet3D(CH, dtype=mode)
        m.load_state_dict(W.make_state_dict(W.unet_spec(CH), seed))
        m = m.to(DEV).train()
        dto = m(U.init_dto(x.to(DEV), y[:, 0:1].to(DEV), y[:, 1:2].to(DEV)))
        l = nets.unet_loss(torch.cat((dto.outputs.core, dto.outputs.penu), 1), y.to(DEV))
        l.backward()
        if size <= 60:
            sref = seg.detach().double(); sg = torch.cat((dto.outputs.core, dto.outputs.penu), 1).detach().cpu().double()
            lg = lambda p: torch.log(p / (1 - p))
            print("      %-7s logits: max |dl| / max |l| = %.2e" % (mode, float((lg(sg) - lg(sref)).abs().max() / lg(sref).abs().max())))
        errs = {n: rel(p.grad.cpu(), gref[n]) for n, p in m.named_parameters()}
        big = {n: e for n, e in errs.items() if dict(m.named_parameters())[n].numel() > 64}
        small = {n: e for n, e in errs.items() if n not in big}
        wb = max(big, key=big.get); ws = max(small, key=small.get)
        srt = sorted(big.values())
        print("%3d^3 %-7s large tensors: worst %.3e (%s) median %.3e | <=64 elements: worst %.3e (%s)" % (size, mode, big[wb], wb, srt[len(srt) // 2], small[ws], ws), flush=True)
