"""probe: bf16-mode parameter gradients against the bf16-EMULATING oracle (storage points rounded like the kernels' tensors)"""
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
for size, seed in (((44, 44, 44), 11), ((48, 48, 48), 12), ((60, 60, 60), 14)):
    x, y = W.unet_inputs(2, size, seed)
    sd = W.make_state_dict(W.unet_spec(CH), seed)
    names = nets.trainable(sd)
    for k in names:
        sd[k].requires_grad_(True)
    seg = nets.unet_forward(sd, x, training=True, q=nets.round_bf16)
    loss = nets.unet_loss(seg, y)
    gref = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    m = Unet3D(CH, dtype="bf16")
    m.load_state_dict(W.make_state_dict(W.unet_spec(CH), seed))
    m = m.to(DEV).train()
    dto = m(U.init_dto(x.to(DEV), y[:, 0:1].to(DEV), y[:, 1:2].to(DEV)))
    l = nets.unet_loss(torch.cat((dto.outputs.core, dto.outputs.penu), 1), y.to(DEV))
    l.backward()
    big = sorted((rel(p.grad.cpu(), gref[n]), n) for n, p in m.named_parameters() if p.numel() > 64)
    small = sorted((rel(p.grad.cpu(), gref[n]), n) for n, p in m.named_parameters() if p.numel() <= 64)
    print(size, "large: worst", big[-3:], "median %.3f" % big[len(big) // 2][0])
    print("        small: worst", small[-3:], "median %.3f" % small[len(small) // 2][0], flush=True)
