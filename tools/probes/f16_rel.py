import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import stroke_prediction_amd
from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Unet3D import Unet3D
import stroke_prediction_amd.common.dto.UnetDto as U
CH = [2, 16, 32, 64, 32, 16, 32, 2]
seed, size = int(os.environ.get("SEED", "11")), (44, 44, 44)
x, y = W.unet_inputs(2, size, seed)
sd = W.make_state_dict(W.unet_spec(CH), seed)
names = nets.trainable(sd)
for k in names: sd[k].requires_grad_(True)
seg = nets.unet_forward(sd, x, training=True)
loss = nets.unet_loss(seg, y)
ref = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
for mode in sys.argv[1:]:
    m = Unet3D(CH, dtype=mode); m.load_state_dict(W.make_state_dict(W.unet_spec(CH), seed)); m = m.cuda().train()
    dto = m(U.init_dto(x.cuda(), y[:, 0:1].cuda(), y[:, 1:2].cuda()))
    s = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    nets.unet_loss(s, y.cuda()).backward()
    print(mode, "seg err", float((s.detach().cpu() - seg.detach()).abs().max()))
    for k, p in m.named_parameters():
        a, b = p.grad.detach().cpu().double(), ref[k].double()
        if b.numel() > 64 and (k.startswith("block1.bn_conv_relu_2x.1.w") or k.startswith("block5.bn_conv_relu_2x.4.w") or k.startswith("classify.0.w")):
            print("  %-36s rel %.4f" % (k, float((a - b).norm() / b.norm())))
