"""Run the 3-step fixture trajectory several times; report losses and the largest per-parameter differences between runs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import importlib
from oracle import weights as W, nets
pkg = importlib.import_module("stroke_prediction_amd")
from stroke_prediction_amd.common.model.Unet3D import Unet3D
import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil
from stroke_prediction_amd.optim import FusedAdam
DEV = torch.device("cuda:0")
CH = [2, 16, 32, 64, 32, 16, 32, 2]
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", sys.argv[1] if len(sys.argv) > 1 else "unet_44.npz"))
CH = [int(c) for c in fx["channels"]] if "channels" in fx else CH
seed = int(fx["seed"]); size = tuple(int(s) for s in np.atleast_1d(fx["size"])); size = size * 3 if len(size) == 1 else size
x, y = W.unet_inputs(2, size, seed)
xd, yd = x.to(DEV), y.to(DEV)
print("fixture losses", [float(fx["loss/%d" % s]) for s in range(3)])
runs = []
for r in range(6):
    model = Unet3D(CH, dtype="f32"); model.load_state_dict(W.make_state_dict(W.unet_spec(CH), seed)); model = model.to(DEV).train()
    opt = (FusedAdam if r < 4 else torch.optim.Adam)(model.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    losses, snaps = [], []
    for step in range(3):
        dto = model(UnetDtoUtil.init_dto(xd, yd[:, 0:1], yd[:, 1:2]))
        loss = nets.unet_loss(torch.cat((dto.outputs.core, dto.outputs.penu), 1), yd)
        losses.append(loss.item())
        opt.zero_grad(); loss.backward()
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        opt.step()
        snaps.append((g, {n: p.detach().clone() for n, p in model.named_parameters()}))
    print(r, type(opt).__name__, ["%.7f" % l for l in losses])
    runs.append((losses, snaps))
base = runs[0]
for r in range(1, len(runs)):
    for step in range(3):
        worst = []
        for n in base[1][step][1]:
            d = (base[1][step][1][n] - runs[r][1][step][1][n]).abs().max().item()
            gd = (base[1][step][0][n] - runs[r][1][step][0][n]).abs().max().item()
            worst.append((d, gd, n))
        worst.sort(reverse=True)
        print("run", r, "step", step, "max param diff", ["%s p%.2e g%.2e" % (n, d, gd) for d, gd, n in worst[:3]])
for k in fx.files:
    if k.startswith("phead3/"):
        n = k[7:]
        d = np.abs(fx[k] - base[1][2][1][n].reshape(-1)[:8].cpu().numpy()).max()
        if d > 5e-4: print("vs fixture (first 8 elements)", n, d)
