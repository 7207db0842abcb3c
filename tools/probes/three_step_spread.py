import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import stroke_prediction_amd
from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Unet3D import Unet3D
import stroke_prediction_amd.common.dto.UnetDto as U
from stroke_prediction_amd.optim import FusedAdam
CH = [2, 16, 32, 64, 32, 16, 32, 2]; DEV = "cuda:0"
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for fname in ["unet_44.npz", "unet_48.npz", "unet_44x48x52.npz"]:
    fx = np.load(os.path.join(root, "tests/golden", fname))
    seed = int(fx["seed"]); size = tuple(int(s) for s in np.atleast_1d(fx["size"])); size = size * 3 if len(size) == 1 else size
    x, y = W.unet_inputs(2, size, seed); xd, yd = x.to(DEV), y.to(DEV)
    for rep in range(2):
        model = Unet3D(CH, dtype="f32"); model.load_state_dict(W.make_state_dict(W.unet_spec(CH), seed)); model = model.to(DEV).train()
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
        d = []
        for step in range(3):
            dto = model(U.init_dto(xd, yd[:, 0:1], yd[:, 1:2]))
            loss = nets.unet_loss(torch.cat((dto.outputs.core, dto.outputs.penu), 1), yd)
            d.append(abs(loss.item() - float(fx["loss/%d" % step])))
            opt.zero_grad(); loss.backward(); opt.step()
        print(fname, rep, ["%.2e" % v for v in d])
