"""Loss trajectories of the 4-scale U-Net on one fixed synthetic batch per precision mode (does the fp8 mode train?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stroke_prediction_amd  # noqa
from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Unet3D import LargeUnet3D
import stroke_prediction_amd.common.dto.UnetDto as U
from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
from stroke_prediction_amd.runtime import f8 as F8
CH4 = [2, 32, 64, 128, 256, 128, 64, 32, 32, 2]
DEV = "cuda:0"
size = (int(sys.argv[1]),) * 3
nsteps = int(sys.argv[2])
F8.F8_MIN_PLANES = 8
seed = 5
torch.manual_seed(seed)
x = torch.randn((2, 2) + size, device=DEV)
# a learnable target: blobs that depend on the input (thresholded smoothed channel difference)
for mode in sys.argv[3:]:
    model = LargeUnet3D(CH4, dtype=mode)
    model.load_state_dict(W.make_state_dict(W.unet_spec(CH4), seed))
    model = model.to(DEV).train()
    out = model.output_size(size)
    c = [(s - o) // 2 for s, o in zip(size, out)]
    sm = torch.nn.functional.avg_pool3d(x, 5, 1, 2)[:, :, c[0]:c[0] + out[0], c[1]:c[1] + out[1], c[2]:c[2] + out[2]]
    y = torch.stack(((sm[:, 0] > 0.15), (sm[:, 1] - sm[:, 0] > 0.1)), 1).float()
    opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    attach_flat_grads(model)
    losses = []
    for step in range(nsteps):
        dto = model(U.init_dto(x, y[:, 0:1], y[:, 1:2]))
        loss = nets.unet_loss(torch.cat((dto.outputs.core, dto.outputs.penu), 1), y)
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    print(mode, " ".join("%.4f" % l for l in losses[::max(1, nsteps // 12)]), "final %.4f" % losses[-1])
