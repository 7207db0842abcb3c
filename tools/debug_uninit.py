"""Diagnostic: poison every torch.empty allocation with NaN and report which gradients / outputs pick it up."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import stroke_prediction_amd  # noqa
from oracle import weights as W
from stroke_prediction_amd.common.model.Unet3D import Unet3D
from stroke_prediction_amd.common.metrics import BatchDiceLoss
import stroke_prediction_amd.common.dto.UnetDto as UD
from stroke_prediction_amd.optim import attach_flat_grads
CH = [2, 16, 32, 64, 32, 16, 32, 2]
dev = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
x, y = W.unet_inputs(2, (44, 44, 44), 31)
crit = BatchDiceLoss([1.0])
m = Unet3D(CH, dtype=dtype)
m.load_state_dict(W.make_state_dict(W.unet_spec(CH), 31))
m = m.to(dev).train()
attach_flat_grads(m)
for step in range(2):
    dto = m(UD.init_dto(x.to(dev), y[:, 0:1].to(dev), y[:, 1:2].to(dev)))
    loss = (crit(dto.outputs.core, dto.given_variables.core) + crit(dto.outputs.penu, dto.given_variables.penu)) / 2
    loss.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in m.named_parameters() if not torch.isfinite(p.grad).all()]
    print("step", step, "loss", float(loss), "non-finite grads:", bad)
    m.zero_grad()
