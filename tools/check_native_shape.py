"""U-Net train step at the reference's native patch 68 x 104 x 104 in bf16 and f32 mode: finite outputs,
the two modes agree within bf16 tolerance, gradients finite.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stroke_prediction_amd  # noqa
from stroke_prediction_amd.common.model.Unet3D import Unet3D
import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil
from stroke_prediction_amd.common.metrics import BatchDiceLoss
DEV = "cuda:0"
CH = [2, 16, 32, 64, 32, 16, 32, 2]
torch.manual_seed(3)
size = tuple(int(v) for v in (sys.argv[1:4] or (68, 104, 104)))
B = 2
m16 = Unet3D(CH, dtype="bf16").to(DEV).train()
m32 = Unet3D(CH, dtype="f32").to(DEV).train()
m32.load_state_dict(m16.state_dict())
x = torch.randn((B, 2) + size, device=DEV)
out = m16.output_size(size)
y = (torch.rand((B, 2) + tuple(out), device=DEV) > 0.7).float()
crit = BatchDiceLoss([1.0])
res = {}
for tag, m in (("bf16", m16), ("f32", m32)):
    dto = m(UnetDtoUtil.init_dto(x, y[:, 0:1], y[:, 1:2]))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    loss = (crit(dto.outputs.core, y[:, 0:1]) + crit(dto.outputs.penu, y[:, 1:2])) / 2
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert torch.isfinite(seg).all() and torch.isfinite(g).all(), tag
    res[tag] = (seg.detach(), float(loss), g)
    print(tag, "out", tuple(seg.shape), "loss %.6f" % float(loss), "|g| %.4e" % float(g.norm()))
d = (res["bf16"][0] - res["f32"][0]).abs().max().item()
cos = torch.nn.functional.cosine_similarity(res["bf16"][2], res["f32"][2], dim=0).item()
print("max |p16 - p32| = %.3e, loss diff %.2e, grad cosine %.4f" % (d, abs(res["bf16"][1] - res["f32"][1]), cos))
assert d < 3e-2 and cos > 0.9
print("ok")
