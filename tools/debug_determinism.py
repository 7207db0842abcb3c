"""Diagnostic: two fresh models, same weights and input, single process: gradients must agree to atomics-order noise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stroke_prediction_amd  # noqa
from oracle import weights as W
from stroke_prediction_amd.common.model.Unet3D import Unet3D
from stroke_prediction_amd.common.metrics import BatchDiceLoss
import stroke_prediction_amd.common.dto.UnetDto as UD
from stroke_prediction_amd.optim import attach_flat_grads
from stroke_prediction_amd.runtime import ops as O
CH = [2, 16, 32, 64, 32, 16, 32, 2]
dev = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
if os.environ.get("NO_PARTS"):
    O.WGRAD_PARTS = False
x, y = W.unet_inputs(4, (44, 44, 44), 31)
crit = BatchDiceLoss([1.0])
junk = []
def run():
    m = Unet3D(CH, dtype=dtype)
    m.load_state_dict(W.make_state_dict(W.unet_spec(CH), 31))
    m = m.to(dev).train()
    attach_flat_grads(m)
    dto = m(UD.init_dto(x.to(dev), y[:, 0:1].to(dev), y[:, 1:2].to(dev)))
    loss = (crit(dto.outputs.core, dto.given_variables.core) + crit(dto.outputs.penu, dto.given_variables.penu)) / 2
    loss.backward()
    torch.cuda.synchronize()
    junk.append(torch.full((1 << 24,), 3.0e4, device=dev))       # perturb the allocator between runs
    eng = list(m._engines.values())[0]
    extra = {"seg": torch.cat((dto.outputs.core, dto.outputs.penu), 1).detach().cpu().clone(),
             "c52.y": eng.c52.y.float().cpu().clone(), "c52.dz": eng.c52.dz.float().cpu().clone(),
             "c52.dbias_sums": eng.c52.dbias_sums.cpu().clone(), "c51.dz": eng.c51.dz.float().cpu().clone(),
             "hpart": getattr(eng, "_hpart", torch.zeros(1)).cpu().clone()}
    for k, v in extra.items():
        print("   %-16s finite=%s norm=%.6e" % (k, bool(torch.isfinite(v).all()), float(v.double().norm())))
    return {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}, extra
a, ea = run()
b, eb = run()
for k in ("c52.y", "c52.dz", "c51.dz"):
    d = (ea[k] - eb[k]).abs()
    idx = torch.nonzero(d > 1e-5 * ea[k].abs().max())
    print(k, "max abs diff %.3e of max %.3e; elements differing: %d of %d" % (float(d.max()), float(ea[k].abs().max()), idx.shape[0], d.numel()))
    for i in idx[:8]:
        i = tuple(int(t) for t in i)
        print("    at", i, "a=%.6e b=%.6e   y_a=%.6e y_b=%.6e" % (float(ea[k][i]), float(eb[k][i]), float(ea["c52.y"][i]) if k != "c51.dz" else 0, float(eb["c52.y"][i]) if k != "c51.dz" else 0))
worst = sorted(((float((a[n] - b[n]).norm() / (a[n].norm() + 1e-30)), n) for n in a), reverse=True)[:6]
for n in ("classify.0.weight", "classify.0.bias", "classify.2.weight", "classify.2.bias"):
    print("%-45s %.3e" % (n, float((a[n] - b[n]).norm() / (a[n].norm() + 1e-30))))
for d, n in worst:
    print("%-45s %.3e" % (n, d))
