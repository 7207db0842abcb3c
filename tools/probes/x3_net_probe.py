"""probe: bf16x3 forward / backward at the headline shape, synchronising after every step (argv: B size nsteps)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import stroke_prediction_amd
from oracle import nets
from stroke_prediction_amd.common.model.Unet3D import Unet3D
import stroke_prediction_amd.common.dto.UnetDto as U
from stroke_prediction_amd.runtime import ops as O, lib as L
B, size, nsteps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
DEV = "cuda:0"
torch.manual_seed(0)
m = Unet3D([2, 16, 32, 64, 32, 16, 32, 2], dtype="bf16x3").to(DEV).train()
out = m.output_size((size,) * 3)
x = torch.randn(B, 2, size, size, size, device=DEV)
y = (torch.rand((B, 2) + tuple(out), device=DEV) > 0.7).float()
_orig = L.call
def traced(name, *a):
    _orig(name, *a)
    torch.cuda.synchronize()
    print("ok", name, flush=True)
if os.environ.get("TRACE"):
    L.call = traced
for s in range(nsteps):
    dto = m(U.init_dto(x, y[:, 0:1], y[:, 1:2]))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    torch.cuda.synchronize(); print("fwd ok", s, float(seg.mean()), flush=True)
    loss = nets.unet_loss(seg, y)
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize(); print("bwd ok", s, float(loss), flush=True)
if os.environ.get("GRAPH"):
    mode = os.environ["GRAPH"]
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        dto = m(U.init_dto(x, y[:, 0:1], y[:, 1:2]))
        seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
        if mode == "fb":
            loss = nets.unet_loss(seg, y)
            m.zero_grad()
            loss.backward()
    print("captured", mode, flush=True)
    for s in range(3):
        gr.replay()
        torch.cuda.synchronize(); print("replay ok", s, float(seg.mean()), flush=True)
