import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler; faulthandler.enable()
import torch
import stroke_prediction_amd  # noqa
from oracle import weights as W
from stroke_prediction_amd.common.model.Cae3D import Cae3D, Enc3D, Dec3D
from stroke_prediction_amd.common.metrics import BatchDiceLoss
from stroke_prediction_amd.learner.CaeReconstructionLearner import CaeReconstructionLearner
from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
DEV = "cuda:0"
ch = [1, 16, 24, 32, 100, 200, 1]
mode = sys.argv[1]
cae = Cae3D(Enc3D(64, 28, ch, 5, 1.0), Dec3D(64, 28, ch, 5, 1.0)).to(DEV).train()
opt = FusedAdam(list(cae.parameters()), lr=1e-3, capturable=True)
attach_flat_grads(cae)
class Loader(list):
    batch_size = 2
learner = CaeReconstructionLearner(Loader(), None, cae, opt, None, 1, None, "/tmp/_dbg", BatchDiceLoss([1.0]), verbose=False, graph=False, batch_metrics=False)
labels, clinical = W.cae_inputs(2, 28, 64, 3)
batch = {"case_id": [0, 1], "images": None, "labels": labels.to(DEV), "clinical": clinical.to(DEV)}
for _ in range(2):
    learner._optimise(batch, 30)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    if mode == "fwd":
        with torch.no_grad():
            dto = learner.inference_step(batch)
    elif mode == "fwdgrad":
        dto = learner.inference_step(batch)
        loss = learner.loss_step(dto, 30)
    else:
        learner._optimise(batch, 30)
print("captured", mode, flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", mode, flush=True)
