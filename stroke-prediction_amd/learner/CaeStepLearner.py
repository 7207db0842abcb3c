"""Learns the interpolation step of the frozen shape space (reference ``learner/CaeStepLearner.py:7-29``): the CAE is an
``Enc3DStep`` (frozen encoder stack + trainable 1x1x1 step layers on the clinical globals) and the frozen decoder;
``get_time_to_treatment`` hands the model no step (``None``) so that the encoder predicts it.

loss (``loss_step`` :15-21) = [ mean(|p-i|-(p-i)) + Dice(i, lesion) ] / 2 on the ground-truth reconstructions.
The gradient reaches the step layers through the frozen decoder's data gradient (``Cae3D._frozen_backward``) and the
torch-side latent interpolation; the frozen encoder's three passes record no autograd node."""
import torch

from common import metrics
from common.dto.CaeDto import CaeDto
from learner.CaeReconstructionLearner import CaeReconstructionLearner


class CaeStepLearner(CaeReconstructionLearner):
    FN_VIS_BASE = '_cae1step_'
    FNB_MARKS = '_cae1step'
    N_EPOCHS_ADAPT_BETA1 = 4

    def graph_key(self, epoch):
        return 0

    def loss_step(self, dto: CaeDto, epoch):
        rec = dto.reconstructions.gtruth
        diff_penu_fuct = rec.penu - rec.interpolation
        loss = metrics.batch_mean(torch.abs(diff_penu_fuct) - diff_penu_fuct)
        loss = loss + self._criterion(rec.interpolation, dto.given_variables.gtruth.lesion)
        return loss / 2

    def get_time_to_treatment(self, batch, global_variables, step):
        if step is None:
            return None              # Enc3DStep._get_step predicts it from the globals (Cae3D.py:137-141)
        normalization = self._get_normalization(batch)
        ttt = (step * torch.ones(global_variables.size(0), 1, device=normalization.device)) / normalization
        return ttt.reshape(-1, 1, 1, 1, 1)
