"""Phase-2 learner (reference ``learner/CaePredictionLearner.py:10-57``): trains a NEW encoder on the U-Net segmentations
against the frozen shape CAE.

loss (``loss_step`` :42-57) = [ mean(|p-i|-(p-i)) + mean(|p-c|-(p-c)) over the decoded INPUT latents + Dice(i, lesion)
+ mean|z_gt.i - z_in.i| + mean|z_gt.c - z_in.c| + mean|z_gt.p - z_in.p| ] / 6.

What the accelerated path adds: the gradient runs through the frozen decoder as data-gradient convolutions and
BatchNorm-backward terms only (``Cae3D._frozen_backward``: no weight-gradient kernel, nothing written to the CAE's gradient
buffers) into the trainable encoder; the frozen CAE's own two calls record no autograd node at all.  As in the reference the
CAE stays in whatever mode ``run_training`` puts it (train mode: live BatchNorm statistics, running statistics keep moving)."""
import torch

import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common import metrics
from common.dto.CaeDto import CaeDto
from common.inference.CaeEncInference import CaeEncInference
from learner.Learner import Learner


class CaePredictionLearner(Learner, CaeEncInference):
    FN_VIS_BASE = '_cae2_'
    FNB_MARKS = '_cae2'
    N_EPOCHS_ADAPT_BETA1 = 4

    def __init__(self, dataloader_training, dataloader_validation, cae_model, enc_model, optimizer, scheduler, n_epochs,
                 path_previous_base, path_outputs_base, criterion, normalization_hours_penumbra=10, **learner_kw):
        # (the new encoder must exist before Learner.__init__ may call load_model)
        CaeEncInference.__init__(self, cae_model, enc_model, normalization_hours_penumbra)
        Learner.__init__(self, dataloader_training, dataloader_validation, cae_model, optimizer, scheduler, n_epochs,
                         path_previous_base, path_outputs_base, **learner_kw)
        self._model.freeze(True)
        self._criterion = criterion

    def load_model(self, cuda=True):
        Learner.load_model(self, cuda)
        enc = torch.load(self.path('load', self.FNB_MODEL, '_enc'), weights_only=False)
        self._new_enc = enc.cuda() if cuda else enc

    def save_model(self, suffix=''):
        Learner.save_model(self, suffix)
        if not self._is_rank0():
            return
        import copy
        torch.save(copy.deepcopy(self._new_enc).cpu(), self.path('save', self.FNB_MODEL, '_enc' + suffix))      # (the live encoder is not moved: Learner.save_model)

    def adapt_betas(self, epoch):
        pass

    def loss_step(self, dto: CaeDto, epoch):
        rec, lat_in, lat_gt = dto.reconstructions.inputs, dto.latents.inputs, dto.latents.gtruth
        diff_penu_fuct = rec.penu - rec.interpolation
        diff_penu_core = rec.penu - rec.core
        loss = metrics.batch_mean(torch.abs(diff_penu_fuct) - diff_penu_fuct)
        loss = loss + metrics.batch_mean(torch.abs(diff_penu_core) - diff_penu_core)
        loss = loss + self._criterion(rec.interpolation, dto.given_variables.gtruth.lesion)
        loss = loss + metrics.batch_mean(torch.abs(lat_gt.interpolation - lat_in.interpolation))
        loss = loss + metrics.batch_mean(torch.abs(lat_gt.core - lat_in.core))
        loss = loss + metrics.batch_mean(torch.abs(lat_gt.penu - lat_in.penu))
        return loss / 6

    def batch_metrics_step(self, dto: CaeDto, epoch):
        rec, gt = dto.reconstructions.gtruth, dto.given_variables.gtruth
        batch_metrics = MetricMeasuresDtoInit.init_dto()
        batch_metrics.lesion = metrics.binary_measures_torch(rec.interpolation, gt.lesion, self.is_cuda)
        batch_metrics.core = metrics.binary_measures_torch(rec.core, gt.core, self.is_cuda)
        batch_metrics.penu = metrics.binary_measures_torch(rec.penu, gt.penu, self.is_cuda)
        return batch_metrics

    def print_epoch(self, epoch, phase, epoch_metrics):
        f = lambda v: float('nan') if v is None else float(v)
        print('\nEpoch {}/{} {} loss: {:.3} - DC:{:.3}, HD:{:.3}, ASSD:{:.3}, DC core:{:.3}, DC penu.:{:.3}'.format(
            epoch + 1, self._n_epochs, phase, f(epoch_metrics.loss), f(epoch_metrics.lesion.dc),
            f(epoch_metrics.lesion.hd), f(epoch_metrics.lesion.assd), f(epoch_metrics.core.dc),
            f(epoch_metrics.penu.dc)), end=' ')

    def plot_epoch(self, plot, epochs):
        plot.plot(epochs, [dto.loss for dto in self._metric_dtos['training']], 'r-')
        plot.plot(epochs, [dto.loss for dto in self._metric_dtos['validate']], 'g-')
        plot.plot(epochs, [dto.lesion.dc for dto in self._metric_dtos['validate']], 'k-')
        plot.plot(epochs, [dto.core.dc for dto in self._metric_dtos['validate']], 'c+')
        plot.plot(epochs, [dto.penu.dc for dto in self._metric_dtos['validate']], 'm+')
        plot.set_ylabel('L Train.(red)/Val.(green) | Dice Val. Lesion(b), Core(c), Penu(m)')
        plot.set_ylim(0, 1)
