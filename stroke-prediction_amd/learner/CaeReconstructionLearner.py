"""CAE shape-reconstruction learner (reference ``learner/CaeReconstructionLearner.py``).

loss (``loss_step`` :52-70) = [ mean(|p-i|-(p-i)) + mean(|p-c|-(p-c)) + Dice(c) + Dice(p) + Dice(l)
+ f * mean|z_i - z_l| ] / (5 + f),  f = min(0.04*max(0, epoch-25), 1); beta1 warm-up 0.5 -> 0.9 over the
first four epochs (``adapt_betas`` :28-40)."""
import torch

import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common import metrics
from common.dto.CaeDto import CaeDto
from common.inference.CaeInference import CaeInference
from learner.Learner import Learner


class CaeReconstructionLearner(Learner, CaeInference):
    FN_VIS_BASE = '_cae1_'
    FNB_MARKS = '_cae1'
    N_EPOCHS_ADAPT_BETA1 = 4

    def __init__(self, dataloader_training, dataloader_validation, cae_model, optimizer, scheduler, n_epochs,
                 path_previous_base, path_outputs_base, criterion, normalization_hours_penumbra=10, verbose=True, **learner_kw):
        Learner.__init__(self, dataloader_training, dataloader_validation, cae_model, optimizer, scheduler, n_epochs,
                         path_previous_base, path_outputs_base, **learner_kw)
        CaeInference.__init__(self, cae_model, normalization_hours_penumbra)
        self._criterion = criterion
        self._verbose = verbose

    def adapt_betas(self, epoch):
        betas = self._optimizer.defaults['betas']
        if epoch > self.N_EPOCHS_ADAPT_BETA1:
            return
        if epoch < self.N_EPOCHS_ADAPT_BETA1:
            betas = (betas[0] - 0.1 * (self.N_EPOCHS_ADAPT_BETA1 - epoch),) + tuple(betas[1:])
        for param_group in self._optimizer.param_groups:
            param_group['betas'] = tuple(betas)
        if self._verbose:
            print('Momentum betas have been set to:', tuple(betas), end=' ')

    def get_start_epoch(self):
        return len(self._metric_dtos['training'])

    def get_start_min_loss(self):
        losses = [dto.loss for dto in self._metric_dtos['validate']]
        return min(losses) if losses else float('inf')

    def graph_key(self, epoch):
        return min(0.04 * max(0, epoch - 25), 1)       # the latent-loss ramp is a Python constant of the captured step

    def loss_step(self, dto: CaeDto, epoch):
        factor = min(0.04 * max(0, epoch - 25), 1)
        if self._verbose:
            print(factor, end=' ')
        rec, gt, lat = dto.reconstructions.gtruth, dto.given_variables.gtruth, dto.latents.gtruth
        fused = metrics.cae_reconstruction_loss(rec, gt, lat, factor, self._criterion)      # three launches instead of ~60 (same value)
        if fused is not None:
            return fused
        diff_penu_fuct = rec.penu - rec.interpolation
        diff_penu_core = rec.penu - rec.core
        loss = metrics.batch_mean(torch.abs(diff_penu_fuct) - diff_penu_fuct)      # (= torch.mean outside the exact data-parallel mode)
        loss = loss + metrics.batch_mean(torch.abs(diff_penu_core) - diff_penu_core)
        loss = loss + self._criterion(rec.core, gt.core)
        loss = loss + self._criterion(rec.penu, gt.penu)
        loss = loss + self._criterion(rec.lesion, gt.lesion)
        loss = loss + factor * metrics.batch_mean(torch.abs(lat.interpolation - lat.lesion))
        return loss / (5 + factor)

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
