"""U-Net segmentation learner (reference ``learner/UnetSegmentationLearner.py``; its constructor there
calls ``Learner.__init__`` without ``self`` and cannot run -- this one implements the intended behaviour,
mirroring the self-consistent ``CaeReconstructionLearner``)."""
import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common import metrics
from common.dto.UnetDto import UnetDto
from common.inference.UnetInference import UnetInference
from learner.Learner import Learner


class UnetSegmentationLearner(Learner, UnetInference):
    FNB_MARKS = '_unet'

    def __init__(self, dataloader_training, dataloader_validation, unet_model, optimizer, scheduler, n_epochs,
                 criterion, path_previous_base=None, path_outputs_base='/tmp/unet-segmentation',
                 surface_metrics=True, **learner_kw):
        Learner.__init__(self, dataloader_training, dataloader_validation, unet_model, optimizer, scheduler,
                         n_epochs, path_previous_base, path_outputs_base, **learner_kw)
        UnetInference.__init__(self, unet_model)
        self._criterion = criterion
        self._surface_metrics = surface_metrics

    def loss_step(self, dto: UnetDto, epoch):
        """(Dice(core) + Dice(penu)) / 2, UnetSegmentationLearner.py:21-28."""
        return metrics.mean_of_channel_losses(self._criterion, (dto.outputs.core, dto.outputs.penu),
                                              (dto.given_variables.core, dto.given_variables.penu))

    def batch_metrics_step(self, dto: UnetDto, epoch):
        batch_metrics = MetricMeasuresDtoInit.init_dto()
        batch_metrics.core = metrics.binary_measures_torch(dto.outputs.core, dto.given_variables.core, self.is_cuda)
        batch_metrics.penu = metrics.binary_measures_torch(dto.outputs.penu, dto.given_variables.penu, self.is_cuda)
        return batch_metrics

    def get_start_epoch(self):
        return len(self._metric_dtos['training'])

    def get_start_min_loss(self):
        losses = [dto.loss for dto in self._metric_dtos['validate']]
        return min(losses) if losses else float('inf')

    def print_epoch(self, epoch, phase, epoch_metrics):
        print('\nEpoch {}/{} {} loss: {:.3} - DC Core:{:.3}, DC Penumbra:{:.3}'.format(
            epoch + 1, self._n_epochs, phase, float(epoch_metrics.loss), float(epoch_metrics.core.dc),
            float(epoch_metrics.penu.dc)), end=' ')

    def plot_epoch(self, plot, epochs):
        plot.plot(epochs, [dto.loss for dto in self._metric_dtos['training']], 'r-')
        plot.plot(epochs, [dto.loss for dto in self._metric_dtos['validate']], 'g-')
        plot.plot(epochs, [dto.core.dc for dto in self._metric_dtos['validate']], 'c+')
        plot.plot(epochs, [dto.penu.dc for dto in self._metric_dtos['validate']], 'm+')
        plot.set_ylabel('L Train.(red)/Val.(green) | Dice Val. Core(c), Penu(m)')
