"""Batch dict -> UnetDto -> model (reference ``common/inference/UnetInference.py:15-27``)."""
import common.dto.UnetDto as UnetDtoUtil
from common import data
from common.inference.Inference import Inference


class UnetInference(Inference):
    def __init__(self, model):
        Inference.__init__(self, model)

    def inference_step(self, batch):
        images = batch[data.KEY_IMAGES]
        labels = batch[data.KEY_LABELS]
        core_gt = labels[:, 0:1]        # == labels[:, 0].unsqueeze(1)
        penu_gt = labels[:, 1:2]
        if self.is_cuda:
            dev = next(self._model.parameters()).device
            images = images.to(dev, non_blocking=True)
            core_gt = core_gt.to(dev, non_blocking=True)
            penu_gt = penu_gt.to(dev, non_blocking=True)
        return self._model(UnetDtoUtil.init_dto(images, core_gt, penu_gt))
