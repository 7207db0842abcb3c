"""Batch dict -> CaeDto -> model (reference ``common/inference/CaeInference.py:18-69``).

``time_to_treatment = tA->tR / (normalization_hours_penumbra - tO->tA)`` per sample as a
B x 1 x 1 x 1 x 1 float tensor; labels[:, 0..2] are the core / penumbra / follow-up lesion masks.
As in the reference, ``inference_step`` sets ``dto.mode`` (not ``dto.flag``), so the models run
their default branch set."""
import torch

import common.dto.CaeDto as CaeDtoUtil
from common import data
from common.dto.CaeDto import CaeDto
from common.inference.Inference import Inference


class CaeInference(Inference):
    def __init__(self, model, normalization_hours_penumbra=10):
        Inference.__init__(self, model)
        self._normalization_hours_penumbra = normalization_hours_penumbra

    def _device(self):
        return next(self._model.parameters()).device

    def _get_normalization(self, batch):
        to_to_ta = batch[data.KEY_GLOBAL][:, 0].reshape(-1, 1).float()
        return self._normalization_hours_penumbra - to_to_ta

    def get_time_to_treatment(self, batch, global_variables, step):
        normalization = self._get_normalization(batch)
        if step is None:
            ta_to_tr = batch[data.KEY_GLOBAL][:, 1].reshape(-1, 1).float()
            ttt = ta_to_tr / normalization
        else:
            ttt = (step * torch.ones(global_variables.size(0), 1, device=normalization.device)) / normalization
        return ttt.reshape(-1, 1, 1, 1, 1)

    def init_clinical_variables(self, batch: dict, step):
        globals_incl_time = batch[data.KEY_GLOBAL].float()
        n = globals_incl_time.size(0)
        time_to_treatment = self.get_time_to_treatment(batch, globals_incl_time, step)
        dev = self._device() if self.is_cuda else globals_incl_time.device
        # created on the target device: no host->device copy of constants inside the (graph-capturable) step
        type_core = torch.zeros(n, 1, 1, 1, 1, device=dev)
        type_penumbra = torch.ones(n, 1, 1, 1, 1, device=dev)
        if time_to_treatment is not None:        # (CaeStepLearner: the encoder predicts the step)
            time_to_treatment = time_to_treatment.to(dev)
        globals_incl_time = globals_incl_time.to(dev)
        return CaeDtoUtil.init_dto(globals_incl_time, time_to_treatment, type_core, type_penumbra,
                                   None, None, None, None, None)

    def init_gtruth_segm_variables(self, batch: dict, dto: CaeDto):
        labels = batch[data.KEY_LABELS]
        if self.is_cuda:
            labels = labels.to(self._device(), non_blocking=True)
        dto.given_variables.gtruth.core = labels[:, 0:1].float()
        dto.given_variables.gtruth.penu = labels[:, 1:2].float()
        dto.given_variables.gtruth.lesion = labels[:, 2:3].float()
        return dto

    def infer(self, dto: CaeDto):
        return self._model(dto)

    def inference_step(self, batch: dict, step=None):
        dto = self.init_clinical_variables(batch, step)
        dto.mode = CaeDtoUtil.FLAG_GTRUTH
        dto = self.init_gtruth_segm_variables(batch, dto)
        return self.infer(dto)
