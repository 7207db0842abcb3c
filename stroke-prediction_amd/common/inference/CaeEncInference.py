"""Batch dict -> CaeDto -> new encoder -> frozen decoder -> frozen CAE (reference ``common/inference/CaeEncInference.py:9-42``).

Phase 2 of the reference's pipeline: a NEW encoder sees the U-Net's core / penumbra segmentations
(``batch['images'][:, 0:2]``), the frozen shape CAE decodes its latents (``reconstructions.inputs.*``) and, in a second
call, encodes / decodes the manual masks (``latents.gtruth.*``: the targets of the latent terms of the loss).

One deliberate difference, stated because it changes behaviour: the reference writes the branch selector to ``dto.mode``
while the models read ``dto.flag``, so its second model call re-enters the ``inputs`` branch with the
latents of the first one in place and trips its own "do not overwrite" assertion (``Cae3D.py:110``) -- the class cannot run
as written.  Here the selector is written to ``dto.flag`` (and to ``dto.mode``, for readers of either), which is the evident
intent: first call ``inputs`` only, second call ``gtruth`` only.
"""
import common.dto.CaeDto as CaeDtoUtil
from common import data
from common.dto.CaeDto import CaeDto
from common.inference.CaeInference import CaeInference


class CaeEncInference(CaeInference):
    def __init__(self, model, new_enc, normalization_hours_penumbra=10):
        CaeInference.__init__(self, model, normalization_hours_penumbra)
        self._new_enc = new_enc

    def infer(self, dto: CaeDto):
        pass

    def init_unet_segm_variables(self, batch: dict, dto: CaeDto):
        images = batch[data.KEY_IMAGES]
        if self.is_cuda:
            images = images.to(self._device(), non_blocking=True)
        dto.given_variables.inputs.core = images[:, 0:1].float()
        dto.given_variables.inputs.penu = images[:, 1:2].float()
        return dto

    def inference_step(self, batch: dict, step=None):
        dto = self.init_clinical_variables(batch, step)

        dto.mode = dto.flag = CaeDtoUtil.FLAG_INPUTS
        dto = self.init_unet_segm_variables(batch, dto)
        dto = self._new_enc(dto)
        dto = self._model.dec(dto)

        dto.mode = dto.flag = CaeDtoUtil.FLAG_GTRUTH
        dto = self.init_gtruth_segm_variables(batch, dto)
        dto = self._model(dto)

        return dto
