"""Base of everything that pushes batches through a model (reference ``common/inference/Inference.py``)."""


class Inference(object):
    IMSHOW_VMAX_CBV = 12
    IMSHOW_VMAX_TTD = 40
    FN_VIS_BASE = '_visual_'
    INFERENCE_INITALIZED = False

    def __init__(self, model):
        # Learner subclasses initialise this base twice through multiple inheritance: keep the first model
        if not self.INFERENCE_INITALIZED:
            self._model = model
            self.INFERENCE_INITALIZED = True

    def inference_step(self, batch: dict):
        raise NotImplementedError

    @property
    def is_cuda(self) -> bool:
        return next(self._model.parameters()).is_cuda
