"""Forward-only evaluation loop (reference ``tester/Tester.py:9-45``): loads a whole pickled model,
freezes it, switches BatchNorm to running statistics and pushes one case at a time through
``inference_step``."""
import torch

import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common.dto.Dto import Dto
from common.inference.Inference import Inference


class Tester(Inference):
    def __init__(self, dataloader, path_model, path_outputs_base: str = '/tmp/'):
        model = torch.load(path_model, weights_only=False) if isinstance(path_model, str) else path_model
        Inference.__init__(self, model)
        assert dataloader.batch_size == 1, "You must ensure a batch size of 1 for correct case metric measures."
        self._dataloader = dataloader
        self._path_outputs_base = path_outputs_base
        self._model.freeze(True)
        self._model.eval()

    def infer_batch(self, batch: dict):
        with torch.no_grad():
            dto = self.inference_step(batch)
        batch_metrics = self.batch_metrics_step(dto)
        self.save_inference(dto, batch)
        return batch_metrics, dto

    def batch_metrics_step(self, dto: Dto):
        return MetricMeasuresDtoInit.init_dto()

    def save_inference(self, dto: Dto, batch: dict):
        pass

    def print_inference(self, batch: dict, metrics, dto: Dto = None):
        pass

    def run_inference(self):
        for batch in self._dataloader:
            batch_metrics, dto = self.infer_batch(batch)
            self.print_inference(batch, batch_metrics, dto)
