"""Per-case evaluation of a trained U-Net (reference ``tester/UnetSegmentationTester.py:13-45``): forward through the
eval-mode model, confusion / overlap / surface measures per case on the device.  The reference additionally writes the
probability maps as NIfTI volumes registered to the case's TTD map (nibabel, private paths): out of scope here, the
``save_inference`` hook writes ``.npy`` arrays in the same (x, y, z) orientation instead when an output base is given."""
import numpy as np

import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common import data, metrics
from common.dto.MetricMeasuresDto import MetricMeasuresDto
from common.dto.UnetDto import UnetDto
from common.inference.UnetInference import UnetInference
from tester.Tester import Tester


class UnetSegmentationTester(Tester, UnetInference):
    def __init__(self, dataloader, path_model, path_outputs_base='/tmp/', padding=None):
        Tester.__init__(self, dataloader, path_model, path_outputs_base=path_outputs_base)
        self._pad = padding

    def batch_metrics_step(self, dto: UnetDto):
        m = MetricMeasuresDtoInit.init_dto()
        m.core = metrics.binary_measures_torch(dto.outputs.core, dto.given_variables.core, self.is_cuda)
        m.penu = metrics.binary_measures_torch(dto.outputs.penu, dto.given_variables.penu, self.is_cuda)
        return m

    def _to_xyz(self, t):
        """(1, 1, z, y, x) network output -> (x, y, z) volume with the patch padding removed"""
        v = t.detach().float().cpu().numpy()[0, 0].transpose(2, 1, 0)
        if self._pad is not None and all(p > 0 for p in self._pad):
            v = v[self._pad[0]:-self._pad[0], self._pad[1]:-self._pad[1], self._pad[2]:-self._pad[2]]
        return v

    def save_inference(self, dto: UnetDto, batch: dict, suffix=''):
        if not self._path_outputs_base:
            return
        case_id = int(batch[data.KEY_CASE_ID])
        for name, t in (('_core', dto.outputs.core), ('_penu', dto.outputs.penu)):
            np.save(self._path_outputs_base + '_' + str(case_id) + name + str(suffix) + '.npy', self._to_xyz(t))

    def print_inference(self, batch: dict, batch_metrics: MetricMeasuresDto, dto: UnetDto = None):
        print('Case Id {}:\t DC Core:{:.3},\tDC Penumbra:{:.3}'.format(int(batch[data.KEY_CASE_ID]), float(batch_metrics.core.dc),
                                                                      float(batch_metrics.penu.dc)))
