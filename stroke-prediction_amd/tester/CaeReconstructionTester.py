"""Per-case evaluation of a trained shape CAE (reference ``tester/CaeReconstructionTester.py:13-62``): the predicted
lesion is the decoded latent interpolation at the case's normalised time to treatment; Dice / Hausdorff / ASSD of
prediction vs follow-up lesion and the Dice of the core / penumbra reconstructions.  NIfTI export (nibabel, private
paths) is replaced by ``.npy`` arrays in (x, y, z) orientation when an output base is given."""
import numpy as np

import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common import data, metrics
from common.dto.CaeDto import CaeDto
from common.dto.MetricMeasuresDto import MetricMeasuresDto
from common.inference.CaeInference import CaeInference
from tester.Tester import Tester


class CaeReconstructionTester(Tester, CaeInference):
    def __init__(self, dataloader, path_model, path_outputs_base='/tmp/', normalization_hours_penumbra=10):
        Tester.__init__(self, dataloader, path_model, path_outputs_base=path_outputs_base)
        CaeInference.__init__(self, self._model, normalization_hours_penumbra)

    def batch_metrics_step(self, dto: CaeDto):
        rec, gt = dto.reconstructions.gtruth, dto.given_variables.gtruth
        m = MetricMeasuresDtoInit.init_dto()
        m.lesion = metrics.binary_measures_torch(rec.interpolation, gt.lesion, self.is_cuda)
        m.core = metrics.binary_measures_torch(rec.core, gt.core, self.is_cuda)
        m.penu = metrics.binary_measures_torch(rec.penu, gt.penu, self.is_cuda)
        return m

    def save_inference(self, dto: CaeDto, batch: dict, suffix=''):
        if not self._path_outputs_base:
            return
        case_id = int(batch[data.KEY_CASE_ID])
        rec = dto.reconstructions.gtruth
        for name, t in (('_core', rec.core), ('_pred', rec.interpolation), ('_penu', rec.penu)):
            np.save(self._path_outputs_base + '_' + str(case_id) + name + str(suffix) + '.npy',
                    t.detach().float().cpu().numpy()[0, 0].transpose(2, 1, 0))

    def print_inference(self, batch: dict, batch_metrics: MetricMeasuresDto, dto: CaeDto, note=''):
        g = batch[data.KEY_GLOBAL]
        f = lambda v: float('nan') if v is None else float(v)
        les = batch_metrics.lesion
        print('Case Id={}\ttA-tO={:.3f}\ttR-tA={:.3f}\tnormalized_time_to_treatment={:.3f}\t-->\tDC={:.3f}\tHD={:.3f}\tASSD={:.3f}'
              '\tDC Core={:.3f}\tDC Penumbra={:.3f}\tPrecision={:.3}\tRecall/Sensitivity={:.3}\tSpecificity={:.3}\t'
              'DistToCornerPRC={:.3}\t{}'.format(int(batch[data.KEY_CASE_ID]), float(g[:, 0].reshape(-1)[0]), float(g[:, 1].reshape(-1)[0]),
                                                 float(dto.given_variables.time_to_treatment.reshape(-1)[0]), f(les.dc), f(les.hd),
                                                 f(les.assd), f(batch_metrics.core.dc), f(batch_metrics.penu.dc), f(les.precision),
                                                 f(les.sensitivity), f(les.specificity), f(les.prc_euclidean_distance), note))
