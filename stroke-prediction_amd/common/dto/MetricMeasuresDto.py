"""Metric accumulators (reference ``common/dto/MetricMeasuresDto.py:5-75``): element-wise ``add`` of
another DTO of the same type and ``div`` by a scalar, skipping ``None`` and infinite entries."""
import math

from common.dto.Dto import Dto


class MeasuresDto(Dto):
    def add(self, other):
        if not isinstance(other, type(self)):
            raise Exception('A' + str(type(self)) + 'must be added')
        for name, value in other:
            mine = getattr(self, name)
            if mine is None:
                setattr(self, name, value)
            elif isinstance(value, MeasuresDto):
                mine.add(value)
            else:
                setattr(self, name, mine + value)

    def div(self, divisor):
        for name, value in self:
            if value is None:
                continue
            if isinstance(value, MeasuresDto):
                value.div(divisor)
            elif not (isinstance(value, float) and math.isinf(value)):
                setattr(self, name, value / divisor)


class BinaryMeasuresDto(MeasuresDto):
    def __init__(self, dc, hd, assd, precision, sensitivity, specificity):
        MeasuresDto.__init__(self, dc=dc, hd=hd, assd=assd, precision=precision, sensitivity=sensitivity,
                             specificity=specificity)

    @property
    def prc_euclidean_distance(self):
        """distance to the ideal (1, 1) corner of the precision/recall plane"""
        return math.sqrt((1 - self.precision) ** 2 + (1 - self.sensitivity) ** 2)


class MetricMeasuresDto(MeasuresDto):
    def __init__(self, loss, core, penu, lesion):
        MeasuresDto.__init__(self, loss=loss, core=core, penu=penu, lesion=lesion)


def init_dto(loss=None, core_dc=None, core_hd=None, core_assd=None, penu_dc=None, penu_hd=None, penu_assd=None,
             lesion_dc=None, lesion_hd=None, lesion_assd=None, lesion_precision=None, lesion_sensitivity=None,
             lesion_specificity=None):
    return MetricMeasuresDto(loss,
                             BinaryMeasuresDto(core_dc, core_hd, core_assd, None, None, None),
                             BinaryMeasuresDto(penu_dc, penu_hd, penu_assd, None, None, None),
                             BinaryMeasuresDto(lesion_dc, lesion_hd, lesion_assd, lesion_precision,
                                               lesion_sensitivity, lesion_specificity))
