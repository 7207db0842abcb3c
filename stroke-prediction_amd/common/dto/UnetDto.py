"""U-Net DTO (reference ``common/dto/UnetDto.py:4-28``): ``given_variables`` carries
``input_modalities`` and the optional ground-truth masks, ``outputs`` receives ``core``/``penu``."""
from common.dto.Dto import Dto


class UnetDto(Dto):
    def __init__(self, given_variables, outputs):
        Dto.__init__(self, given_variables=given_variables, outputs=outputs)


def init_dto(input_modalities, gtruth_core=None, gtruth_penumbra=None, gtruth_lesion=None):
    given = Dto(input_modalities=input_modalities, core=gtruth_core, penu=gtruth_penumbra, lesion=gtruth_lesion)
    return UnetDto(given, Dto(core=None, penu=None, lesion=None))
