"""CAE DTO (reference ``common/dto/CaeDto.py:1-46``).  ``flag`` selects which branches the
encoder/decoder run (``Cae3D.py:103,111,228,234``); it defaults to FLAG_DEFAULT = both."""
from common.dto.Dto import Dto

FLAG_DEFAULT = 'default'
FLAG_GTRUTH = 'gtruth'
FLAG_INPUTS = 'inputs'


class CaeDto(Dto):
    def __init__(self, given_variables, latents, reconstructions):
        Dto.__init__(self, given_variables=given_variables, latents=latents, reconstructions=reconstructions)
        self.flag = FLAG_DEFAULT


def _result_tree():
    return Dto(inputs=Dto(core=None, penu=None, interpolation=None),
               gtruth=Dto(core=None, penu=None, interpolation=None, lesion=None))


def init_dto(global_variables, time_to_treatment, type_core, type_penumbra, inputs_core, inputs_penu,
             gtruth_core, gtruth_penumbra, gtruth_lesion):
    given = Dto(globals=global_variables, time_to_treatment=time_to_treatment,
                scalar_types=Dto(core=type_core, penu=type_penumbra),
                inputs=Dto(core=inputs_core, penu=inputs_penu),
                gtruth=Dto(core=gtruth_core, penu=gtruth_penumbra, lesion=gtruth_lesion))
    return CaeDto(given, _result_tree(), _result_tree())
