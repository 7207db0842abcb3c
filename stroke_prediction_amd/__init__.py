"""Import alias: the package lives in ``stroke-prediction_amd/`` (not an importable name);
``import stroke_prediction_amd`` resolves to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "stroke-prediction_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
