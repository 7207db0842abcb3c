"""MI355X-native 3-D U-Net / CAE training path behind the stroke-prediction API.

Layout mirrors the reference so that it drops in under ``train_unet_segmentation.py`` /
``train_shape_reconstruction.py``: with this directory on ``sys.path`` the reference's own
imports (``from common.model.Unet3D import Unet3D``, ``from learner.Learner import Learner`` ...)
resolve to the modules here.  ``import stroke_prediction_amd`` puts it there and makes
``stroke_prediction_amd.common`` / ``.learner`` / ``.tester`` aliases of those same modules.
``runtime/`` holds the ctypes binding of ``libstroke_amd.so`` (hand-written gfx950 kernels in
``csrc/``), the convolution planner and the network engines.
"""
import importlib as _importlib
import importlib.abc as _abc
import importlib.util as _util
import os as _os
import sys as _sys

__version__ = "0.1.0"

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "stroke-prediction_amd")
if not _os.path.isdir(_PKG_DIR):           # imported directly from inside the hyphenated directory
    _PKG_DIR = _os.path.dirname(_os.path.abspath(__file__))
if _PKG_DIR not in _sys.path:
    _sys.path.insert(0, _PKG_DIR)


class _DropInAlias(_abc.MetaPathFinder, _abc.Loader):
    """``stroke_prediction_amd.{common,learner,tester}[.x.y]`` -> the top-level drop-in module objects."""
    ROOTS = ("common", "learner", "tester")
    PREFIX = "stroke_prediction_amd."

    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(self.PREFIX) and fullname[len(self.PREFIX):].split(".")[0] in self.ROOTS:
            return _util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return _importlib.import_module(spec.name[len(self.PREFIX):])

    def exec_module(self, module):
        pass


if not any(type(f).__name__ == "_DropInAlias" for f in _sys.meta_path):
    _sys.meta_path.insert(0, _DropInAlias())
