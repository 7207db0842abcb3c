"""The C-ABI library loads on a CPU-only box and exports every symbol include/stroke_amd.h declares."""
import ctypes
import os
import re

from stroke_prediction_amd.runtime import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "stroke_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L.build()
    lib = ctypes.CDLL(L.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    assert sorted(names) == L.EXPORTS, (sorted(set(names) ^ set(L.EXPORTS)))


def test_version_and_error_text():
    lib = L.load()
    assert lib.sp_version() >= 100
    # argument validation happens before any GPU work: callable without a device
    rc = lib.sp_bn_stats(None, 0, 10, 8, None, None)
    assert rc == -1
    assert "sp_bn_stats" in L.last_error()
