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


def test_zm_config_matches_the_planner():
    """sp_conv3d_zm_config (C side) and plan.ZM_CONFIGS (planner) describe the same kernels (no GPU needed: pure host code)"""
    import ctypes as C
    from stroke_prediction_amd.runtime import plan as P
    lib = L.load()
    for p in range(1, 8):
        for nt in range(1, 8):
            mt, ns, nw = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            rc = lib.sp_conv3d_zm_config(p, nt, C.byref(mt), C.byref(ns), C.byref(nw))
            if (p, nt) in P.ZM_CONFIGS:
                assert rc == 0 and (mt.value, ns.value, nw.value) == P.ZM_CONFIGS[(p, nt)], (p, nt)
            else:
                assert rc != 0, (p, nt)
