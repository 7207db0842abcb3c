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


def test_plane_serial_config_matches_the_planner():
    """sp_conv3d_zm_config_ps (plane-serial instances, round 5) and plan.ZM_CONFIGS_PS agree"""
    import ctypes as C
    from stroke_prediction_amd.runtime import plan as P
    lib = L.load()
    for nt in range(1, 5):
        for hl in (0, 1):
            mt, ns, nw = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            rc = lib.sp_conv3d_zm_config_ps(nt, hl, C.byref(mt), C.byref(ns), C.byref(nw))
            if (nt, bool(hl)) in P.ZM_CONFIGS_PS:
                assert rc == 0 and (mt.value, ns.value, nw.value) == P.ZM_CONFIGS_PS[(nt, bool(hl))], (nt, hl)
            else:
                assert rc != 0, (nt, hl)


def test_conv3d_plan_tables_match_the_planner():
    """sp_conv3d_plan / sp_conv3d_tables (the header-only caller's route to the z-marching kernel) build the same K tables
    as runtime/plan.py:zm_plan does for the Python host side, forward and data gradient (pure host code: no GPU)"""
    import ctypes as C
    import numpy as np
    from stroke_prediction_amd.runtime import plan as P
    lib = L.load()
    for cin, cout in ((16, 16), (16, 32), (32, 16), (32, 32), (48, 16), (16, 48)):
        for grad in (0, 1):
            dims = (20, 22, 40)
            d = L.Conv3dDesc(2, cin, cout, *dims, grad)
            pl = L.Conv3dPlan()
            op = (P.conv_dgrad_op(cin, cout, 3, 1, 0, dims, cout, cin, 0) if grad else
                  P.conv_fwd_op(cin, cout, 3, 1, 0, dims, cin, cout, 0))
            z = P.zm_plan(op)
            if z is not None:      # (the C planner keeps the classic tile: NW MT rows of 16 voxels)
                z = P.zm_plan(op, tile=(16, z["NW"] * z["MT"]))
            rc = lib.sp_conv3d_plan(C.byref(d), C.byref(pl))
            if z is None:
                assert rc != 0, (cin, cout, grad)
                continue
            assert rc == 0, L.last_error()
            assert (pl.P, pl.NT, pl.MT, pl.NW, pl.KS, pl.nsteps, pl.ITH) == (z["P"], z["NT"], z["MT"], z["NW"], z["KS"], z["nsteps"], z["ITH"])
            assert (pl.Do, pl.Ho, pl.Wo) == tuple(op.subs[0].out_dims) and (pl.o0, pl.o0H, pl.o0W) == tuple(op.subs[0].o0)
            assert (pl.Di, pl.Hi, pl.Wi) == tuple(op.in_dims)
            ktab = np.zeros(4 * pl.KS, np.int32); kmap = np.zeros(12 * pl.KS, np.int32)
            assert lib.sp_conv3d_tables(C.byref(d), C.byref(pl), ktab.ctypes.data, kmap.ctypes.data) == 0
            np.testing.assert_array_equal(ktab, z["ktab"])
            np.testing.assert_array_equal(kmap, z["kmap"])
            assert pl.workspace_bytes >= pl.off_wfrag + pl.nsteps * pl.NT * 1024 and pl.off_ktab >= 256
    # round 5: padded convolutions and their data gradients (the CAE's stride-1 layers) and stride-1 transposed convolutions
    for cin, cout, pad in ((16, 16, (1, 0, 0)), (32, 32, (1, 2, 2)), (16, 32, (1, 1, 1))):
        for kind in ("fwd", "grad", "convT"):
            dims = (9, 20, 22)
            d = L.Conv3dDesc(2, cin, cout, *dims, 1 if kind == "grad" else 0, *pad, 1 if kind == "convT" else 0)
            pl = L.Conv3dPlan()
            op = {"fwd": lambda: P.conv_fwd_op(cin, cout, 3, 1, pad, dims, cin, cout, 0),
                  "grad": lambda: P.conv_dgrad_op(cin, cout, 3, 1, pad, dims, cout, cin, 0),
                  "convT": lambda: P.convT_fwd_op(cin, cout, 3, 1, pad, dims, cin, cout, 0)}[kind]()
            z = P.zm_plan(op, tile="classic")
            assert z is not None and lib.sp_conv3d_plan(C.byref(d), C.byref(pl)) == 0, L.last_error()
            assert (pl.Di, pl.Hi, pl.Wi) == tuple(op.in_dims) and (pl.Do, pl.Ho, pl.Wo) == tuple(op.subs[0].out_dims), (kind, pad)
            assert (pl.o0, pl.o0H, pl.o0W) == tuple(op.subs[0].o0), (kind, pad, (pl.o0, pl.o0H, pl.o0W), op.subs[0].o0)
            ktab = np.zeros(4 * pl.KS, np.int32); kmap = np.zeros(12 * pl.KS, np.int32)
            assert lib.sp_conv3d_tables(C.byref(d), C.byref(pl), ktab.ctypes.data, kmap.ctypes.data) == 0
            np.testing.assert_array_equal(ktab, z["ktab"])
            np.testing.assert_array_equal(kmap, z["kmap"])
    # channel counts without a kernel, and malformed descriptors, are refused with a message
    bad = L.Conv3dDesc(1, 64, 64, 20, 20, 20, 0)
    assert lib.sp_conv3d_plan(C.byref(bad), C.byref(L.Conv3dPlan())) == -1 and "sp_conv3d_plan" in L.last_error()
    bad = L.Conv3dDesc(1, 8, 16, 20, 20, 20, 0)
    assert lib.sp_conv3d_plan(C.byref(bad), C.byref(L.Conv3dPlan())) == -1


def test_zm8_config_matches_the_planner():
    """sp_conv3d_zm8_config (fp8 kernel instances) and plan.ZM8_CONFIGS agree"""
    import ctypes as C
    from stroke_prediction_amd.runtime import plan as P
    lib = L.load()
    for p in range(1, 10):
        for nt in range(1, 6):
            mt, ns, nw = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            rc = lib.sp_conv3d_zm8_config(p, nt, C.byref(mt), C.byref(ns), C.byref(nw))
            if (p, nt) in P.ZM8_CONFIGS:
                assert rc == 0 and (mt.value, ns.value, nw.value) == P.ZM8_CONFIGS[(p, nt)], (p, nt)
            else:
                assert rc != 0, (p, nt)


def test_argument_validation_refuses_before_any_gpu_work():
    """every entry point checks its arguments on the host and returns SP_EINVAL with a message naming itself -- callable without
    a device, and what the sanitizer build (tools/build_asan.sh) walks through"""
    import ctypes as C
    lib = L.load()
    a = L.ConvArgs()
    for name in ("sp_conv3d_zm", "sp_conv3d_zm8"):
        assert getattr(lib, name)(C.byref(a), None, None) == -1 and name in L.last_error()
    assert lib.sp_conv3d_igemm(C.byref(a), None) == -1
    w = L.WgradArgs()
    assert lib.sp_conv3d_wgrad(C.byref(w), None) == -1
    f = L.ConvFcArgs()
    assert lib.sp_conv_fc(C.byref(f), None) == -1
    assert lib.sp_quantize_f8(None, 16, 0, None, 0, 10, 0, 1.0, None) == -1 and "sp_quantize_f8" in L.last_error()
    assert lib.sp_conv_prep_f8(None, 0, 0, 16, 16, None, 1, 1, None, None, None, 27, None, None, None, 1.0, None) == -1
    assert lib.sp_maxpool2_fwd_q8(None, None, 0, 1, 4, 4, 4, 16, None, None, 0, 0, 1.0, None) == -1
    assert lib.sp_bn_act_bwd_q8(None, None, None, 0, 10, 16, 1, 0.01, None, None, None, 0, 0, 1.0, None) == -1
    assert lib.sp_adam_step_flat(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == -1
    assert lib.sp_allreduce_flat(None, None, 10, None) == -1 and "sp_allreduce_flat" in L.last_error()


def test_rccl_unique_id_through_the_c_abi():
    """sp_comm_* resolve librccl at run time; creating a unique id needs no GPU"""
    import ctypes as C
    lib = L.load()
    if not lib.sp_comm_available():
        import pytest
        pytest.skip("no librccl in this process")
    a, b = C.create_string_buffer(128), C.create_string_buffer(128)
    assert lib.sp_comm_unique_id(a) == 0 and lib.sp_comm_unique_id(b) == 0
    assert a.raw != b.raw and any(a.raw)
    assert lib.sp_comm_init_rank(None, 1, a, 0) == -1


def test_sanitizer_build_of_the_host_code_passes_this_file():
    """SURVEY 5 / VERDICT r2 item 5c: the C ABI's host code under AddressSanitizer + UBSan (tools/build_asan.sh; the device
    code is untouched) runs this file clean.  The variant library is (re)built whenever it is missing or older than a source --
    and the test is never re-entered from the sanitizer run itself."""
    import subprocess
    import pytest
    if os.environ.get("SP_LIB_PATH", "").endswith("_asan.so"):
        pytest.skip("already inside the sanitizer run")
    so = os.path.join(ROOT, "stroke-prediction_amd", "lib", "variants", "libstroke_amd_asan.so")
    if not os.path.exists(os.path.join(ROOT, "tools", "build_asan.sh")):
        pytest.skip("CPU box only: the sanitizer scripts do not travel to the GPU pool (.gpurunignore)")
    # the variant must be as new as the sources: a missing or STALE one is rebuilt here (about two minutes, objects cached per
    # source under lib/variants/asan_obj), never skipped -- a skipped sanitizer run looks like a clean one
    srcs = [os.path.join(ROOT, "stroke-prediction_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "stroke-prediction_amd", "csrc"))]
    srcs.append(os.path.join(ROOT, "include", "stroke_amd.h"))
    if os.environ.get("SP_RUN_ASAN") == "1" or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run([os.path.join(ROOT, "tools", "build_asan.sh")], check=True, timeout=1500)
    assert os.path.exists(so) and all(os.path.getmtime(s) <= os.path.getmtime(so) for s in srcs), "tools/build_asan.sh left a stale variant"
    r = subprocess.run([os.path.join(ROOT, "tools", "run_asan_tests.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "passed" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
