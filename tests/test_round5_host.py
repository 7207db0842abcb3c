"""Round-5 host-side logic (CPU only): the two-shot exchange's chunk / tail arithmetic, the z-marching tile chooser's limits, the
bench line's flat roofline / co-headline keys."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("n", [1, 5, 7, 8, 64, 1000, 355014, 4716955])
def test_two_shot_plan_covers_every_element_once(n, world):
    """parallel.DirectComm.two_shot_plan: reduce-scatter + all-gather over world * chunk elements and a plain all-reduce of the
    tail -- emulated on lists of per-rank buffers, the result is the element-wise sum on every rank (VERDICT r4 "next" 6b: the case
    n % world != 0 used to fall back to one all-reduce silently)"""
    import stroke_prediction_amd  # noqa: F401
    from stroke_prediction_amd.parallel import DirectComm
    chunk, tail = DirectComm.two_shot_plan(n, world)
    assert chunk * world + tail == n and 0 <= tail < world and chunk == n // world
    m = min(n, 4096)                                       # emulate on a prefix-sized problem of the same residue class
    m = max(world, m - (m % world) + (n % world)) if n >= world else n
    chunk, tail = DirectComm.two_shot_plan(m, world)
    rng = np.random.default_rng(n + world)
    bufs = [rng.integers(-5, 6, size=m).astype(np.int64) for _ in range(world)]
    want = sum(bufs)
    if chunk > 0:
        # reduce-scatter: rank r ends with the sum of [r chunk, (r + 1) chunk) at its own offset (sp_reduce_scatter_flat)
        parts = [sum(b[r * chunk:(r + 1) * chunk] for b in bufs) for r in range(world)]
        for r in range(world):
            bufs[r][r * chunk:(r + 1) * chunk] = parts[r]
        # all-gather: every rank receives every rank's chunk (sp_allgather_flat)
        for r in range(world):
            for q in range(world):
                bufs[r][q * chunk:(q + 1) * chunk] = parts[q]
        if tail:
            t = sum(b[world * chunk:] for b in bufs)
            for r in range(world):
                bufs[r][world * chunk:] = t
    else:
        t = sum(bufs)
        bufs = [t.copy() for _ in range(world)]
    for r in range(world):
        np.testing.assert_array_equal(bufs[r], want)


def test_direct_communicator_is_opt_in(monkeypatch):
    """DataParallelSync builds a communicator of its own only when asked (direct=True / SP_DIST_DIRECT=1)"""
    import inspect
    import stroke_prediction_amd  # noqa: F401
    from stroke_prediction_amd import parallel
    src = inspect.getsource(parallel.DataParallelSync.__init__)
    assert "SP_DIST_DIRECT_MIN_BYTES" not in src and "SP_DIST_DIRECT" in src


def test_step_roofline_model_of_the_headline_step():
    """bench.step_roofline_unet: the module-level yardstick of the whole step -- its FLOPs are SURVEY 8(d)'s 345.7 GFLOP/sample for
    the convolutions plus the head, its bytes between the conv-only 553 MB/sample x 3 passes and the unfused reference's traffic"""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.step_roofline_unet(128, bench.CHANNELS, 4, 2500.0, 3.0)
    conv_flop = 4 * 345.7e9
    assert conv_flop < r["flops_algorithmic"] < 1.01 * conv_flop
    assert 3 * 4 * 553e6 * 0.9 < r["bytes_algorithmic"] < 4 * 4 * 823e6
    assert abs(r["frac"] - r["ideal_ms"] / 3.0) < 1e-12 and 0.9 < r["ideal_ms"] < 1.5
    # per-layer max(MFMA, HBM) can only exceed both whole-step quotients
    assert r["ideal_ms"] >= 1e3 * r["flops_algorithmic"] / 2500e12 and r["ideal_ms"] >= 1e3 * r["bytes_algorithmic"] / (bench.HBM_PEAK_GBS * 1e9)


def test_tolerance_mode_lands_in_config_as_flat_scalars():
    """the co-headline keys the driver's record keeps (scalars inside `config`)"""
    sys.path.insert(0, ROOT)
    import bench
    res = {"dtype": "bf16", "config": {"workload": "w"},
           "secondary": {"unet_bf16x3": {"ms_per_step": 4.0, "value": 2.1e9, "roofline": {"frac": 0.3, "conv_roofline": {"frac": 0.4}}}},
           "parity": {"trained_eval": {"bf16": {"max_abs_logit_over_max_logit": 3e-2}, "bf16x3": {"max_abs_logit_over_max_logit": 5e-5}},
                      "trained_train": {"bf16": {"max_abs_logit_over_max_logit": 5e-2}, "bf16x3": {"max_abs_logit_over_max_logit": 1e-4}},
                      "note": "text"}}
    bench.tolerance_mode_into_config(res)
    c = res["config"]
    assert c["tolerance_mode"] == "bf16x3" and c["tolerance_mode_ms_per_step"] == 4.0 and c["tolerance_mode_max_rel_logit"] == 1e-4
    assert c["headline_mode_max_rel_logit"] == 5e-2 and c["tolerance_mode_conv_frac"] == 0.4
    assert all(not isinstance(v, (dict, list)) for v in c.values())
    json.dumps(res)
