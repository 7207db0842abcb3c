"""Sample transforms of common/data.py:215-351 (SURVEY.md 8 "next" row N4): the scipy/numpy oracle against its recorded
fixture (CPU), and the device pipeline (HIP kernels + torch index ops) against the oracle (GPU)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import transforms as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transforms.npz")


def test_oracle_reproduces_fixture():
    """elastic_transform against outputs of the reference's ``ElasticDeform.elastic_transform`` (data.py:326-339)"""
    g = np.load(GOLD)
    assert "reference common/data.py" in str(g["generator"])
    for name in ("a", "b"):
        alpha, sigma, seed = g[name + "_params"]
        for kind in ("smooth", "binary"):
            res, _ = T.elastic_transform(g["%s_%s_in" % (name, kind)].astype(np.float64), alpha, sigma, np.random.RandomState(int(seed)))
            np.testing.assert_allclose(res, g["%s_%s_out" % (name, kind)], rtol=0, atol=2e-6)


def _fixture_sample(g):
    return {"case_id": 17, "clinical_idx": 0, "images": g["s_images"].copy(), "labels": g["s_labels"].copy(),
            "clinical": g["s_clinical"].astype(np.float32)}          # (outputs are stored as float32)


def test_oracle_sample_transforms_match_reference_classes():
    """every transform class of the reference, run on one recorded sample (the random draws replayed from the seeds)"""
    g = np.load(GOLD)
    s = _fixture_sample(g)
    # ElasticDeform.__call__: one RandomState through the three label channels (data.py:341-351)
    alpha, sigma = g["s_elastic_params"]
    got = T.elastic_deform({k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in s.items()}, alpha, sigma, False,
                           np.random.RandomState(int(g["s_elastic_seed"])))
    np.testing.assert_allclose(got["labels"], g["s_elastic_labels"], rtol=0, atol=2e-6)
    # HemisphericFlipFixedToCaseId(15) on case 17 flips; HemisphericFlip after random.seed(5): the recorded toss
    f = T.hemispheric_flip(s, True)
    for k in ("images", "labels", "clinical"):
        assert np.array_equal(f[k], g["s_flip_fixed_" + k])
    random.seed(int(g["s_flip_random_seed"]))
    f = T.hemispheric_flip(s, random.random() > 0.5)
    for k in ("images", "labels"):
        assert np.array_equal(f[k], g["s_flip_random_" + k])
    # PadImages(2, 3, 1, pad_value=7)
    p = T.pad_images(s, (2, 3, 1), 7)
    assert np.array_equal(p["images"], g["s_pad_images"]) and np.array_equal(p["labels"], g["s_pad_labels"])
    # RandomPatch(12, 10, 6, 2, 3, 1) after random.seed(9): three random.randint draws in x, y, z order (data.py:262-264)
    random.seed(int(g["s_patch_seed"]))
    o = (random.randint(0, 20 - 12), random.randint(0, 20 - 10), random.randint(0, 8 - 6))
    r = T.random_patch(s, 12, 10, 6, (2, 3, 1), o)
    assert np.array_equal(r["images"], g["s_patch_images"]) and np.array_equal(r["labels"], g["s_patch_labels"])
    # ToTensor
    for k in ("images", "labels", "clinical"):
        assert np.array_equal(T.to_tensor_layout(s[k]), g["s_totensor_" + k])


def test_oracle_layout_transforms():
    rs = np.random.RandomState(0)
    s = {"case_id": 3, "images": rs.rand(8, 8, 4, 2).astype(np.float32), "labels": rs.rand(6, 6, 2, 3).astype(np.float32),
         "clinical": rs.rand(1, 1, 1, 5).astype(np.float32)}
    f = T.hemispheric_flip(s, True)
    assert np.array_equal(f["images"][0], s["images"][-1]) and np.array_equal(f["labels"][::-1], s["labels"])
    p = T.pad_images(s, (1, 2, 3), 7.0)
    assert p["images"].shape == (10, 12, 10, 2) and p["images"][0, 0, 0, 0] == 7.0
    assert np.array_equal(p["images"][1:-1, 2:-2, 3:-3], s["images"])
    assert T.to_tensor_layout(s["images"]).shape == (2, 4, 8, 8)


def _close_but_for_edge_flips(got, want, atol, frac=2e-4):
    """fp32 coordinates against scipy's float64: a sampling point within 1e-5 of the volume's border can land on the
    other side (cval instead of an interpolated value) -- tolerate a vanishing fraction of such voxels."""
    bad = np.abs(got - want) > atol
    assert bad.mean() <= frac, (bad.sum(), bad.size, float(np.abs(got - want).max()))


@pytest.mark.gpu
def test_elastic_transform_matches_scipy():
    from stroke_prediction_amd.common.data import ElasticDeform
    g = np.load(GOLD)
    ed = ElasticDeform()
    for name in ("a", "b"):
        alpha, sigma, seed = g[name + "_params"]
        for kind in ("smooth", "binary"):
            img = torch.from_numpy(g["%s_%s_in" % (name, kind)]).cuda()
            out, _ = ed.elastic_transform(img, alpha, sigma, np.random.RandomState(int(seed)))
            _close_but_for_edge_flips(out.cpu().numpy(), g["%s_%s_out" % (name, kind)], 2e-4)
    # the reference's configuration (alpha 100, sigma 4: 33 taps) at the CAE volume size, against scipy run here
    rs = np.random.RandomState(5)
    vol = (rs.rand(128, 128, 28) > 0.5).astype(np.float64)
    vol = T.gaussian_filter(vol, 2.0)
    want, _ = T.elastic_transform(vol.copy(), 100, 4, np.random.RandomState(42))
    got, _ = ed.elastic_transform(torch.from_numpy(vol.astype(np.float32)).cuda(), 100, 4, np.random.RandomState(42))
    _close_but_for_edge_flips(got.cpu().numpy(), want, 2e-4)


@pytest.mark.gpu
def test_gaussian_filter_and_warp_kernels():
    from stroke_prediction_amd.runtime import lib as L, ops as O
    rs = np.random.RandomState(1)
    for shape, sigma in (((20, 17, 9), 1.0), ((7, 5, 3), 4.0), ((33, 40, 28), 2.5)):      # ragged, and radius > extent
        x = rs.rand(*shape).astype(np.float32)
        xd = torch.from_numpy(x).cuda()
        dst, tmp = torch.empty_like(xd), torch.empty_like(xd)
        L.call("sp_gaussian_filter3d", O.ptr(xd), O.ptr(dst), O.ptr(tmp), shape[0], shape[1], shape[2], sigma, 4.0, O.stream())
        want = T.gaussian_filter(x.astype(np.float64), sigma, mode="constant", cval=0)
        np.testing.assert_allclose(dst.cpu().numpy(), want, rtol=0, atol=2e-6)
        # warp with explicit displacement fields incl. points outside the volume and exactly on its border
        d = [(rs.rand(*shape) * 6 - 3).astype(np.float32) for _ in range(3)]
        d[0][0, :, :] = 0.0
        d[0][-1, :, :] = 0.0
        grid = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
        coords = [grid[a] + d[a].astype(np.float64) for a in range(3)]
        want = T.map_coordinates(x.astype(np.float64), coords, order=1)
        dd = [torch.from_numpy(v).cuda() for v in d]
        out = torch.empty_like(xd)
        L.call("sp_map_coordinates_linear", O.ptr(xd), O.ptr(dd[0]), O.ptr(dd[1]), O.ptr(dd[2]), 1.0, 1.0, 1.0, 0.0, O.ptr(out),
               shape[0], shape[1], shape[2], O.stream())
        _close_but_for_edge_flips(out.cpu().numpy(), want, 1e-5)


@pytest.mark.gpu
def test_device_pipeline_matches_reference_semantics():
    from stroke_prediction_amd.common import data as D
    rs = np.random.RandomState(2)
    s = {"case_id": 12, "clinical_idx": 0, "images": rs.rand(16, 16, 10, 2).astype(np.float32),
         "labels": (rs.rand(16, 16, 10, 3) > 0.5).astype(np.float32), "clinical": rs.rand(1, 1, 1, 5).astype(np.float32)}
    dev = D.to_device(s)
    # flips
    f = D.HemisphericFlipFixedToCaseId(10)(dev)
    ref = T.hemispheric_flip(s, True)
    for k in ("images", "labels", "clinical"):
        assert np.array_equal(f[k].cpu().numpy(), ref[k])
    assert D.HemisphericFlipFixedToCaseId(20)(dev) is dev
    random.seed(3)
    toss = random.random() > 0.5
    random.seed(3)
    f2 = D.HemisphericFlip()(dev)
    assert np.array_equal(f2["images"].cpu().numpy(), T.hemispheric_flip(s, toss)["images"])
    # patch: same random offsets as the reference's random.randint calls
    random.seed(4)
    o = (random.randint(0, 16 - 12), random.randint(0, 16 - 12), random.randint(0, 10 - 8))
    random.seed(4)
    p = D.RandomPatch(12, 12, 8, 2, 2, 1)(dev)
    pr = T.random_patch(s, 12, 12, 8, (2, 2, 1), o)
    assert np.array_equal(p["images"].cpu().numpy(), pr["images"]) and np.array_equal(p["labels"].cpu().numpy(), pr["labels"])
    # pad + layout
    q = D.PadImages(2, 3, 1, pad_value=-1)(dev)
    assert np.array_equal(q["images"].cpu().numpy(), T.pad_images(s, (2, 3, 1), -1)["images"])
    t = D.ToTensor()(dev)
    assert np.array_equal(t["labels"].cpu().numpy(), T.to_tensor_layout(s["labels"])) and tuple(t["images"].shape) == (2, 10, 16, 16)
    # elastic deformation of all label channels with ONE random state (data.py:341-351): patch the clock-seeded state
    import stroke_prediction_amd.common.data as mod
    want = T.elastic_deform({k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in s.items()}, 20, 2, True,
                            np.random.RandomState(9))
    real = mod.np.random.RandomState
    try:
        mod.np.random.RandomState = lambda seed=None: real(9)
        got = D.ElasticDeform(20, 2, apply_to_images=True)(D.to_device(s))
    finally:
        mod.np.random.RandomState = real
    _close_but_for_edge_flips(got["labels"].cpu().numpy(), want["labels"], 2e-4, frac=1e-3)
    _close_but_for_edge_flips(got["images"].cpu().numpy(), want["images"], 2e-4, frac=1e-3)
    with pytest.raises(RuntimeError):
        D.ElasticDeform()(s)        # numpy sample: no silent CPU path


@pytest.mark.gpu
def test_device_pipeline_matches_reference_fixture():
    """the device transforms against outputs of the reference's own classes (tests/golden/transforms.npz)"""
    from stroke_prediction_amd.common import data as D
    import stroke_prediction_amd.common.data as mod
    g = np.load(GOLD)
    s = _fixture_sample(g)
    dev = D.to_device(s)
    f = D.HemisphericFlipFixedToCaseId(split_id=15)(dev)
    for k in ("images", "labels"):
        assert np.array_equal(f[k].cpu().numpy(), g["s_flip_fixed_" + k])
    random.seed(int(g["s_flip_random_seed"]))
    f = D.HemisphericFlip()(dev)
    assert np.array_equal(f["images"].cpu().numpy(), g["s_flip_random_images"])
    q = D.PadImages(2, 3, 1, pad_value=7)(dev)
    assert np.array_equal(q["images"].cpu().numpy(), g["s_pad_images"])
    random.seed(int(g["s_patch_seed"]))
    r = D.RandomPatch(12, 10, 6, 2, 3, 1)(dev)
    assert np.array_equal(r["images"].cpu().numpy(), g["s_patch_images"]) and np.array_equal(r["labels"].cpu().numpy(), g["s_patch_labels"])
    t = D.ToTensor()(dev)
    for k in ("images", "labels", "clinical"):
        assert np.array_equal(t[k].cpu().numpy(), g["s_totensor_" + k])
    alpha, sigma = g["s_elastic_params"]
    real = mod.np.random.RandomState
    try:      # the reference seeds from the wall clock (data.py:327-329); the fixture recorded the seed it drew
        mod.np.random.RandomState = lambda seed=None: real(int(g["s_elastic_seed"]))
        got = D.ElasticDeform(float(alpha), float(sigma))(D.to_device(s))
    finally:
        mod.np.random.RandomState = real
    _close_but_for_edge_flips(got["labels"].cpu().numpy(), g["s_elastic_labels"], 2e-4, frac=1e-3)
