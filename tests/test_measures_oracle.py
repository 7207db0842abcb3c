"""Known-answer tests that pin ``oracle/measures.py`` (the restatement of MedPy 0.3.0's ``metric.binary``, which the
reference calls in common/metrics.py:31-46 and which is absent from this image).  Every expected value below is derived by
hand from MedPy's documented definitions: surface = mask minus its erosion with the connectivity-1 cross (array edge counts
as background), surface distance = Euclidean distance to the nearest surface voxel of the other object, HD = the larger of
the two directed maxima, ASSD = the mean of the two directed means.  CPU only."""
import math

import numpy as np
import pytest

from oracle import measures as M


def _vol(shape, *boxes):
    a = np.zeros(shape, dtype=bool)
    for lo, hi in boxes:
        a[tuple(slice(l, h) for l, h in zip(lo, hi))] = True
    return a


def test_two_single_voxels():
    a = _vol((6, 6, 6), ((1, 1, 1), (2, 2, 2)))
    b = _vol((6, 6, 6), ((1, 1, 4), (2, 2, 5)))
    assert M.hd(a, b) == 3.0 and M.assd(a, b) == 3.0 and M.dc(a, b) == 0.0


def test_two_offset_cubes():
    """2x2x2 cubes 3 voxels apart along x: every voxel of such a cube is a surface voxel; the x = 1 layer of A is 3 away
    from B's nearest layer, the x = 2 layer 2 away -> directed mean 2.5, directed max 3, both directions alike"""
    a = _vol((8, 6, 6), ((1, 2, 2), (3, 4, 4)))
    b = _vol((8, 6, 6), ((4, 2, 2), (6, 4, 4)))
    assert M.hd(a, b) == 3.0
    assert M.assd(a, b) == pytest.approx(2.5, abs=1e-12)
    np.testing.assert_allclose(sorted(M.surface_distances(a, b)), [2.0] * 4 + [3.0] * 4)


def test_voxel_inside_cube():
    """A = centre voxel, B = 3x3x3 cube around it.  B's surface is its 26 shell voxels (the centre erodes away):
    A -> B is 1 (face neighbours); B -> A: 6 face voxels at 1, 12 edge voxels at sqrt 2, 8 corners at sqrt 3"""
    a = _vol((7, 7, 7), ((3, 3, 3), (4, 4, 4)))
    b = _vol((7, 7, 7), ((2, 2, 2), (5, 5, 5)))
    back = (6 * 1.0 + 12 * math.sqrt(2) + 8 * math.sqrt(3)) / 26
    assert M.hd(a, b) == pytest.approx(math.sqrt(3), abs=1e-12)
    assert M.assd(a, b) == pytest.approx(0.5 * (1.0 + back), abs=1e-12)
    assert M.dc(a, b) == pytest.approx(2 * 1 / (1 + 27))
    assert M.precision(a, b) == 1.0 and M.recall(a, b) == pytest.approx(1 / 27)
    assert M.specificity(a, b) == 1.0


def test_reference_call_shape_is_five_dimensional():
    """the reference hands MedPy the whole (B, 1, D, H, W) tensor (common/metrics.py:49-62): the connectivity-1 cross is
    then 5-dimensional, the extent-1 channel axis puts background on both sides of every voxel, so EVERY object voxel is a
    surface voxel (the cube's centre too) and A -> B becomes 0"""
    a = _vol((1, 1, 7, 7, 7), ((0, 0, 3, 3, 3), (1, 1, 4, 4, 4)))
    b = _vol((1, 1, 7, 7, 7), ((0, 0, 2, 2, 2), (1, 1, 5, 5, 5)))
    back = (0.0 + 6 * 1.0 + 12 * math.sqrt(2) + 8 * math.sqrt(3)) / 27
    assert M.hd(a, b) == pytest.approx(math.sqrt(3), abs=1e-12)
    assert M.assd(a, b) == pytest.approx(0.5 * (0.0 + back), abs=1e-12)
    # two samples in the batch are one voxel apart along the batch axis: an object in sample 0 and one in sample 1 at the
    # same position are at distance 1
    c = np.zeros((2, 1, 4, 4, 4), dtype=bool)
    d = np.zeros((2, 1, 4, 4, 4), dtype=bool)
    c[0, 0, 1, 1, 1] = True
    d[1, 0, 1, 1, 1] = True
    assert M.hd(c, d) == 1.0 and M.assd(c, d) == 1.0


def test_empty_masks_and_thresholding():
    z = np.zeros((4, 4, 4))
    t = np.zeros((4, 4, 4))
    t[1:3, 1:3, 1:3] = 1.0
    m = M.binary_measures(z, t)
    assert m["dc"] == 0.0 and m["precision"] == 0.0 and m["sensitivity"] == 0.0 and m["specificity"] == 1.0
    assert math.isinf(m["hd"]) and math.isinf(m["assd"])
    m = M.binary_measures(z, z)
    assert m["dc"] == 0.0 and math.isinf(m["hd"])
    with pytest.raises(RuntimeError):
        M.hd(z, t)
    # threshold 0.5 is exclusive (common/metrics.py:32-33: ``result > binary_threshold``)
    p = np.full((4, 4, 4), 0.5)
    assert M.binary_measures(p, t)["dc"] == 0.0
    p[1:3, 1:3, 1:3] = 0.51
    m = M.binary_measures(p, t)
    assert m["dc"] == 1.0 and m["hd"] == 0.0 and m["assd"] == 0.0
