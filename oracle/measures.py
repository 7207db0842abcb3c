"""CPU oracle of the evaluation measures (TEST INFRASTRUCTURE ONLY: imported by tests/ and nowhere in the product path).

The reference computes its batch metrics with the third-party package **MedPy 0.3.0** (``requirements.txt:5``;
``common/metrics.py:3,31-46``: ``medpy.metric.binary.{dc,hd,assd,precision,sensitivity,specificity}``), which is not
vendored in ``/root/reference`` and not installed in this image.  This file restates MedPy's published algorithm
(``medpy/metric/binary.py`` of release 0.3.0) on numpy + scipy.ndimage:

* ``dc``  = 2|A n B| / (|A| + |B|), 0.0 when both are empty;  ``precision`` = tp / (tp + fp), ``recall`` (= the reference's
  ``sensitivity``) = tp / (tp + fn), ``specificity`` = tn / (tn + fp); each 0.0 on a zero denominator.
* ``__surface_distances(result, reference, voxelspacing=None, connectivity=1)``: footprint =
  ``generate_binary_structure(result.ndim, connectivity)``; border = mask XOR ``binary_erosion(mask, footprint, iterations=1)``
  (border_value 0: voxels on the array edge are border voxels); ``dt = distance_transform_edt(~reference_border)``;
  returns ``dt[result_border]``.  Raises when an object is empty.
* ``hd``  = max(directed max A->B, directed max B->A);  ``assd`` = mean(directed mean A->B, directed mean B->A).

Parity status: MedPy cannot be run here, so these functions are pinned by hand-computed known answers derived from those
definitions (``tests/test_measures_oracle.py``: single voxels, offset cubes, voxel-in-cube, empty masks, and the
5-dimensional ``(B, 1, D, H, W)`` call shape of the reference, ``common/metrics.py:49-62``) -- parity with MedPy itself is
pinned by definition, not by running it.
"""
import numpy


def _as_bool(a):
    return numpy.atleast_1d(numpy.asarray(a).astype(bool))


def counts(result, reference):
    r, t = _as_bool(result), _as_bool(reference)
    return (float(numpy.count_nonzero(r & t)), float(numpy.count_nonzero(r & ~t)),
            float(numpy.count_nonzero(~r & t)), float(numpy.count_nonzero(~r & ~t)))


def dc(result, reference):
    tp, fp, fn, _ = counts(result, reference)
    size = (tp + fp) + (tp + fn)
    return 2.0 * tp / size if size > 0 else 0.0


def precision(result, reference):
    tp, fp, _, _ = counts(result, reference)
    return tp / (tp + fp) if tp + fp > 0 else 0.0


def recall(result, reference):
    tp, _, fn, _ = counts(result, reference)
    return tp / (tp + fn) if tp + fn > 0 else 0.0


sensitivity = recall


def specificity(result, reference):
    _, fp, _, tn = counts(result, reference)
    return tn / (tn + fp) if tn + fp > 0 else 0.0


def surface_distances(result, reference, connectivity=1):
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    r, t = _as_bool(result), _as_bool(reference)
    if not r.any():
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if not t.any():
        raise RuntimeError("The second supplied array does not contain any binary object.")
    footprint = generate_binary_structure(r.ndim, connectivity)
    rb = r ^ binary_erosion(r, structure=footprint, iterations=1)
    tb = t ^ binary_erosion(t, structure=footprint, iterations=1)
    return distance_transform_edt(~tb)[rb]


def hd(result, reference):
    return max(surface_distances(result, reference).max(), surface_distances(reference, result).max())


def asd(result, reference):
    return surface_distances(result, reference).mean()


def assd(result, reference):
    return numpy.mean((asd(result, reference), asd(reference, result)))


def binary_measures(result, target, binary_threshold=0.5):
    """``binary_measures_numpy`` of the reference (common/metrics.py:31-46) as a plain dict: threshold, the four overlap
    measures, and Hausdorff / ASSD only when both masks are non-empty (inf otherwise)."""
    r = numpy.asarray(result) > binary_threshold
    t = numpy.asarray(target) > binary_threshold
    out = dict(dc=dc(r, t), hd=numpy.inf, assd=numpy.inf, precision=precision(r, t), sensitivity=recall(r, t),
               specificity=specificity(r, t))
    if r.any() and t.any():
        out["hd"], out["assd"] = float(hd(r, t)), float(assd(r, t))
    return out
