"""Records tests/golden/transforms.npz from the REFERENCE's own transform classes (common/data.py:215-351:
``ElasticDeform``, ``PadImages``, ``RandomPatch``, ``HemisphericFlip[FixedToCaseId]``, ``ToTensor``), imported unmodified in
the build container with the same empty stand-in modules as make_golden.py for the absent optional imports
(nibabel, torchvision).  Inputs are small seeded volumes; the host random generators the reference draws from
(``random``, ``numpy.random.RandomState``) are seeded and the seeds recorded, so the oracle (oracle/transforms.py) and the
device pipeline can replay every case.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_transforms.py
"""
import os
import random
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402

import_reference()
warnings.simplefilter("ignore")
from common import data as R  # noqa: E402  (the reference's common/data.py)


def blobs(shape, seed):
    rs = np.random.RandomState(seed)
    g = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).astype(np.float64)
    img = np.zeros(shape)
    for _ in range(4):
        c = rs.rand(3) * np.array(shape)
        r = 2.0 + rs.rand() * min(shape) / 4
        img += np.exp(-((g - c) ** 2).sum(-1) / (2 * r * r))
    return img


out = {"generator": np.array("reference common/data.py via tests/golden/make_golden_transforms.py")}
# ---- ElasticDeform.elastic_transform (data.py:326-339), the method the __call__ loops over channels
ed = R.ElasticDeform()
for name, shape, alpha, sigma, seed in (("a", (24, 24, 10), 30.0, 2.0, 7), ("b", (16, 16, 6), 12.0, 1.5, 11)):
    img = blobs(shape, seed)
    lab = (img > 0.5).astype(np.float64)
    for kind, arr in (("smooth", img), ("binary", lab)):
        res, _ = ed.elastic_transform(arr.copy(), alpha, sigma, np.random.RandomState(seed + 100))
        out["%s_%s_in" % (name, kind)] = arr.astype(np.float32)
        out["%s_%s_out" % (name, kind)] = res.astype(np.float32)
    out["%s_params" % name] = np.array([alpha, sigma, seed + 100], dtype=np.float64)

# ---- ElasticDeform.__call__ on a whole sample: one RandomState threads through the label channels (data.py:341-351).
# Its seed comes from the wall clock (data.py:327-329): the clock is pinned for the call and the seed it yields recorded.
rs = np.random.RandomState(3)
sample = {R.KEY_CASE_ID: 17, R.KEY_IMAGES: rs.rand(20, 20, 8, 2).astype(np.float32),
          R.KEY_LABELS: np.stack([(blobs((20, 20, 8), 40 + c) > 0.5) for c in range(3)], -1).astype(np.float32),
          R.KEY_GLOBAL: rs.rand(1, 1, 1, 5)}
out["s_images"], out["s_labels"], out["s_clinical"] = sample[R.KEY_IMAGES].copy(), sample[R.KEY_LABELS].copy(), sample[R.KEY_GLOBAL].copy()


class _Clock:
    """datetime.datetime.now() -> a fixed instant (second 12, microsecond 345), i.e. seed 357"""
    class datetime:
        @staticmethod
        def now():
            import datetime as _d
            return _d.datetime(2018, 1, 1, 0, 0, 12, 345)


real_dt = R.datetime
R.datetime = _Clock
try:
    lab = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in sample.items()}
    res = R.ElasticDeform(alpha=20, sigma=2)(lab)
finally:
    R.datetime = real_dt
out["s_elastic_labels"] = res[R.KEY_LABELS].astype(np.float32)
out["s_elastic_seed"] = np.array(12 + 345)
out["s_elastic_params"] = np.array([20.0, 2.0])


class Np113(np.ndarray):
    """The reference tests for a missing entry with ``array != []`` (data.py:222-226, 262-268, 287...).  Under its pinned
    numpy 1.13 (requirements.txt:2) that comparison evaluates to a plain ``True``; numpy >= 1.25 raises a broadcast error
    instead.  The sample arrays are handed over as this subclass, which restores exactly that one comparison -- the
    reference classes themselves run unmodified."""

    def __ne__(self, other):
        if isinstance(other, list) and len(other) == 0:
            return True
        return np.ndarray.__ne__(self, other)


def call(tf, s):
    return tf({k: (v.copy().view(Np113) if isinstance(v, np.ndarray) else v) for k, v in s.items()})


status = {}
for name, tf, seed in (("flip_fixed", R.HemisphericFlipFixedToCaseId(split_id=15), None),
                       ("flip_random", R.HemisphericFlip(), 5),
                       ("pad", R.PadImages(2, 3, 1, pad_value=7), None),
                       ("patch", R.RandomPatch(12, 10, 6, 2, 3, 1), 9),
                       ("totensor", R.ToTensor(), None)):
    if seed is not None:
        random.seed(seed)
    try:
        r = call(tf, sample)
        status[name] = "ok"
        for k in (R.KEY_IMAGES, R.KEY_LABELS, R.KEY_GLOBAL):
            v = r[k]
            if hasattr(v, "numpy"):
                v = v.numpy()
            if isinstance(v, np.ndarray) and v.size:
                out["s_%s_%s" % (name, k)] = np.ascontiguousarray(np.asarray(v)).astype(np.float32)
        if seed is not None:
            out["s_%s_seed" % name] = np.array(seed)
    except Exception as e:          # recorded, not hidden: the class cannot run under this numpy
        status[name] = "%s: %s" % (type(e).__name__, str(e)[:80])
out["status"] = np.array(repr(status))
np.savez_compressed(os.path.join(HERE, "transforms.npz"), **out)
print(status)
print({k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.shape})
