"""Records tests/golden/transforms.npz: inputs and expected outputs of the reference's elastic deformation
(common/data.py:326-339) evaluated with scipy.ndimage -- the reference's own dependency for this path; common/data.py
itself cannot be imported in this image (nibabel is absent).  Run from the repo root: python tests/golden/make_golden_transforms.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import transforms as T  # noqa: E402


def blobs(shape, seed):
    rs = np.random.RandomState(seed)
    g = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).astype(np.float64)
    img = np.zeros(shape)
    for _ in range(4):
        c = rs.rand(3) * np.array(shape)
        r = 2.0 + rs.rand() * min(shape) / 4
        img += np.exp(-((g - c) ** 2).sum(-1) / (2 * r * r))
    return img


out = {}
for name, shape, alpha, sigma, seed in (("a", (24, 24, 10), 30.0, 2.0, 7), ("b", (16, 16, 6), 12.0, 1.5, 11)):
    img = blobs(shape, seed)
    lab = (img > 0.5).astype(np.float64)
    for kind, arr in (("smooth", img), ("binary", lab)):
        res, _ = T.elastic_transform(arr.copy(), alpha, sigma, np.random.RandomState(seed + 100))
        out["%s_%s_in" % (name, kind)] = arr.astype(np.float32)
        out["%s_%s_out" % (name, kind)] = res.astype(np.float32)
    out["%s_params" % name] = np.array([alpha, sigma, seed + 100], dtype=np.float64)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "transforms.npz"), **out)
print({k: v.shape for k, v in out.items()})
