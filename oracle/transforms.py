"""CPU oracle of the reference's sample transforms (TEST INFRASTRUCTURE ONLY: imported by tests/ and nowhere in the
product path).  Restates ``common/data.py:215-351`` of the reference on numpy + scipy.ndimage (the reference's own
third-party dependency for this path: ``scipy.ndimage.gaussian_filter`` / ``map_coordinates``, data.py:14-15).
Parity status: PINNED -- ``tests/golden/transforms.npz`` holds inputs and outputs of the reference's own classes
(``ElasticDeform`` incl. its ``__call__``, ``PadImages``, ``RandomPatch``, ``HemisphericFlip[FixedToCaseId]``, ``ToTensor``),
recorded by ``tests/golden/make_golden_transforms.py`` from the imported reference module; ``tests/test_transforms.py``
replays them through these functions.
"""
import random

import numpy as np
from scipy.ndimage import gaussian_filter, map_coordinates

KEY_IMAGES, KEY_LABELS, KEY_GLOBAL = "images", "labels", "clinical"


def elastic_transform(image, alpha=100, sigma=4, random_state=None):
    """data.py:326-339 (the random_state is always given here; the reference seeds a fresh one from the wall clock)."""
    shape = image.shape
    dx = gaussian_filter((random_state.rand(*shape) * 2 - 1), sigma, mode="constant", cval=0) * alpha
    dy = gaussian_filter((random_state.rand(*shape) * 2 - 1), sigma, mode="constant", cval=0) * alpha
    dz = gaussian_filter((random_state.rand(*shape) * 2 - 1), sigma, mode="constant", cval=0) * alpha * 0.22
    x, y, z = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]))
    indices = np.reshape(y + dy, (-1, 1)), np.reshape(x + dx, (-1, 1)), np.reshape(z + dz, (-1, 1))
    return map_coordinates(image, indices, order=1).reshape(shape), random_state


def elastic_deform(sample, alpha=100, sigma=4, apply_to_images=False, random_state=None):
    """ElasticDeform.__call__, data.py:341-351: one random state shared by all label (and image) channels."""
    sample[KEY_LABELS][:, :, :, 0], random_state = elastic_transform(sample[KEY_LABELS][:, :, :, 0], alpha, sigma, random_state)
    for c in range(1, sample[KEY_LABELS].shape[3]):
        sample[KEY_LABELS][:, :, :, c], _ = elastic_transform(sample[KEY_LABELS][:, :, :, c], alpha, sigma, random_state)
    if apply_to_images and len(sample[KEY_IMAGES]):
        for c in range(sample[KEY_IMAGES].shape[3]):
            sample[KEY_IMAGES][:, :, :, c], _ = elastic_transform(sample[KEY_IMAGES][:, :, :, c], alpha, sigma, random_state)
    return sample


def hemispheric_flip(sample, flip):
    """HemisphericFlip / HemisphericFlipFixedToCaseId, data.py:215-246 (the coin toss / case-id test is the argument)."""
    if not flip:
        return sample
    return {k: (np.flip(v, 0).copy() if k in (KEY_IMAGES, KEY_LABELS, KEY_GLOBAL) and len(v) else v) for k, v in sample.items()}


def random_patch(sample, w, h, d, pad, origin):
    """RandomPatch.__call__, data.py:259-277, with the three random offsets given."""
    rx, ry, rz = origin
    out = dict(sample)
    out[KEY_IMAGES] = sample[KEY_IMAGES][rx:rx + w, ry:ry + h, rz:rz + d, :]
    out[KEY_LABELS] = sample[KEY_LABELS][rx:rx + w - 2 * pad[0], ry:ry + h - 2 * pad[1], rz:rz + d - 2 * pad[2], :]
    return out


def pad_images(sample, pad, pad_value=0.0):
    """PadImages.__call__, data.py:288-296."""
    sx, sy, sz, sc = sample[KEY_IMAGES].shape
    out = dict(sample)
    img = np.ones((sx + 2 * pad[0], sy + 2 * pad[1], sz + 2 * pad[2], sc), dtype=np.float32) * float(pad_value)
    img[pad[0]:-pad[0], pad[1]:-pad[1], pad[2]:-pad[2], :] = sample[KEY_IMAGES]
    out[KEY_IMAGES] = img
    return out


def to_tensor_layout(a):
    """ToTensor, data.py:302-310: (x, y, z, c) -> (c, z, y, x)."""
    return np.transpose(a, (3, 2, 1, 0))
