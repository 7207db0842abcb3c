"""Batch-dict contract and sample transforms of the reference data layer (``common/data.py``), on the GPU.

The NIfTI dataset itself stays out of scope (private data, nibabel); what is mirrored here is what sits between a
loaded sample and the network (SURVEY.md 8 "next" row N4): the transform classes of ``common/data.py:215-351`` with the
same names, constructor arguments and sample-dict semantics, operating on **device tensors**.  A sample is the
reference's dict -- ``images`` / ``labels`` as ``(x, y, z, c)`` arrays, ``clinical`` -- whose arrays are
``torch.cuda`` fp32 tensors (``to_device(sample)`` uploads a numpy sample once); every transform returns device
tensors, so a pipeline ``Compose([HemisphericFlip(), ElasticDeform(), ToTensor()])`` never leaves the GPU.  Flip,
patch, pad and the layout permutation are index arithmetic (torch views + one copy); the elastic deformation runs on the
HIP kernels ``sp_gaussian_filter3d`` / ``sp_map_coordinates_linear`` (csrc/sp_transform.hip).  Random decisions come
from the same host generators as in the reference (``random`` / ``numpy.random.RandomState``), so a seeded pipeline
reproduces the reference's augmentation; ``ElasticDeform(device_noise=True)`` draws the noise on the device instead.
"""
import datetime
import random

import numpy as np
import torch

KEY_CASE_ID = 'case_id'
KEY_CLINICAL_IDX = 'clinical_idx'
KEY_IMAGES = 'images'
KEY_LABELS = 'labels'
KEY_GLOBAL = 'clinical'

DIM_HORIZONTAL_NUMPY_3D = 0
DIM_DEPTH_NUMPY_3D = 2
DIM_CHANNEL_NUMPY_3D = 3
DIM_CHANNEL_TORCH3D_5 = 1     # tensors are B x C x D x H x W


def _present(v):
    """The reference marks a missing entry with ``[]`` (data.py:102-105)."""
    if isinstance(v, torch.Tensor):
        return v.numel() > 0
    if isinstance(v, np.ndarray):
        return v.size > 0
    return False


def emptyCopyFromSample(sample):
    """data.py:102-105."""
    result = {KEY_CASE_ID: int(sample[KEY_CASE_ID]), KEY_CLINICAL_IDX: sample.get(KEY_CLINICAL_IDX, 0),
              KEY_IMAGES: [], KEY_LABELS: [], KEY_GLOBAL: []}
    return result


def to_device(sample, device="cuda"):
    """Upload the arrays of a numpy sample (fp32) -- the one host-to-device copy of the pipeline."""
    out = dict(sample)
    for k in (KEY_IMAGES, KEY_LABELS, KEY_GLOBAL):
        v = sample.get(k, [])
        if isinstance(v, np.ndarray) and v.size:
            out[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(device)
    return out


def _require_cuda(t, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("%s (stroke_prediction_amd) works on CUDA tensors; upload the sample with to_device() first" % what)


class HemisphericFlipFixedToCaseId(object):
    """Flip along the X axis for case ids above ``split_id`` (data.py:215-231)."""

    def __init__(self, split_id):
        self.split_id = split_id

    def __call__(self, sample):
        if int(sample[KEY_CASE_ID]) > self.split_id:
            return _flip(sample)
        return sample


class HemisphericFlip(object):
    """Flip along the X axis with probability 1/2 (data.py:234-246; ``random.random()`` like the reference)."""

    def __call__(self, sample):
        if random.random() > 0.5:
            return _flip(sample)
        return sample


def _flip(sample):
    result = emptyCopyFromSample(sample)
    for k in (KEY_IMAGES, KEY_LABELS, KEY_GLOBAL):
        if _present(sample[k]):
            _require_cuda(sample[k], "HemisphericFlip")
            result[k] = torch.flip(sample[k], (DIM_HORIZONTAL_NUMPY_3D,))
    return result


class RandomPatch(object):
    """Random patches of a certain size; labels cropped by the network's padding (data.py:249-277)."""

    def __init__(self, w, h, d, pad_x, pad_y, pad_z):
        self._padx, self._pady, self._padz = pad_x, pad_y, pad_z
        self._w, self._h, self._d = w, h, d

    def __call__(self, sample):
        sx, sy, sz, _ = sample[KEY_IMAGES].shape
        rand_x = random.randint(0, sx - self._w)
        rand_y = random.randint(0, sy - self._h)
        rand_z = random.randint(0, sz - self._d)
        result = emptyCopyFromSample(sample)
        if _present(sample[KEY_IMAGES]):
            result[KEY_IMAGES] = sample[KEY_IMAGES][rand_x: rand_x + self._w, rand_y: rand_y + self._h,
                                                    rand_z: rand_z + self._d, :]
        if _present(sample[KEY_LABELS]):
            result[KEY_LABELS] = sample[KEY_LABELS][rand_x: rand_x + self._w - 2 * self._padx,
                                                    rand_y: rand_y + self._h - 2 * self._pady,
                                                    rand_z: rand_z + self._d - 2 * self._padz, :]
        result[KEY_GLOBAL] = sample[KEY_GLOBAL]
        return result


class PadImages(object):
    """Pad images with a constant in all 6 directions (data.py:280-296)."""

    def __init__(self, pad_x, pad_y, pad_z, pad_value=0):
        self._padx, self._pady, self._padz = pad_x, pad_y, pad_z
        self._pad_value = float(pad_value)

    def __call__(self, sample):
        result = emptyCopyFromSample(sample)
        if _present(sample[KEY_IMAGES]):
            img = sample[KEY_IMAGES]
            _require_cuda(img, "PadImages")
            sx, sy, sz, sc = img.shape
            out = torch.full((sx + 2 * self._padx, sy + 2 * self._pady, sz + 2 * self._padz, sc), self._pad_value,
                             dtype=torch.float32, device=img.device)
            out[self._padx:sx + self._padx, self._pady:sy + self._pady, self._padz:sz + self._padz, :] = img
            result[KEY_IMAGES] = out
        result[KEY_LABELS] = sample[KEY_LABELS]
        result[KEY_GLOBAL] = sample[KEY_GLOBAL]
        return result


class ToTensor(object):
    """(x, y, z, c) -> (c, z, y, x) (data.py:299-310); the arrays already are tensors here, the permutation is a view."""

    def __call__(self, sample):
        result = emptyCopyFromSample(sample)
        for k in (KEY_IMAGES, KEY_LABELS, KEY_GLOBAL):
            if _present(sample[k]):
                v = sample[k] if isinstance(sample[k], torch.Tensor) else torch.from_numpy(sample[k])
                result[k] = v.permute(3, 2, 1, 0)
        return result


class ElasticDeform(object):
    """Elastic deformation [Simard2003] of the label (and optionally image) channels (data.py:313-351): three
    Gaussian-smoothed uniform noise fields scaled by alpha (the third by 0.22 * alpha) displace the sampling grid,
    first-order interpolation, zero outside.  Noise: ``random_state.rand`` on the host exactly like the reference (one
    5.5 MB upload per channel at 128 x 128 x 28), or ``device_noise=True``: torch's device generator."""

    def __init__(self, alpha=100, sigma=4, apply_to_images=False, device_noise=False):
        self._alpha = alpha
        self._sigma = sigma
        self._apply_to_images = apply_to_images
        self._device_noise = device_noise

    def _smoothed_noise(self, shape, sigma, random_state, device, tmp):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        if self._device_noise:
            noise = torch.rand(shape, dtype=torch.float32, device=device) * 2 - 1
        else:
            noise = torch.from_numpy((random_state.rand(*shape) * 2 - 1).astype(np.float32)).to(device)
        field = torch.empty_like(noise)
        L.call("sp_gaussian_filter3d", O.ptr(noise), O.ptr(field), O.ptr(tmp), shape[0], shape[1], shape[2], float(sigma), 4.0,
               O.stream())
        return field

    def elastic_transform(self, image, alpha=100, sigma=4, random_state=None):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        _require_cuda(image, "ElasticDeform")
        new_seed = datetime.datetime.now().second + datetime.datetime.now().microsecond
        if random_state is None:
            random_state = np.random.RandomState(new_seed)
        shape = tuple(image.shape)
        if len(shape) != 3 or shape[0] != shape[1]:
            # the reference adds meshgrid(indexing='xy') grids of shape (s1, s0, s2) to fields of shape (s0, s1, s2)
            raise ValueError("elastic_transform needs a (n, n, d) volume (data.py:336-337), got %r" % (shape,))
        img = image.contiguous().float()
        tmp = torch.empty_like(img)
        dx = self._smoothed_noise(shape, sigma, random_state, img.device, tmp)
        dy = self._smoothed_noise(shape, sigma, random_state, img.device, tmp)
        dz = self._smoothed_noise(shape, sigma, random_state, img.device, tmp)
        out = torch.empty_like(img)
        # indices = (y + dy, x + dx, z + dz) with x, y, z = meshgrid(...) in 'xy' indexing: y is the axis-0 index and x
        # the axis-1 index -- axis 0 is displaced by the SECOND field, axis 1 by the first (data.py:336-337)
        L.call("sp_map_coordinates_linear", O.ptr(img), O.ptr(dy), O.ptr(dx), O.ptr(dz), float(alpha), float(alpha),
               float(alpha) * 0.22, 0.0, O.ptr(out), shape[0], shape[1], shape[2], O.stream())
        return out, random_state

    def __call__(self, sample):
        labels = sample[KEY_LABELS]
        _require_cuda(labels, "ElasticDeform")
        if not labels.is_contiguous():
            labels = sample[KEY_LABELS] = labels.contiguous()
        res, random_state = self.elastic_transform(labels[:, :, :, 0], self._alpha, self._sigma)
        labels[:, :, :, 0] = res
        for c in range(1, labels.shape[3]):
            labels[:, :, :, c], _ = self.elastic_transform(labels[:, :, :, c], self._alpha, self._sigma,
                                                           random_state=random_state)
        if self._apply_to_images and _present(sample[KEY_IMAGES]):
            images = sample[KEY_IMAGES]
            if not images.is_contiguous():
                images = sample[KEY_IMAGES] = images.contiguous()
            for c in range(images.shape[3]):
                images[:, :, :, c], _ = self.elastic_transform(images[:, :, :, c], self._alpha, self._sigma,
                                                               random_state=random_state)
        return sample
