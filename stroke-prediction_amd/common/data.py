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


class ResamplePlaneXY(object):
    """Down- or upsample every (x, y) slice (data.py:354-381): nearest neighbour (``ndi.zoom(order=0)``) or, with
    ``mode='bilinear'``, first order.  ``scipy.ndimage.zoom`` maps output index ``o`` to input coordinate
    ``o * (n_in - 1) / (n_out - 1)`` with ``n_out = round(n_in * factor)``; order 0 takes ``floor(c + 0.5)``, order 1
    interpolates between the two neighbours.  Index arithmetic on whatever the sample holds (device tensors or numpy)."""

    def __init__(self, scale_factor=1, mode='nearest'):
        self._scale_factor = scale_factor
        self._order = 1 if mode == 'bilinear' else 0

    def _axis(self, n_in, device):
        n_out = int(round(n_in * self._scale_factor))
        if n_out == n_in:
            return None
        c = torch.arange(n_out, dtype=torch.float64, device=device) * ((n_in - 1) / max(n_out - 1, 1))
        return c

    def _resample(self, vol):
        if self._scale_factor == 1:
            return vol
        was_numpy = isinstance(vol, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(vol)) if was_numpy else vol
        for axis in (0, 1):
            c = self._axis(t.shape[axis], t.device)
            if c is None:
                continue
            if self._order == 0:
                t = t.index_select(axis, torch.floor(c + 0.5).long().clamp_(0, t.shape[axis] - 1))
            else:
                lo = torch.floor(c).long().clamp_(0, t.shape[axis] - 1)
                hi = (lo + 1).clamp_(max=t.shape[axis] - 1)
                w = (c - lo.double()).to(t.dtype if t.is_floating_point() else torch.float32)
                shape = [1] * t.dim()
                shape[axis] = -1
                a, b = t.index_select(axis, lo).float(), t.index_select(axis, hi).float()
                t = (a + (b - a) * w.view(shape)).to(t.dtype if t.is_floating_point() else torch.float32)
        return t.numpy() if was_numpy else t

    def __call__(self, sample):
        result = emptyCopyFromSample(sample)
        result[KEY_GLOBAL] = sample[KEY_GLOBAL]
        for k in (KEY_IMAGES, KEY_LABELS):
            if _present(sample[k]):
                result[k] = self._resample(sample[k])
        return result


class Compose(object):
    """``torchvision.transforms.Compose`` for sample dicts (the reference's loader factories wrap their transform lists
    in it, data.py:121-124).  ``device``: numpy samples are uploaded once before the first transform (``to_device``), so
    the whole chain runs on device tensors; ``None`` leaves them where they are (host pipelines, CPU tests)."""

    def __init__(self, transforms, device=None):
        self.transforms = list(transforms)
        self.device = device

    def __call__(self, sample):
        if self.device is not None:
            sample = to_device(sample, self.device)
        for t in self.transforms:
            sample = t(sample)
        return sample


# ---------------------------------------------------------------------------------------------- datasets and loaders
# The reference reads a private 29-subject NIfTI + CSV data set from hard-wired paths (data.py:30-99).  The loader
# factories below keep its signatures (data.py:113-212) and the batch-dict contract; the samples come from
# ``StrokeLindaDataset3D`` when that data set is reachable (nibabel importable and the root directory present), and from
# ``SyntheticStrokeDataset3D`` -- deterministic blob volumes of the same shapes and channel meanings -- otherwise.

N_SYNTHETIC_CASES = 29


def _blob_field(rng, shape, sigma):
    """smoothed uniform noise in [0, 1], low resolution (cheap on the host)"""
    from scipy.ndimage import gaussian_filter
    f = gaussian_filter(rng.rand(*shape), sigma, mode="constant")
    f -= f.min()
    return f / max(float(f.max()), 1e-12)


def synthetic_sample(case_id, xy=256, z=28, n_modalities=2, n_labels=3, n_globals=5):
    """One case in the reference's sample layout: ``images`` (x, y, z, n_modalities), ``labels`` (x, y, z, n_labels) --
    nested binary blobs core < follow-up lesion < penumbra -- and ``clinical`` (1, 1, 1, n_globals) =
    (tO->tA, tA->tR, NIHSS, sex, age).  Deterministic in ``case_id``."""
    rng = np.random.RandomState(7919 + int(case_id))
    low = max(8, xy // 4)
    up = xy // low
    f = _blob_field(rng, (low, low, z), (3.0, 3.0, 2.0))
    thr = np.quantile(f, [0.93, 0.80, 0.86])                 # core, penumbra, lesion: core < lesion < penumbra as sets
    big = lambda a: np.repeat(np.repeat(a, up, axis=0), up, axis=1)
    labels = np.stack([big(f > thr[0]), big(f > thr[1]), big(f > thr[2])], axis=3)[..., :n_labels].astype(np.float32)
    imgs = [big(_blob_field(rng, (low, low, z), (2.0, 2.0, 1.0))) * s for s in (12.0, 40.0)][:n_modalities]
    images = np.stack(imgs, axis=3).astype(np.float32) if imgs else []
    clinical = np.array([rng.uniform(0.5, 4.0), rng.uniform(0.5, 5.0), float(rng.randint(0, 25)), float(rng.randint(0, 2)),
                         rng.uniform(40, 90)][:n_globals]).reshape((1, 1, 1, n_globals))
    return {KEY_CASE_ID: int(case_id), KEY_IMAGES: images, KEY_LABELS: labels, KEY_GLOBAL: clinical}


def synthetic_shape_batch(batch, d=28, hw=128, seed=0):
    """A ready batch for the CAE path (bench.py, smoke tests): ``labels`` (B, 3, d, hw, hw) binary blobs and
    ``clinical`` (B, 5, 1, 1, 1), i.e. ``ToTensor()``-layout samples of ``synthetic_sample`` stacked."""
    labels, clinical = [], []
    for b in range(batch):
        s = synthetic_sample(seed * 131 + b, xy=hw, z=d, n_modalities=0)
        labels.append(torch.from_numpy(s[KEY_LABELS]).permute(3, 2, 1, 0))
        clinical.append(torch.from_numpy(s[KEY_GLOBAL].astype(np.float32)).permute(3, 2, 1, 0))
    return torch.stack(labels).contiguous(), torch.stack(clinical).contiguous()


class SyntheticStrokeDataset3D(torch.utils.data.Dataset):
    """Stand-in for ``StrokeLindaDataset3D`` (data.py:30-99): same ``__getitem__`` contract, synthetic content."""

    def __init__(self, modalities=[], labels=[], transform=None, single_case_id=None, xy=256, z=28, n_cases=N_SYNTHETIC_CASES):
        self._modalities, self._labels, self._transform = modalities, labels, transform
        self._xy, self._z = xy, z
        self._item_index_map = [{KEY_CASE_ID: c, KEY_CLINICAL_IDX: c - 1} for c in range(1, n_cases + 1)
                                if single_case_id is None or single_case_id == c]

    def __len__(self):
        return len(self._item_index_map)

    def __getitem__(self, item):
        case_id = self._item_index_map[item][KEY_CASE_ID]
        result = synthetic_sample(case_id, self._xy, self._z, n_modalities=min(2, len(self._modalities)),
                                  n_labels=min(3, len(self._labels)) if self._labels else 0)
        if not self._labels:
            result[KEY_LABELS] = []
        if self._transform:
            result = self._transform(result)
        return result


class StrokeLindaDataset3D(torch.utils.data.Dataset):
    """The reference's NIfTI + CSV data set (data.py:30-99): ``<root>/<case>/train<case><suffix>.nii.gz`` volumes and one
    CSV row of clinical values per case.  Needs nibabel and the (private) data; see ``dataset_available``."""
    PATH_ROOT = '/share/data_zoe1/lucas/Linda_Segmentations'
    PATH_CSV = '/share/data_zoe1/lucas/Linda_Segmentations/clinical_cleaned.csv'

    def __init__(self, root_dir=PATH_ROOT, modalities=[], labels=[], clinical=PATH_CSV, transform=None, single_case_id=None):
        import csv
        self._root_dir, self._modalities, self._labels, self._transform = root_dir, modalities, labels, transform
        with open(clinical, 'r') as f:
            self._clinical = list(csv.reader(f, delimiter=','))[1:]            # one header row
        self._item_index_map = [{KEY_CASE_ID: int(row[0]), KEY_CLINICAL_IDX: i} for i, row in enumerate(self._clinical)
                                if single_case_id is None or single_case_id == int(row[0])]

    def _volume(self, case_id, suffix):
        import os
        import nibabel as nib
        fn = os.path.join(self._root_dir, '{1}/{0}{1}{2}.nii.gz'.format('train', str(case_id), suffix))
        return np.asarray(nib.load(fn).get_fdata())[:, :, :, np.newaxis]

    def __len__(self):
        return len(self._item_index_map)

    def __getitem__(self, item):
        entry = self._item_index_map[item]
        case_id = entry[KEY_CASE_ID]
        values = [float(v) for v in self._clinical[entry[KEY_CLINICAL_IDX]][1:]]
        result = {KEY_CASE_ID: case_id, KEY_IMAGES: [], KEY_LABELS: [], KEY_GLOBAL: []}
        if values:
            result[KEY_GLOBAL] = np.array(values).reshape((1, 1, 1, len(values)))
        for key, names in ((KEY_LABELS, self._labels), (KEY_IMAGES, self._modalities)):
            if names:
                result[key] = np.concatenate([self._volume(case_id, n) for n in names], axis=DIM_CHANNEL_NUMPY_3D)
        return self._transform(result) if self._transform else result


def dataset_available():
    import os
    if os.environ.get("SP_SYNTHETIC_DATA"):
        return False
    try:
        import nibabel  # noqa: F401
    except Exception:
        return False
    return os.path.isdir(StrokeLindaDataset3D.PATH_ROOT) and os.path.isfile(StrokeLindaDataset3D.PATH_CSV)


def _dataset(modalities, labels, transform_list, device):
    tf = Compose(transform_list, device=device)
    if dataset_available():
        return StrokeLindaDataset3D(modalities=modalities, labels=labels, transform=tf)
    return SyntheticStrokeDataset3D(modalities=modalities, labels=labels, transform=tf)


def _pipeline_device():
    """the transform chain runs on the GPU when there is one (ElasticDeform / PadImages are HIP / device ops)"""
    return "cuda" if torch.cuda.is_available() else None


def set_np_seed(workerid):
    np.random.seed(torch.initial_seed() % np.iinfo(np.int32).max)


def _loader(dataset, items, batch_size, num_workers, pin_memory, seeded):
    from torch.utils.data import DataLoader
    from torch.utils.data.sampler import SubsetRandomSampler
    if getattr(getattr(dataset, "transform", None), "device", None) is not None:
        num_workers = 0      # the transform chain uploads and runs HIP kernels: a forked worker cannot re-initialise the GPU
    return DataLoader(dataset, batch_size=batch_size, sampler=SubsetRandomSampler(items), num_workers=num_workers,
                      pin_memory=pin_memory, worker_init_fn=set_np_seed if seeded else None)


def _fold_items(dataset, indices, shuffle, random_seed):
    items = sorted(set(range(len(dataset))) & set(indices))
    if shuffle:
        np.random.RandomState(random_seed).shuffle(items)
    return items


def split_data_loader3D(modalities, labels, indices, batch_size, random_seed=None, valid_size=0.5, shuffle=True,
                        num_workers=4, pin_memory=False, train_transform=[], valid_transform=[]):
    """data.py:113-147: one fold -> (training loader, validation loader); the first ``valid_size`` share of the
    (seed-shuffled) fold validates."""
    assert 0 <= valid_size <= 1, "[!] valid_size should be in the range [0, 1]."
    assert train_transform and valid_transform, "You must provide at least a numpy-to-torch transformation."
    dev = _pipeline_device()
    ds_train, ds_valid = _dataset(modalities, labels, train_transform, dev), _dataset(modalities, labels, valid_transform, dev)
    items = _fold_items(ds_train, indices, shuffle, random_seed)
    split = int(np.floor(valid_size * len(items)))
    return (_loader(ds_train, items[split:], batch_size, num_workers, pin_memory, True),
            _loader(ds_valid, items[:split], batch_size, num_workers, pin_memory, False))


def single_data_loader3D(modalities, labels, indices, batch_size, random_seed=None, valid_size=0.5, shuffle=True,
                         num_workers=4, pin_memory=False, train_transform=[]):
    """data.py:150-172."""
    assert train_transform, "You must provide at least a numpy-to-torch transformation."
    ds = _dataset(modalities, labels, train_transform, _pipeline_device())
    return _loader(ds, _fold_items(ds, indices, shuffle, random_seed), batch_size, num_workers, pin_memory, True)


def get_stroke_shape_training_data(modalities, labels, train_transform, valid_transform, fold_indices, ratio, seed=4,
                                   batchsize=2, split=True):
    """data.py:175-182 (``num_workers=0``: the transforms run in the training process -- here on its GPU)."""
    if split:
        return split_data_loader3D(modalities, labels, fold_indices, batchsize, random_seed=seed, valid_size=ratio,
                                   train_transform=train_transform, valid_transform=valid_transform, num_workers=0)
    return single_data_loader3D(modalities, labels, fold_indices, batchsize, random_seed=seed, valid_size=ratio,
                                train_transform=train_transform, num_workers=0), None


get_stroke_prediction_training_data = get_stroke_shape_training_data      # data.py:185-192: the same factory


def get_testdata(modalities, labels, indices, random_seed=None, shuffle=True, num_workers=4, pin_memory=False, transform=[]):
    """data.py:195-212: batch size 1 (the case metrics are computed per batch)."""
    assert transform, "You must provide at least a numpy-to-torch transformation."
    ds = _dataset(modalities, labels, transform, _pipeline_device())
    return _loader(ds, _fold_items(ds, indices, shuffle, random_seed), 1, 0 if _pipeline_device() else num_workers, pin_memory, True)
