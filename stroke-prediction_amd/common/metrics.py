"""Loss and evaluation measures (reference ``common/metrics.py``).

``BatchDiceLoss`` (metrics.py:8-28) keeps its constructor and call signature; on GPU tensors the three
whole-batch reductions and the backward run as two fused HIP kernels (``sp_dice_sums``, ``sp_dice_bwd``)
instead of six torch reductions plus temporaries.  The binary measures (metrics.py:31-62) are restated
on numpy/scipy because medpy is not a dependency here: Dice, precision, sensitivity, specificity and
the surface distances HD / ASSD (definitions of ``medpy.metric.binary`` 0.3.0).
"""
import numpy
import torch
from torch.nn.modules.loss import _Loss as LossModule

from common.dto.MetricMeasuresDto import BinaryMeasuresDto


_W_CACHE = {}


def _weights_on(device, weights):
    key = (str(device), weights)
    if key not in _W_CACHE:
        _W_CACHE[key] = torch.tensor(weights, dtype=torch.float64, device=device)
    return _W_CACHE[key]


class _DiceFn(torch.autograd.Function):
    """loss = 1 - sum_c w_c (2 I_c + eps) / (O_c + T_c + eps); sums over batch and volume per channel."""

    @staticmethod
    def forward(ctx, outputs, targets, weights, eps):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        o = outputs.contiguous().float()
        t = targets.contiguous().float()
        B, C = o.shape[0], o.shape[1]
        dhw = o.numel() // (B * C)
        sums = torch.zeros(C, 3, dtype=torch.float64, device=o.device)
        L.call("sp_dice_sums", O.ptr(o), O.ptr(t), B, C, dhw, O.ptr(sums), O.stream())
        from stroke_prediction_amd.runtime.layers import SYNC, _allreduce
        if SYNC["on"]:                  # Dice is a ratio of WHOLE-batch sums (metrics.py:24-27): make them global
            _allreduce(sums)
        w = _weights_on(o.device, weights)      # cached: no host->device copy inside a (graph-captured) step
        num = 2.0 * sums[:, 0] + eps
        den = sums[:, 1] + sums[:, 2] + eps
        ctx.save_for_backward(o, t, w, num, den)
        return (1.0 - (w * num / den).sum()).float()

    @staticmethod
    def backward(ctx, gloss):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        o, t, w, num, den = ctx.saved_tensors
        B, C = o.shape[0], o.shape[1]
        up = gloss.double()
        ca = (-2.0 * w / den * up).float().contiguous()
        cb = (2.0 * w * num / (den * den) * up).float().contiguous()
        d = torch.empty_like(o)
        L.call("sp_dice_bwd", O.ptr(o), O.ptr(t), O.ptr(ca), O.ptr(cb), B, C, o.numel() // (B * C), O.ptr(d), O.stream())
        return d, None, None, None


class BatchDiceLoss(LossModule):
    def __init__(self, label_weights, epsilon=0.0000001, dim=1):
        super(BatchDiceLoss, self).__init__()
        self._epsilon = epsilon
        self._dim = dim
        self._label_weights = label_weights
        print("DICE Loss weights classes' output by", label_weights)

    def forward(self, outputs, targets):
        assert targets.shape[self._dim] == len(self._label_weights), \
            'Ground truth number of labels does not match with label weight vector'
        assert outputs.shape == targets.shape
        if not outputs.is_cuda or self._dim != 1:
            raise RuntimeError("BatchDiceLoss (stroke_prediction_amd) runs on the GPU with channel dim 1 only")
        return _DiceFn.apply(outputs, targets, tuple(float(w) for w in self._label_weights), float(self._epsilon))


# ---------------------------------------------------------------------------------------------- evaluation measures

def _surface_distances(result, reference):
    """Distances from the border voxels of ``result`` to the border of ``reference`` (medpy definition)."""
    from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure
    footprint = generate_binary_structure(result.ndim, 1)
    rb = result ^ binary_erosion(result, structure=footprint, iterations=1)
    fb = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    dt = distance_transform_edt(~fb)
    return dt[rb]


def _hd(a, b):
    return max(_surface_distances(a, b).max(), _surface_distances(b, a).max())


def _assd(a, b):
    return numpy.mean((_surface_distances(a, b).mean(), _surface_distances(b, a).mean()))


def _measures_from_counts(tp, fp, fn, tn):
    size = (tp + fp) + (tp + fn)
    return BinaryMeasuresDto(2.0 * tp / size if size > 0 else 0.0, numpy.inf, numpy.inf,
                             tp / (tp + fp) if tp + fp > 0 else 0.0,
                             tp / (tp + fn) if tp + fn > 0 else 0.0,
                             tn / (tn + fp) if tn + fp > 0 else 0.0)


def binary_measures_numpy(result, target, binary_threshold=0.5, distances=True):
    r = result > binary_threshold
    t = target > binary_threshold
    out = _measures_from_counts(float(numpy.count_nonzero(r & t)), float(numpy.count_nonzero(r & ~t)),
                                float(numpy.count_nonzero(~r & t)), float(numpy.count_nonzero(~r & ~t)))
    if distances and r.any() and t.any():
        out.hd = _hd(r, t)
        out.assd = _assd(r, t)
    return out


DISTANCE_METRICS = True      # Hausdorff / ASSD (CPU distance transforms, as medpy does); False keeps the metrics on the GPU


def binary_measures_torch(result, target, cuda, binary_threshold=0.5, distances=None):
    """``metrics.py:48-62`` of the reference.  Tensors on the GPU: the four confusion counts come from one HIP reduction
    (``sp_confusion_counts``, 32 bytes to the host); the volumes are copied to the host only for the surface distances,
    and only when ``distances`` (default: module flag ``DISTANCE_METRICS``) asks for them."""
    distances = DISTANCE_METRICS if distances is None else distances
    if isinstance(result, torch.Tensor) and isinstance(target, torch.Tensor) and result.is_cuda and target.is_cuda:
        from stroke_prediction_amd.runtime import lib as L, ops as O
        r = result.detach().float().contiguous()
        t = target.detach().float().contiguous()
        counts = torch.zeros(4, dtype=torch.int64, device=r.device)
        L.call("sp_confusion_counts", O.ptr(r), O.ptr(t), float(binary_threshold), r.numel(), O.ptr(counts), O.stream())
        tp, fp, fn, tn = (float(v) for v in counts.tolist())
        out = _measures_from_counts(tp, fp, fn, tn)
        if distances and tp + fp > 0 and tp + fn > 0:
            rn, tn_ = r.cpu().numpy() > binary_threshold, t.cpu().numpy() > binary_threshold
            out.hd = _hd(rn, tn_)
            out.assd = _assd(rn, tn_)
        return out
    result = result.detach().cpu().numpy() if isinstance(result, torch.Tensor) else result
    target = target.detach().cpu().numpy() if isinstance(target, torch.Tensor) else target
    return binary_measures_numpy(result, target, binary_threshold=binary_threshold, distances=distances)
