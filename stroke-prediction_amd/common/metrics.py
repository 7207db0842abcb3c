"""Loss and evaluation measures (reference ``common/metrics.py``).

``BatchDiceLoss`` (metrics.py:8-28) keeps its constructor and call signature; on GPU tensors the three
whole-batch reductions and the backward run as two fused HIP kernels (``sp_dice_sums``, ``sp_dice_bwd``)
instead of six torch reductions plus temporaries.  The binary measures (metrics.py:31-62: Dice, precision,
sensitivity, specificity, Hausdorff distance and ASSD as ``medpy.metric.binary`` 0.3.0 defines them) run on the device:
``sp_confusion_counts`` and ``sp_surface_distances`` (border extraction + exact Euclidean distance transform).  There is
no host implementation here; the checker is ``oracle/measures.py``.
"""
import numpy
import torch
from torch.nn.modules.loss import _Loss as LossModule

from common.dto.MetricMeasuresDto import BinaryMeasuresDto


_W_CACHE = {}


def _weights_on(device, weights):
    key = (str(device), weights)
    if key not in _W_CACHE:
        _W_CACHE[key] = torch.tensor(weights, dtype=torch.float32, device=device)
    return _W_CACHE[key]


def _batch_strided(x):
    """(tensor, batch stride in elements) for a (B, C, ...) fp32 tensor that is contiguous within each sample --
    channel-slice views such as ``dto.outputs.core`` qualify and are read in place; anything else is copied."""
    x = x if x.dtype == torch.float32 else x.float()
    inner = 1
    ok = True
    for size, stride in zip(reversed(x.shape[1:]), reversed(x.stride()[1:])):
        if size != 1 and stride != inner:
            ok = False
            break
        inner *= size
    if not ok or (x.shape[0] > 1 and x.stride(0) < inner):
        x = x.contiguous()
        return x, inner
    return x, (x.stride(0) if x.shape[0] > 1 else inner)


class _GlobalMeanFn(torch.autograd.Function):
    """mean over the GLOBAL batch in the exact data-parallel mode: local sum, all-reduce, / (elements x world); the backward
    hands every local element 1 / (elements x world) of the upstream gradient -- the local backward then yields this rank's
    share of the gradient of the global mean, and the SUM of the flat gradient buffers is the whole-batch gradient."""

    @staticmethod
    def forward(ctx, t, world):
        from stroke_prediction_amd.runtime.layers import _allreduce
        s = t.sum(dtype=torch.float64).reshape(1)
        _allreduce(s)
        ctx.n, ctx.shape = t.numel() * world, t.shape
        return (s[0] / ctx.n).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.n).expand(ctx.shape), None


def batch_mean(t):
    """``torch.mean`` of a per-sample tensor as the reference's single process sees it (the loss recipes of
    CaeReconstructionLearner.py:58-67 average over the whole batch): the global mean when the exact data-parallel mode is on
    (parallel.DataParallelSync(mode="exact")), the plain mean otherwise."""
    from stroke_prediction_amd.runtime.layers import SYNC
    if SYNC["on"] and SYNC["world"] > 1:
        return _GlobalMeanFn.apply(t, SYNC["world"])
    return torch.mean(t)


_DICE_SUMS = {}      # (device, C, stream) -> [replica rows of the Dice sums (zero between calls), busy]


class _DiceFn(torch.autograd.Function):
    """loss = 1 - sum_c w_c (2 I_c + eps) / (O_c + T_c + eps); sums over batch and volume per channel.  Three HIP
    launches forward (sums, finalize) and one backward; the scalar algebra never leaves the device."""

    @staticmethod
    def forward(ctx, outputs, targets, weights, eps):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        o, obs = _batch_strided(outputs)
        t, tbs = _batch_strided(targets)
        B, C = o.shape[0], o.shape[1]
        dhw = o.numel() // (B * C)
        # ONE accumulator per (device, C), zeroed once: sp_dice_finalize_clear leaves it zero again (no fill launch per call -- 5 us of a
        # captured step's dependent chain).  ``busy``: a call that died between the two launches left sums behind -> zero them here.
        key = (o.device, C, int(torch.cuda.current_stream(o.device).cuda_stream))      # (per stream: launches of one stream are ordered)
        ent = _DICE_SUMS.get(key)
        if ent is None or ent[1]:
            ent = _DICE_SUMS[key] = [torch.zeros(L.SP_REDUCE_ROWS, (3 * C + 15) // 16 * 16, dtype=torch.float64, device=o.device), False]
        sums = ent[0]   # replica rows
        ent[1] = True
        L.call("sp_dice_sums", O.ptr(o), obs, O.ptr(t), tbs, B, C, dhw, O.ptr(sums), O.stream())
        from stroke_prediction_amd.runtime.layers import SYNC, _allreduce
        if SYNC["on"]:                  # Dice is a ratio of WHOLE-batch sums (metrics.py:24-27): make them global
            _allreduce(sums)
        w = _weights_on(o.device, weights)      # cached: no host->device copy inside a (graph-captured) step
        loss = torch.empty((), dtype=torch.float32, device=o.device)
        coef = torch.empty(2 * C, dtype=torch.float32, device=o.device)
        L.call("sp_dice_finalize_clear", O.ptr(sums), O.ptr(w), float(eps), C, O.ptr(loss), O.ptr(coef), O.stream())
        ent[1] = False
        ctx.save_for_backward(o, t, coef)
        ctx.strides = (obs, tbs)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        o, t, coef = ctx.saved_tensors
        obs, tbs = ctx.strides
        B, C = o.shape[0], o.shape[1]
        up = gloss if (gloss.dtype == torch.float32 and gloss.is_contiguous()) else gloss.float().contiguous()
        d = torch.empty(o.shape, dtype=torch.float32, device=o.device)
        L.call("sp_dice_bwd", O.ptr(o), obs, O.ptr(t), tbs, O.ptr(coef), O.ptr(up), B, C, o.numel() // (B * C), O.ptr(d),
               O.stream())
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


class _CaeLossFn(torch.autograd.Function):
    """CaeReconstructionLearner.loss_step (reference :52-70) as three HIP launches (sp_cae_loss_fwd / _bwd) instead of ~60 torch and
    Dice kernels: [ mean(|p-i|-(p-i)) + mean(|p-c|-(p-c)) + Dice(c) + Dice(p) + Dice(l) + f mean|zi - zl| ] / (5 + f).  The four
    gradients come back as consecutive slices of ONE tensor in the order the reconstructions lie in memory, so that a decoder
    call that produced them stacked on the batch axis (Cae3D._StackManyFn) takes the buffer as it is."""

    @staticmethod
    def forward(ctx, c, p, l, i, tc, tp, tl, zi, zl, factor, weight, eps):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        recs = [_batch_strided(t) for t in (c, p, l, i)]
        gts = [_batch_strided(t) for t in (tc, tp, tl)]
        zi_, zl_ = zi.contiguous().float(), zl.contiguous().float()
        B = c.shape[0]
        dhw = c.numel() // B
        dev = c.device
        sums = torch.zeros(L.SP_REDUCE_ROWS, 16, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(8, dtype=torch.float32, device=dev)
        args = []
        for t, bs in recs + gts:
            args += [O.ptr(t), bs]
        L.call("sp_cae_loss_fwd", *args, B, dhw, O.ptr(zi_), O.ptr(zl_), zi_.numel(), float(weight), float(eps), float(factor),
               O.ptr(sums), O.ptr(loss), O.ptr(coef), O.stream())
        ctx.save_for_backward(*[t for t, _ in recs + gts], zi_, zl_, coef)
        ctx.strides = [bs for _, bs in recs + gts]
        ctx.shapes = (tuple(c.shape), tuple(zi.shape), tuple(zl.shape))
        return loss

    @staticmethod
    def backward(ctx, gloss):
        from stroke_prediction_amd.runtime import lib as L, ops as O
        *ts, zi_, zl_, coef = ctx.saved_tensors
        B = ts[0].shape[0]
        dhw = ts[0].numel() // B
        up = gloss if (gloss.dtype == torch.float32 and gloss.is_contiguous()) else gloss.float().contiguous()
        # slot k of the gradient buffer for the reconstruction that lies k-th in memory (equal spacing = slices of one stacked tensor)
        order = sorted(range(4), key=lambda k: ts[k].data_ptr())
        dall = torch.empty((4 * B,) + ctx.shapes[0][1:], dtype=torch.float32, device=ts[0].device)
        d = [None] * 4
        for slot, k in enumerate(order):
            d[k] = dall[slot * B:(slot + 1) * B]
        dzi, dzl = torch.empty_like(zi_), torch.empty_like(zl_)
        args = []
        for t, bs in zip(ts, ctx.strides):
            args += [O.ptr(t), bs]
        L.call("sp_cae_loss_bwd", *args, B, dhw, O.ptr(coef), O.ptr(up), O.ptr(d[0]), O.ptr(d[1]), O.ptr(d[2]), O.ptr(d[3]),
               O.ptr(zi_), O.ptr(zl_), zi_.numel(), O.ptr(dzi), O.ptr(dzl), O.stream())
        return d[0], d[1], d[2], d[3], None, None, None, dzi.view(ctx.shapes[1]), dzl.view(ctx.shapes[2]), None, None, None


def cae_reconstruction_loss(rec, gt, lat, factor, criterion):
    """The loss of CaeReconstructionLearner.loss_step on the fused kernels, or None when they do not apply (host tensors, the
    exact data-parallel mode -- its means and Dice sums are global --, several label classes): the caller composes it then."""
    import os
    from stroke_prediction_amd.runtime.layers import SYNC
    ts = (rec.core, rec.penu, rec.lesion, rec.interpolation, gt.core, gt.penu, gt.lesion)
    if os.environ.get("SP_CAE_FUSED_LOSS", "1") == "0" or SYNC["on"] or not isinstance(criterion, BatchDiceLoss) or len(criterion._label_weights) != 1 \
            or criterion._dim != 1 or any(t is None or not t.is_cuda or t.dim() != 5 or t.shape[1] != 1 or t.shape != ts[0].shape for t in ts) \
            or lat.interpolation is None or lat.lesion is None or lat.interpolation.shape != lat.lesion.shape:
        return None
    return _CaeLossFn.apply(rec.core, rec.penu, rec.lesion, rec.interpolation, gt.core.float(), gt.penu.float(), gt.lesion.float(),
                            lat.interpolation, lat.lesion, float(factor), float(criterion._label_weights[0]), float(criterion._epsilon))


def _stacked_base(parts):
    """The (B, n, ...) fp32 tensor whose consecutive channel slices are exactly ``parts`` (each (B, 1, ...)), or None."""
    base = getattr(parts[0], "_base", None)
    n = len(parts)
    if base is None or base.dtype != torch.float32 or not base.is_contiguous() or base.dim() != parts[0].dim() \
            or base.shape[1] != n or base.shape[0] != parts[0].shape[0] or tuple(base.shape[2:]) != tuple(parts[0].shape[2:]):
        return None
    per = base[0, 0].numel()
    for i, p in enumerate(parts):
        if getattr(p, "_base", None) is not base or p.shape[1] != 1 or p.dtype != torch.float32 \
                or p.data_ptr() != base.data_ptr() + 4 * i * per or _batch_strided(p)[1] != n * per:
            return None
    return base


def mean_of_channel_losses(criterion, outputs, targets):
    """mean_i criterion(outputs[i], targets[i]) -- the reference's ``(Dice(core) + Dice(penu)) / 2``
    (learner/UnetSegmentationLearner.py:21-28).  When the criterion is a single-label BatchDiceLoss and the pairs are the
    consecutive channel slices of one segmentation tensor and one label tensor (what Unet3D.forward / UnetInference
    produce), this is BatchDiceLoss over n channels with weights w/n: evaluated in one sums / finalize / backward
    launch on the base tensors, and the gradient lands on the segmentation directly (no slice-backward zero-fill,
    copy and add per channel).  Anything else: the literal sum of calls."""
    n = len(outputs)
    if isinstance(criterion, BatchDiceLoss) and len(criterion._label_weights) == 1 and criterion._dim == 1 and n > 1 \
            and outputs[0].is_cuda:
        ob, tb = _stacked_base(outputs), _stacked_base(targets)
        if ob is not None and tb is not None:
            w = float(criterion._label_weights[0]) / n
            return _DiceFn.apply(ob, tb, (w,) * n, float(criterion._epsilon))
    total = criterion(outputs[0], targets[0])
    for o, t in zip(outputs[1:], targets[1:]):
        total = total + criterion(o, t)
    return total / n


# ---------------------------------------------------------------------------------------------- evaluation measures

def _measures_from_counts(tp, fp, fn, tn):
    size = (tp + fp) + (tp + fn)
    return BinaryMeasuresDto(2.0 * tp / size if size > 0 else 0.0, numpy.inf, numpy.inf,
                             tp / (tp + fp) if tp + fp > 0 else 0.0,
                             tp / (tp + fn) if tp + fn > 0 else 0.0,
                             tn / (tn + fp) if tn + fp > 0 else 0.0)


def binary_measures_numpy(result, target, binary_threshold=0.5, distances=True):
    """``metrics.py:31-46`` of the reference for host arrays: uploaded once and measured on the device like
    ``binary_measures_torch`` (there is no CPU implementation in this package; MedPy, which the reference calls, is
    restated only as the test oracle ``oracle/measures.py``)."""
    if not torch.cuda.is_available():
        raise RuntimeError("binary_measures_numpy (stroke_prediction_amd) measures on the GPU: no CUDA device available")
    r = torch.from_numpy(numpy.ascontiguousarray(result, dtype=numpy.float32)).cuda()
    t = torch.from_numpy(numpy.ascontiguousarray(target, dtype=numpy.float32)).cuda()
    return binary_measures_torch(r, t, True, binary_threshold=binary_threshold, distances=distances)


DISTANCE_METRICS = True      # Hausdorff / ASSD as the reference's batch metrics report them (metrics.py:42-44)


_SD_WS = {}


def _surface_distances_launch(r, t, threshold):
    """Enqueue sp_surface_distances; returns the fp64[6] result tensor (no synchronisation)."""
    from stroke_prediction_amd.runtime import lib as L, ops as O
    dims = torch.tensor(list(r.shape), dtype=torch.int32)            # host array: read by the launcher, not the kernels
    key = (r.device, r.numel())
    if key not in _SD_WS:
        _SD_WS.clear()                                               # one workspace (4 volumes) alive at a time
        _SD_WS[key] = torch.empty(4 * r.numel(), dtype=torch.float32, device=r.device)
    out = torch.zeros(6, dtype=torch.float64, device=r.device)
    L.call("sp_surface_distances", O.ptr(r), O.ptr(t), float(threshold), r.dim(), dims.data_ptr(), O.ptr(_SD_WS[key]), O.ptr(out),
           O.stream())
    return out


def _surface_metrics_device(r, t, threshold):
    """(hd, assd) of two CUDA fp32 tensors of equal shape (rank <= 5), medpy semantics -- see sp_surface_distances."""
    mx_rt, sm_rt, n_r, mx_tr, sm_tr, n_t = _surface_distances_launch(r, t, threshold).tolist()
    return float(numpy.sqrt(max(mx_rt, mx_tr))), 0.5 * (sm_rt / n_r + sm_tr / n_t)


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
        if distances and r.dim() > 5:
            raise ValueError("surface distances: tensors of rank <= 5 (got %d)" % r.dim())
        sd = _surface_distances_launch(r, t, binary_threshold) if distances else None    # enqueued before the one sync below
        tp, fp, fn, tn = (float(v) for v in counts.tolist())
        out = _measures_from_counts(tp, fp, fn, tn)
        if distances and tp + fp > 0 and tp + fn > 0:
            mx_rt, sm_rt, n_r, mx_tr, sm_tr, n_t = sd.tolist()
            out.hd, out.assd = float(numpy.sqrt(max(mx_rt, mx_tr))), 0.5 * (sm_rt / n_r + sm_tr / n_t)
        return out
    # host tensors / arrays: same measures, on the device
    result = result.detach().cpu().numpy() if isinstance(result, torch.Tensor) else result
    target = target.detach().cpu().numpy() if isinstance(target, torch.Tensor) else target
    return binary_measures_numpy(result, target, binary_threshold=binary_threshold, distances=distances)
