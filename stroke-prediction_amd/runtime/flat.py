"""Flat parameter / gradient storage for the drop-in models.

All parameters of a model live in ONE contiguous fp32 buffer and their gradients in another
(``Parameter.data`` / ``.grad`` are views).  That buffer pair is the operand of the fused Adam
kernel (``sp_adam_step_flat``) and of the single RCCL all-reduce in data-parallel training
(1.42 MB for the default U-Net, 18.9 MB for the CAE: SURVEY.md 5).
"""
import torch


def _bump_epoch():
    from . import ops
    ops.bump_param_epoch()


class FlatParamsMixin:
    _flat_param = None
    _flat_grad = None
    _flat_views = None
    _flat_names = None
    _flat_device = None
    grad_sync = None            # optional callable(flat_grad) -> None, installed by parallel.DataParallelSync
    _flat_parent = None         # set on sub-modules whose parameters live in a parent's flat buffer (Cae3D)
    _flat_nbt = None

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flat_param = None       # .cuda()/.cpu()/.float() re-created the storages: re-flatten lazily
        _bump_epoch()                 # packed weight fragments keyed on the old storages are stale
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        _bump_epoch()                 # (copy_ under no_grad bumps the version counters too; belt and braces)
        return out

    def _flat_root(self):
        return self if self._flat_parent is None else self._flat_parent()._flat_root()

    def _ensure_flat(self):
        if self._flat_parent is not None:
            return self._flat_root()._ensure_flat()
        named = list(self.named_parameters())
        dev = named[0][1].device
        if self._flat_param is not None and self._flat_device == dev and \
                all(p.data_ptr() == v.data_ptr() for (_, p), v in zip(named, self._flat_pviews)):
            return
        total = sum(p.numel() for _, p in named)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        off, pviews, gviews = 0, [], []
        for _, p in named:
            n = p.numel()
            pv = flat[off:off + n].view(p.shape)
            pv.copy_(p.data)
            had_grad = p.grad is not None
            p.data = pv
            gv = gflat[off:off + n].view(p.shape)
            if had_grad:
                gv.copy_(p.grad)
                p.grad = gv
            pviews.append(pv)
            gviews.append(gv)
            off += n
        self._flat_param, self._flat_grad = flat, gflat
        self._flat_pviews, self._flat_views = pviews, gviews
        self._flat_names = [n for n, _ in named]
        self._flat_device = dev
        # BatchNorm step counters: one int64 vector per module that opts in with FLAT_NBT (every BatchNorm under it runs
        # exactly once per forward call of that module); the BatchNorms' buffers are its elements -> one increment per
        # call instead of one tiny kernel per BatchNorm
        for mod in self.modules():
            if getattr(mod, "FLAT_NBT", False):
                bns = [m for m in mod.modules() if m._buffers.get("num_batches_tracked") is not None]
                if bns:
                    nbt = torch.zeros(len(bns), dtype=torch.int64, device=dev)
                    for i, m in enumerate(bns):
                        nbt[i] = m._buffers["num_batches_tracked"].to(dev)
                        m._buffers["num_batches_tracked"] = nbt[i]
                    mod._flat_nbt = nbt

    def _param_dict(self):
        """name -> tensor sharing the Parameter's storage AND its version counter (``p.detach()``; ``p.data`` always
        reports ``_version == 0``): the conv runners key their packed-weight caches on (data_ptr, _version), so
        ``load_state_dict``, ``torch.optim`` steps and in-place edits of a parameter all invalidate them."""
        self._ensure_flat()
        return {n: p.detach() for n, p in self.named_parameters()}

    def _buffer_dict(self):
        self._ensure_flat()
        d = dict(self.named_buffers())
        if getattr(self, "FLAT_NBT", False) and getattr(self, "_flat_nbt", None) is not None:
            d["__nbt_flat__"] = self._flat_nbt
        return d

    def _grad_targets(self):
        """(names, gradient views, inplace).  inplace=True: every ``p.grad`` already IS its view of the flat
        buffer -> kernels accumulate there and autograd gets ``None``; otherwise the flat buffer is zeroed,
        filled, and its views are handed to autograd (which adopts them as ``p.grad``)."""
        self._ensure_flat()
        root = self._flat_root()
        if root is not self:
            # a sub-module of a flat parent: its own parameters' slices of the parent's gradient buffer
            vmap = {id(p): v for (_, p), v in zip(root.named_parameters(), root._flat_views)}
            named = list(self.named_parameters())
            views = [vmap[id(p)] for _, p in named]
            inplace = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for (_, p), v in zip(named, views))
            if not inplace:      # cannot zero the shared buffer here (other sub-modules accumulate into it)
                views = [torch.zeros_like(v) for v in views]
            return [n for n, _ in named], views, inplace
        params = [p for _, p in self.named_parameters()]
        inplace = all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape
                      for p, v in zip(params, self._flat_views))
        if not inplace:
            self._flat_grad.zero_()
        return self._flat_names, self._flat_views, inplace

    # ---- data-parallel gradient exchange, bucketed in reverse layer order (parallel.DataParallelSync) --------------
    # Backward fills the flat gradient buffer from its END (the last layers' parameters) towards its start.  The engine
    # reports "everything from the first parameter named <prefix>* to the end of what is still pending is final"
    # (_grads_ready_from); the installed ``grad_bucket_ready(flat_grad, lo, hi)`` starts an asynchronous all-reduce of
    # that slice on the communication stream while the remaining data / weight gradients are computed, and
    # ``grad_sync(flat_grad, lo, hi)`` (end of backward) reduces the rest and waits for the buckets.
    grad_bucket_ready = None
    _bucket_hi = None
    _grads_synced = False

    def _flat_offset_of(self, prefix):
        off = 0
        for n, p in self.named_parameters():
            if n.startswith(prefix):
                return off
            off += p.numel()
        raise KeyError(prefix)

    def _grads_ready_from(self, prefix):
        root = self._flat_root()
        if root.grad_bucket_ready is None or root.grad_sync is None:
            return
        total = root._flat_grad.numel()
        hi = total if root._bucket_hi is None else root._bucket_hi
        lo = root._flat_offset_of(prefix)
        if lo < hi:
            root.grad_bucket_ready(root._flat_grad, lo, hi)
            root._bucket_hi = lo

    def _after_backward(self):
        root = self._flat_root()
        if root.grad_sync is not None and not root._grads_synced:
            hi = root._flat_grad.numel() if root._bucket_hi is None else root._bucket_hi
            root.grad_sync(root._flat_grad, 0, hi)
            root._bucket_hi = None
            root._grads_synced = True

    def _begin_step(self):
        """a grad-enabled forward starts a new exchange round"""
        root = self._flat_root()
        root._grads_synced = False
        root._bucket_hi = None

    def flat_buffers(self):
        self._ensure_flat()
        root = self._flat_root()
        return root._flat_param, root._flat_grad
