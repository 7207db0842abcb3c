"""Flat parameter / gradient storage for the drop-in models.

All parameters of a model live in ONE contiguous fp32 buffer and their gradients in another
(``Parameter.data`` / ``.grad`` are views).  That buffer pair is the operand of the fused Adam
kernel (``sp_adam_step_flat``) and of the single RCCL all-reduce in data-parallel training
(1.42 MB for the default U-Net, 18.9 MB for the CAE: SURVEY.md 5).
"""
import torch


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
        self._ensure_flat()
        return {n: p.data for n, p in self.named_parameters()}

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

    def _after_backward(self):
        if self.grad_sync is not None:
            self.grad_sync(self._flat_grad)

    def flat_buffers(self):
        self._ensure_flat()
        root = self._flat_root()
        return root._flat_param, root._flat_grad
