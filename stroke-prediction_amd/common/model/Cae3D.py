"""Drop-in convolutional auto-encoder (reference ``common/model/Cae3D.py``): ``Enc3D``, ``Dec3D``, ``Cae3D``
with the reference constructors, ``forward(dto) -> dto``, ``freeze`` and ``state_dict`` keys
(``encoder.{0,1,3,4,...}``, ``decoder.{...}``; ``enc.`` / ``dec.`` prefixes under ``Cae3D``).

The ``torch.nn`` members are parameter containers; every encoder / decoder call is one
``torch.autograd.Function`` running the fused HIP units of ``runtime.cae_engine`` (BatchNorm applied on
the operand load, ELU / Sigmoid and the next layer's batch statistics fused into the conv epilogues,
strided and transposed convolutions through the same table-driven implicit-GEMM kernel).  The latent
interpolation (Cae3D.py:78-89) stays a three-operand torch expression on B x 800 x 1 x 10 x 10 values.
``Enc3DStep`` / ``Enc3DCtp`` (learned step, CTP-conditioned encoder) are outside the accelerated path.
"""
import contextlib
import os
import weakref

import torch
import torch.nn as nn

import common.dto.CaeDto as CaeDtoUtil
from common.dto.CaeDto import CaeDto
from stroke_prediction_amd.runtime import lib as _L
from stroke_prediction_amd.runtime.flat import FlatParamsMixin
from stroke_prediction_amd.runtime import ops as _O


def _containers(table, cm):
    mods = {}
    for i, (kind, ci, co, k, s, p) in enumerate(table):
        mods[str(3 * i)] = nn.BatchNorm3d(cm[ci])
        if kind == "conv":
            mods[str(3 * i + 1)] = nn.Conv3d(cm[ci], cm[co], k, stride=s, padding=p)
        else:
            mods[str(3 * i + 1)] = nn.ConvTranspose3d(cm[ci], cm[co], k, stride=s, padding=p, output_padding=0)
    return nn.ModuleDict(mods)


class _Lease:
    """A StackContext on loan from the pool to one autograd node: returned by backward, or -- when the graph is dropped
    without a backward (a grad-enabled forward whose loss is never differentiated) -- when the node is collected."""

    def __init__(self, pool, key, sc, module=None):
        self.pool, self.key, self.sc, self.module = pool, key, sc, module

    def release(self, dropped=False):
        if self.sc is not None:
            self.pool.release(self.key, self.sc)
            self.sc = None
            if dropped and self.module is not None:
                # the pass will never be differentiated: it no longer counts as "backward still to come", or the stack's
                # gradients would never be declared final again (and the in-backward gradient exchange stay off)
                self.module._n_out = max(0, getattr(self.module, "_n_out", 1) - 1)
        self.module = None

    def __del__(self):
        try:
            self.release(dropped=True)
        except Exception:
            pass


# The 3 encoder / 4 decoder passes of a step (Cae3D.py:105-107,230-233) are independent but for the BatchNorm running statistics
# and the shared parameter gradients.  SP_CAE_STREAMS (default 1): while a hipGraph is being captured every pass of a call runs
# on its own stream -- parallel branches of the graph: the step is ~980 launches, 765 of them under 20 us, which then overlap
# 3-4 deep -- with per-layer events keeping the running-statistics updates in pass order and per-pass gradient buffers added up
# on one stream.  2: in eager launches too.  0: one stream.
CAE_STREAMS = int(os.environ.get("SP_CAE_STREAMS", "1"))
_DBG = set(os.environ.get("SP_CAE_DBG", "").split(","))
_LANE_STREAMS = {}
_REDUCE_STREAMS = {}


def _concurrent_passes():
    return CAE_STREAMS == 2 or (CAE_STREAMS == 1 and torch.cuda.is_current_stream_capturing())


def _lane_stream(device, lane):
    key = (device.index, lane)
    if key not in _LANE_STREAMS:
        _LANE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _LANE_STREAMS[key]


def _reduce_stream(device):
    if device.index not in _REDUCE_STREAMS:
        _REDUCE_STREAMS[device.index] = torch.cuda.Stream(device=device)
    return _REDUCE_STREAMS[device.index]


class _StackFn(torch.autograd.Function):
    """One encoder or decoder call.  ``x`` needs a gradient only for the decoder (the latent).
    opts: None, or dict(lane, nlanes, order, bump_nbt) of a pass that runs concurrently with its siblings."""

    @staticmethod
    def forward(ctx, module, opts, x, *params):
        opts = opts or {}
        ctx.nparams = len(params)
        key, sc = opts.get("sc") or module._pool().acquire(x.shape[0], tuple(x.shape[2:]), module._dtype_code(), x.device, lane=opts.get("lane", 0))
        ctx.concurrent = bool(opts.get("concurrent"))
        with (_O.no_fork() if ctx.concurrent else contextlib.nullcontext()):
            out = sc.forward(x, module._param_dict(), module._buffer_dict(), module.training, bump_nbt=opts.get("bump_nbt", True),
                             order=opts.get("order"))
        ctx.home = opts.get("home")              # the stream the call came from (concurrent passes: joined again after backward)
        ctx.module, ctx.sc = module, sc
        ctx.frozen = module._stack_frozen()      # only the input needs a gradient: data-gradient-only backward
        ctx.lease = _Lease(module._pool(), key, sc, None if ctx.frozen else module)
        if not ctx.frozen:
            module._begin_step()
            module._n_out = getattr(module, "_n_out", 0) + 1       # passes of this stack whose backward is still to come
        ctx.training = module.training
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        module, sc = ctx.module, ctx.sc
        (out,) = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("%s.backward after an eval-mode forward is not supported: the fused backward uses the "
                               "batch-statistics BatchNorm formula; call model.train() for passes that need gradients"
                               % type(module).__name__)
        if ctx.frozen:
            dx = _frozen_backward(ctx, module, sc, dout, out)
            return (None, None, dx) + tuple(None for _ in range(ctx.nparams))
        names, views, inplace = module._grad_targets()
        red = None
        if ctx.concurrent and inplace and "nored" not in _DBG:
            # sibling passes run their backward on other streams at the same time: accumulate into this context's own buffer,
            # then add it to the stack's segment of the flat gradient buffer on ONE stream (adds of all passes in issue order)
            priv, grads = sc.private_grads(names, views)
            with _O.no_fork():
                dx = sc.backward(dout, out, module._param_dict(), grads, ctx.need_dx)
            seg = module._flat_segment()
            cur = torch.cuda.current_stream()
            red = _reduce_stream(dout.device)
            red.wait_stream(cur)
            with torch.cuda.stream(red):
                seg.add_(priv)              # (the segment's offset is not 16-byte aligned: a plain torch add, ~9 MB)
        else:
            with (_O.no_fork() if ctx.concurrent else contextlib.nullcontext()):
                dx = sc.backward(dout, out, module._param_dict(), dict(zip(names, views)), ctx.need_dx)
        ctx.lease.release()
        module._n_out = max(0, getattr(module, "_n_out", 1) - 1)
        if module._n_out == 0:
            if red is not None:
                torch.cuda.current_stream().wait_stream(red)      # every pass of this stack has queued its add by now
            module._stack_grads_final()
        if ctx.concurrent and ctx.home is not None and ctx.home != torch.cuda.current_stream() and "nojoin" not in _DBG:
            # autograd joins a node's stream with its consumers' and with the streams of AccumulateGrad nodes; a pass that
            # returns no gradient tensor (parameters accumulate in place, the encoder input needs none) would otherwise stay
            # un-joined: the optimiser on the home stream would not wait for it, and a stream capture could not end
            ctx.home.wait_stream(torch.cuda.current_stream())
        return (None, None, dx) + tuple(None if inplace else v for v in views)


def _frozen_backward(ctx, module, sc, dout, out):
    """backward through a stack whose parameters are frozen (the phase-2 learners: CaePredictionLearner.py:27,
    train_interpolationstep_after_reconstruction.py:22): the gradient of the stack INPUT only -- data-gradient convolutions
    and the BatchNorm-backward terms of the live batch statistics; no weight-gradient kernel runs, nothing is written to
    the module's gradient buffers"""
    names = [n for n, _ in module.named_parameters()]
    shapes = [p for _, p in module.named_parameters()]
    _, scratch = sc.private_grads(names, shapes)
    with (_O.no_fork() if getattr(ctx, "concurrent", False) else contextlib.nullcontext()):
        dx = sc.backward(dout, out, module._param_dict(), scratch, True, param_grads=False)
    ctx.lease.release()
    return dx


# SP_CAE_BATCHED (default 1): the passes of one encoder / decoder call are stacked along the batch axis and run as ONE pass
# through a grouped StackContext -- one launch per layer for the convolution, the weight gradient and the data gradient of all
# passes, per-pass BatchNorm statistics, running statistics updated in pass order (runtime/layers.py, ConvLayer(groups=G)).
# 0: every pass on its own (sequentially, or as parallel graph branches: SP_CAE_STREAMS).
CAE_BATCHED = int(os.environ.get("SP_CAE_BATCHED", "1"))


class _StackManyFn(torch.autograd.Function):
    """All passes of one encoder / decoder call as one autograd node: forward(module, n, x_0 .. x_{n-1}, *params) -> n outputs."""

    @staticmethod
    def forward(ctx, module, n, *args):
        xs, B = args[:n], args[0].shape[0]
        key, sc = module._pool().acquire(n * B, tuple(xs[0].shape[2:]), module._dtype_code(), xs[0].device, groups=n)
        x = torch.cat([t.float() for t in xs], 0)
        out = sc.forward(x, module._param_dict(), module._buffer_dict(), module.training)
        ctx.module, ctx.sc, ctx.n, ctx.B = module, sc, n, B
        ctx.nparams = len(args) - n
        ctx.frozen = module._stack_frozen()
        ctx.lease = _Lease(module._pool(), key, sc, None if ctx.frozen else module)
        if not ctx.frozen:
            module._begin_step()
            module._n_out = getattr(module, "_n_out", 0) + 1
        ctx.training = module.training
        ctx.need_dx = any(t.requires_grad for t in xs)
        ctx.save_for_backward(out)
        outs = tuple(out[i * B:(i + 1) * B] for i in range(n))
        return outs

    @staticmethod
    def backward(ctx, *douts):
        module, sc, n, B = ctx.module, ctx.sc, ctx.n, ctx.B
        (out,) = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("%s.backward after an eval-mode forward is not supported: the fused backward uses the "
                               "batch-statistics BatchNorm formula; call model.train() for passes that need gradients"
                               % type(module).__name__)
        base = getattr(douts[0], "_base", None) if douts[0] is not None else None
        if base is not None and base.is_contiguous() and tuple(base.shape) == tuple(out.shape) and base.dtype == out.dtype and \
                all(d is not None and getattr(d, "_base", None) is base and d.is_contiguous() and tuple(d.shape) == tuple(out[:B].shape)
                    and d.data_ptr() == base.data_ptr() + k * d.numel() * d.element_size() for k, d in enumerate(douts)):
            dout = base             # the gradients arrive stacked already (metrics._CaeLossFn): no concatenation
        else:
            dout = torch.cat([torch.zeros_like(out[:B]) if d is None else d for d in douts], 0)
        if ctx.frozen:
            dx = _frozen_backward(ctx, module, sc, dout, out)
            dxs = tuple(None for _ in range(n)) if dx is None else tuple(dx[i * B:(i + 1) * B] for i in range(n))
            return (None, None) + dxs + tuple(None for _ in range(ctx.nparams))
        names, views, inplace = module._grad_targets()
        dx = sc.backward(dout, out, module._param_dict(), dict(zip(names, views)), ctx.need_dx)
        ctx.lease.release()
        module._n_out = max(0, getattr(module, "_n_out", 1) - 1)
        if module._n_out == 0:
            module._stack_grads_final()
        dxs = tuple(None for _ in range(n)) if dx is None else tuple(dx[i * B:(i + 1) * B] for i in range(n))
        return (None, None) + dxs + tuple(None if inplace else v for v in views)


class CaeBase(FlatParamsMixin, nn.Module):
    FLAT_NBT = True      # every BatchNorm of a stack runs once per stack call: their step counters advance together

    def __init__(self, size_input_xy=128, size_input_z=28, channels=[1, 16, 32, 64, 128, 1024, 128, 1], n_ch_global=2,
                 alpha=0.01, inner_xy=12, inner_z=3, dtype="bf16"):
        super().__init__()
        assert size_input_xy % 4 == 0 and size_input_z % 4 == 0
        self.channels = list(channels)
        self.n_ch_origin = channels[1]
        self.n_ch_down2x = channels[2]
        self.n_ch_down4x = channels[3]
        self.n_ch_down8x = channels[4]
        self.n_ch_fc = channels[5]
        self._inner_ch = self.n_ch_down8x
        self._inner_xy = inner_xy
        self._inner_z = inner_z
        self.n_ch_global = n_ch_global
        self.n_input = channels[0]
        self.n_classes = channels[-1]
        self.alpha = alpha
        self.compute_dtype = dtype
        self._stack_pool = None

    # ------------------------------------------------------------------ HIP plumbing
    _TABLE, _PREFIX, _LAST_SIGMOID = None, None, False

    def _dtype_code(self):
        return _L.SP_BF16 if self.compute_dtype == "bf16" else _L.SP_F32

    def _pool(self):
        from stroke_prediction_amd.runtime.cae_engine import StackPool
        if self._stack_pool is None:
            self._stack_pool = StackPool(self._TABLE, self._PREFIX, self.channels, self.alpha, self._LAST_SIGMOID)
        return self._stack_pool

    def _run_stack(self, x, opts=None):
        if x is None:
            return None
        if not x.is_cuda or not next(self.parameters()).is_cuda:
            raise RuntimeError("%s (stroke_prediction_amd) runs on the MI355X HIP path only: move the model and its "
                               "inputs to the GPU; there is no CPU fallback" % type(self).__name__)
        self._ensure_flat()
        if x.dtype != torch.float32:
            x = x.float()
        params = [p for _, p in self.named_parameters()]
        if torch.is_grad_enabled() and (x.requires_grad or not self._stack_frozen()):
            return _StackFn.apply(self, opts, x, *params)
        opts = opts or {}
        key, sc = opts.get("sc") or self._pool().acquire(x.shape[0], tuple(x.shape[2:]), self._dtype_code(), x.device, lane=opts.get("lane", 0))
        out = sc.forward(x, self._param_dict(), self._buffer_dict(), self.training, bump_nbt=opts.get("bump_nbt", True),
                         order=opts.get("order"))
        self._pool().release(key, sc)
        return out

    def _run_stack_many(self, xs):
        """The passes of one encoder / decoder call (``None`` entries stay ``None``): one after the other on the current
        stream, or -- see CAE_STREAMS -- each on its own stream with the outputs joined back before returning."""
        idx = [i for i, x in enumerate(xs) if x is not None]
        if CAE_BATCHED and len(idx) > 1 and xs[idx[0]].is_cuda and next(self.parameters()).is_cuda and \
                all(xs[i].shape == xs[idx[0]].shape and xs[i].device == xs[idx[0]].device for i in idx):
            return self._run_stack_batched(xs, idx)
        if len(idx) <= 1 or not xs[idx[0]].is_cuda or CAE_STREAMS == 0:
            return [self._run_stack(x) for x in xs]
        # Pass k always runs in the contexts of lane k (their own packed weights and workspaces), whether or not the lanes run
        # concurrently right now: the eager warm-up steps of Learner(graph=True) thereby create, on one stream, exactly the
        # contexts the captured step uses (building one uploads tables, which a capturing stream may not do).
        conc = _concurrent_passes()
        self._ensure_flat()
        dev = xs[idx[0]].device
        main = torch.cuda.current_stream(dev)
        n = len(idx)
        training = self.training
        nbt = self._buffer_dict().get("__nbt_flat__") if training else None
        if nbt is not None:
            nbt.add_(n)                           # one increment for the n passes (each BatchNorm runs once per pass)
        nlayers = len(self._TABLE)
        outs = list(xs)
        prev = None
        x0 = xs[idx[0]]
        scs = [self._pool().acquire(x0.shape[0], tuple(x0.shape[2:]), self._dtype_code(), dev, lane=lane) for lane in range(n)]
        fork_ev = None
        if conc:
            # weights that depend on the parameters only are packed once, here, before the lanes fork (shared bank)
            with_bwd = torch.is_grad_enabled() and not self._stack_frozen()
            scs[0][1].prepare(self._param_dict(), with_bwd)
            fork_ev = torch.cuda.Event()
            fork_ev.record(main)          # every lane starts HERE (not behind lane 0's work, which goes to the home stream)
        for lane, i in enumerate(idx):
            s = main if (lane == 0 or not conc) else _lane_stream(dev, lane)
            if s is not main:
                s.wait_event(fork_ev)
            rec = [torch.cuda.Event() for _ in range(nlayers)] if (conc and training and lane + 1 < n and "noorder" not in _DBG) else None
            with torch.cuda.stream(s):
                outs[i] = self._run_stack(xs[i], dict(lane=lane, sc=scs[lane], concurrent=conc, home=main, bump_nbt=nbt is None,
                                                      order=(prev, rec) if (conc and training) else None))
            prev = rec
        if conc:
            for lane, i in enumerate(idx):
                if lane:
                    main.wait_stream(_lane_stream(dev, lane))
                    if "norecord" not in _DBG:
                        outs[i].record_stream(main)
        return outs

    def _run_stack_batched(self, xs, idx):
        """all passes of the call in ONE grouped context (CAE_BATCHED)"""
        self._ensure_flat()
        n = len(idx)
        ins = [xs[i] for i in idx]
        params = [p for _, p in self.named_parameters()]
        outs = list(xs)
        if torch.is_grad_enabled() and (any(t.requires_grad for t in ins) or not self._stack_frozen()):
            res = _StackManyFn.apply(self, n, *ins, *params)
        else:
            B = ins[0].shape[0]
            key, sc = self._pool().acquire(n * B, tuple(ins[0].shape[2:]), self._dtype_code(), ins[0].device, groups=n)
            out = sc.forward(torch.cat([t.float() for t in ins], 0), self._param_dict(), self._buffer_dict(), self.training)
            self._pool().release(key, sc)
            res = tuple(out[i * B:(i + 1) * B] for i in range(n))
        for i, r in zip(idx, res):
            outs[i] = r
        return outs

    def _stack_frozen(self):
        """no parameter of the convolution stack itself wants a gradient (``freeze(True)``; the 1x1x1 step layers of an
        ``Enc3DStep`` are torch modules outside the stack and do not count)"""
        pre = self._PREFIX + "."
        return not any(p.requires_grad for n, p in self.named_parameters() if n.startswith(pre))

    def _flat_segment(self):
        """this stack's slice of the (root's) flat gradient buffer"""
        root = self._flat_root()
        n = sum(p.numel() for p in self.parameters())
        if root is self:
            return root._flat_grad[:n]
        for name, m in root.named_children():
            if m is self:
                lo = root._flat_offset_of(name + ".")
                return root._flat_grad[lo:lo + n]
        raise RuntimeError("stack is not a child of its flat root")

    def _stack_grads_final(self):
        """every pass of this stack recorded in the current step has run its backward: its parameter gradients are
        final.  Data-parallel training starts the all-reduce of that part of the flat buffer now (the decoder's four
        backward passes come first, so its bucket travels while the encoder's three run), the rest at the latest in the
        optimiser's step pre-hook (parallel.DataParallelSync)."""
        root = self._flat_root()
        if root is self:
            self._after_backward()
            return
        stacks = [m for m in root.children() if isinstance(m, CaeBase)]
        if all(getattr(m, "_n_out", 0) == 0 for m in stacks):
            root._after_backward()
        else:
            for name, m in root.named_children():
                if m is self and root._flat_grad is not None:
                    # only the segment at the END of what is still pending may go (the flat buffer is exchanged back to front)
                    lo = root._flat_offset_of(name + ".")
                    hi = root._flat_grad.numel() if root._bucket_hi is None else root._bucket_hi
                    if lo + sum(p.numel() for p in self.parameters()) == hi:
                        self._grads_ready_from(name + ".")

    def _grad_targets(self):
        # an encoder / decoder runs 3-4 times per step: never hand the shared flat views to autograd twice
        names, views, inplace = super()._grad_targets()
        if not inplace and self._flat_root() is self:
            self._flat_grad.zero_()
            views = [torch.zeros_like(v) for v in views]
        return names, views, inplace

    def freeze(self, freeze=False):
        requires_grad = not freeze
        for param in self.parameters():
            param.requires_grad = requires_grad

    def __setstate__(self, state):
        """see ``Unet3D.__setstate__``: a ``.model`` file written by the reference's ``Enc3D`` / ``Dec3D`` carries the
        reference's attributes (``n_ch_*``, ``n_input``, ``n_classes``, ``alpha``, CaeBase Cae3D.py:13-26); the channel list
        and the precision mode of this implementation are rebuilt from them."""
        super().__setstate__(state)
        d = self.__dict__
        if "channels" not in d:
            d["channels"] = [d["n_input"], d["n_ch_origin"], d["n_ch_down2x"], d["n_ch_down4x"], d["n_ch_down8x"], d["n_ch_fc"],
                             d["n_classes"]]
        d.setdefault("compute_dtype", "bf16")
        d["_stack_pool"] = None
        d.setdefault("_flat_parent", None)

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_stack_pool"] = None
        state["_flat_parent"] = None
        for k in ("_flat_param", "_flat_grad", "_flat_views", "_flat_pviews", "_flat_names", "_flat_device", "_flat_nbt",
                  "grad_sync", "grad_bucket_ready", "_bucket_hi", "_grads_synced", "_n_out"):
            state.pop(k, None)
        return state


class Enc3D(CaeBase):
    from stroke_prediction_amd.runtime.cae_engine import ENC_LAYERS as _TABLE
    _PREFIX = "encoder"

    def __init__(self, size_input_xy, size_input_z, channels, n_ch_global, alpha, dtype="bf16"):
        super().__init__(size_input_xy, size_input_z, channels, n_ch_global, alpha, inner_xy=10, inner_z=3, dtype=dtype)
        from stroke_prediction_amd.runtime.cae_engine import channel_map
        self.encoder = _containers(self._TABLE, channel_map(channels))

    def _interpolate(self, latent_core, latent_penu, step):
        """core + step * (penu - core), per sample (reference Cae3D.py:78-89)."""
        assert step is not None, 'Step must be given for interpolation!'
        if latent_core is None or latent_penu is None:
            return None
        return latent_core + step * (latent_penu - latent_core)

    def _forward_single(self, input_image):
        return self._run_stack(input_image)

    def _get_step(self, dto: CaeDto):
        return dto.given_variables.time_to_treatment

    def forward(self, dto: CaeDto):
        step = self._get_step(dto)
        if dto.flag == CaeDtoUtil.FLAG_GTRUTH or dto.flag == CaeDtoUtil.FLAG_DEFAULT:
            assert dto.latents.gtruth._is_empty()   # do not overwrite earlier results by mistake
            gt, lat = dto.given_variables.gtruth, dto.latents.gtruth
            lat.core, lat.penu, lat.lesion = self._run_stack_many([gt.core, gt.penu, gt.lesion])
            lat.interpolation = self._interpolate(lat.core, lat.penu, step)
        if dto.flag == CaeDtoUtil.FLAG_INPUTS or dto.flag == CaeDtoUtil.FLAG_DEFAULT:
            assert dto.latents.inputs._is_empty()
            inp, lat = dto.given_variables.inputs, dto.latents.inputs
            lat.core, lat.penu = self._run_stack_many([inp.core, inp.penu])
            lat.interpolation = self._interpolate(lat.core, lat.penu, step)
        return dto


class Enc3DStep(Enc3D):
    """Encoder that can LEARN the interpolation step from the clinical globals when no time to treatment is given
    (reference Cae3D.py:121-142): two 1x1x1 convolutions + ELU reduce the B x n_global x 1 x 1 x 1 vector, a third one
    and a sigmoid yield ``step``.  The three convolutions see B x 5 values: they stay plain ``torch.nn`` modules on the
    device (their gradient reaches them through the torch-side latent interpolation); the encoder stack itself is the
    fused HIP path of ``Enc3D``."""

    def __init__(self, size_input_xy, size_input_z, channels, n_ch_global, alpha, dtype="bf16"):
        super().__init__(size_input_xy, size_input_z, channels, n_ch_global, alpha, dtype=dtype)
        g = self.n_ch_global
        self.reduce = nn.Sequential(nn.Conv3d(g, g, 1), nn.ELU(self.alpha, True), nn.Conv3d(g, g // 2, 1), nn.ELU(self.alpha, True))
        self.step = nn.Conv3d(g // 2, 1, 1)
        nn.init.normal_(self.step.weight, 0, 0.001)      # the reference stresses this initialisation (Cae3D.py:133-134)
        nn.init.normal_(self.step.bias, 0.5, 0.01)
        self.sigmoid = nn.Sigmoid()

    def _get_step(self, dto: CaeDto):
        step = dto.given_variables.time_to_treatment
        if step is None:
            step = self.sigmoid(self.step(self.reduce(dto.given_variables.globals)))
        return step


class Enc3DCtp(Enc3D):
    """CTP-conditioned encoder (reference Cae3D.py:145-169): only reachable from ``train_shape_reconstruction_with_ctp.py``,
    which passes keyword arguments the reference classes do not accept -- outside the accelerated path."""

    def __init__(self, *a, **k):
        raise NotImplementedError("Enc3DCtp is outside the MI355X path (SURVEY.md 2.1 row 2: not used by the two named scripts)")


class Dec3D(CaeBase):
    from stroke_prediction_amd.runtime.cae_engine import DEC_LAYERS as _TABLE
    _PREFIX = "decoder"
    _LAST_SIGMOID = True

    def __init__(self, size_input_xy, size_input_z, channels, n_ch_global, alpha, dtype="bf16"):
        super().__init__(size_input_xy, size_input_z, channels, n_ch_global, alpha, inner_xy=10, inner_z=3, dtype=dtype)
        from stroke_prediction_amd.runtime.cae_engine import channel_map
        self.decoder = _containers(self._TABLE, channel_map(channels))

    def _forward_single(self, input_latent):
        return self._run_stack(input_latent)

    def forward(self, dto: CaeDto):
        if dto.flag == CaeDtoUtil.FLAG_GTRUTH or dto.flag == CaeDtoUtil.FLAG_DEFAULT:
            assert dto.reconstructions.gtruth._is_empty()
            lat, rec = dto.latents.gtruth, dto.reconstructions.gtruth
            rec.core, rec.penu, rec.lesion, rec.interpolation = self._run_stack_many([lat.core, lat.penu, lat.lesion, lat.interpolation])
        if dto.flag == CaeDtoUtil.FLAG_INPUTS or dto.flag == CaeDtoUtil.FLAG_DEFAULT:
            assert dto.reconstructions.inputs._is_empty()
            lat, rec = dto.latents.inputs, dto.reconstructions.inputs
            rec.core, rec.penu, rec.interpolation = self._run_stack_many([lat.core, lat.penu, lat.interpolation])
        return dto


class Cae3D(FlatParamsMixin, nn.Module):
    """enc -> dec on the same DTO (reference Cae3D.py:242-255); owns the flat parameter / gradient buffers
    of both halves so one fused Adam launch and one all-reduce cover the whole model."""

    def __init__(self, enc: Enc3D, dec: Dec3D):
        super().__init__()
        self.enc = enc
        self.dec = dec
        self._adopt()

    def _adopt(self):
        ref = weakref.ref(self)
        self.enc._flat_parent = ref
        self.dec._flat_parent = ref

    def forward(self, dto: CaeDto):
        self._adopt()
        dto = self.enc(dto)
        dto = self.dec(dto)
        return dto

    def freeze(self, freeze: bool):
        self.enc.freeze(freeze)
        self.dec.freeze(freeze)

    def __setstate__(self, state):
        super().__setstate__(state)
        self._adopt()

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_flat_param", "_flat_grad", "_flat_views", "_flat_pviews", "_flat_names", "_flat_device", "_flat_nbt",
                  "grad_sync", "grad_bucket_ready", "_bucket_hi", "_grads_synced", "_n_out"):
            state.pop(k, None)
        return state
