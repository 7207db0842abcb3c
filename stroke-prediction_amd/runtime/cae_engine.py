"""Encoder / decoder stacks of the convolutional auto-encoder (``Enc3D.encoder`` Cae3D.py:39-76,
``Dec3D.decoder`` Cae3D.py:176-220) on the fused HIP units of ``layers.ConvLayer``.

Every encoder/decoder CALL of a training step (three encoder passes, four decoder passes,
Cae3D.py:105-107,230-233) owns a ``StackContext``: its activations stay alive until that call's
backward has run, and its BatchNorm batch statistics are its own -- exactly as in the reference, where
each ``nn.Sequential`` call normalises with the statistics of the tensor it is given.
"""
import torch

from . import lib as L
from . import ops as O
from .layers import ConvLayer, Scratch, STATS_NREP, SYNC, _allreduce
from . import plan as P

import os
PITCH16 = not os.environ.get("SP_CAE_PITCH8")     # bf16: 16-channel pitches everywhere (24 -> 32 ...) so that the DMA kernels apply
FUSED_OUT = bool(int(os.environ.get("SP_CAE_FUSED_OUT", "1")))      # the decoder's BatchNorm -> Conv3d(16, 1, 1) -> Sigmoid tail on csrc/sp_pwout.hip

# (kind, cin_key, cout_key, kernel, stride, padding) -- the layer tables of the reference modules
ENC_LAYERS = [
    ("conv", "in", "o", 3, 1, (1, 0, 0)), ("conv", "o", "o", 3, 1, (1, 0, 0)),
    ("conv", "o", "d2", 3, 2, (1, 1, 1)),
    ("conv", "d2", "d2", 3, 1, (1, 0, 0)), ("conv", "d2", "d2", 3, 1, (1, 0, 0)),
    ("conv", "d2", "d4", 3, 2, (1, 1, 1)),
    ("conv", "d4", "d4", 3, 1, (1, 0, 0)), ("conv", "d4", "d4", 3, 1, (1, 0, 0)),
    ("conv", "d4", "d8", 3, 2, (0, 0, 0)),
    ("conv", "d8", "fc", 3, 1, (0, 0, 0)),
]
DEC_LAYERS = [
    ("convT", "fc", "d8", 3, 1, (0, 0, 0)), ("convT", "d8", "d4", 3, 2, (0, 0, 0)),
    ("conv", "d4", "d4", 3, 1, (1, 2, 2)), ("conv", "d4", "d2", 3, 1, (1, 2, 2)),
    ("convT", "d2", "d2", 2, 2, (0, 0, 0)),
    ("conv", "d2", "d2", 3, 1, (1, 2, 2)), ("conv", "d2", "o", 3, 1, (1, 2, 2)),
    ("convT", "o", "o", 2, 2, (0, 0, 0)),
    ("conv", "o", "o", 3, 1, (1, 2, 2)), ("conv", "o", "o", 3, 1, (1, 2, 2)),
    ("conv", "o", "o", 1, 1, (0, 0, 0)), ("conv", "o", "cls", 1, 1, (0, 0, 0)),
]


def channel_map(channels):
    """``CaeBase.__init__`` Cae3D.py:14-26."""
    return {"in": channels[0], "o": channels[1], "d2": channels[2], "d4": channels[3], "d8": channels[4],
            "fc": channels[5], "cls": channels[-1]}


class StackContext:
    """One call of an encoder / decoder stack: layers (with their activations), scratch, I/O staging."""

    def __init__(self, table, prefix, channels, alpha, batch, in_dims, dtype, device, last_sigmoid, bank=None, groups=1):
        """groups > 1: ``batch`` = groups x (samples of one pass): the passes of one encoder / decoder call stacked along the
        batch axis -- ONE launch per layer for the convolution, its weight gradient and its data gradient, per-pass
        BatchNorm statistics (layers.ConvLayer(groups=...))."""
        O.require_gpu()
        L.load()
        cm = channel_map(channels)
        self.batch, self.dtype, self.device = batch, dtype, device
        self.G = groups
        self.gb = batch // groups
        self.scratch = sc = Scratch(device)
        self.layers = []
        dims = tuple(in_dims)
        n = len(table)
        for i, (kind, ci, co, k, s, p) in enumerate(table):
            last = last_sigmoid and i == n - 1
            lay = ConvLayer("%s.%d" % (prefix, 3 * i + 1), kind, cm[ci], cm[co], k, s, p, dims, batch, dtype, device, sc,
                            bn_prefix="%s.%d" % (prefix, 3 * i), conv_prefix="%s.%d" % (prefix, 3 * i + 1),
                            act=L.ACT_SIGMOID if last else L.ACT_ELU, act_param=0.0 if last else alpha,
                            out_dtype=L.SP_F32 if last else None, need_input_grad=True, bank=bank,
                            pitch=16 if (dtype == L.SP_BF16 and PITCH16) else 8, groups=groups)
            self.layers.append(lay)
            dims = lay.out_dims
        self.in_dims, self.out_dims = tuple(in_dims), dims
        self.cin, self.cout = cm[table[0][1]], cm[table[-1][2]]
        for lay in self.layers:
            lay.reserve_bwd_scratch()
        sc.finalize()
        self.x0 = O.alloc_cl(batch, in_dims, self.layers[0].cpi, dtype, device)
        self.out_dtype = self.layers[-1].out_dtype
        # the output layer BatchNorm -> Conv3d(<= 16, 1, 1x1x1) -> Sigmoid (Cae3D.py:214-218) as three streaming kernels
        # (csrc/sp_pwout.hip): no normalised copy of its input, no 16-channel padded output / output gradient
        kind, _, _, k, s, p = table[-1]
        last = self.layers[-1]
        self.fused_out = bool(FUSED_OUT and last_sigmoid and kind == "conv" and k == 1 and s == 1 and max(P._triple(p)) == 0 and self.cout == 1
                              and dtype == L.SP_BF16 and last.cpi == 16 and n >= 2)
        self._out_sums = self._out_g = None

    def forward(self, x, params, bufs, training, bump_nbt=True, order=None):
        """x: (B, cin, D, H, W) fp32 on the device -> (B, cout, D', H', W') fp32.
        order = (wait, record): per-layer event lists of concurrently running passes of one stack (Cae3D._run_stack_many) --
        layer i starts after the previous pass recorded wait[i], and records record[i] when its own work is enqueued, so the
        BatchNorm running statistics are updated in pass order, as by the reference's sequential calls (Cae3D.py:105-107)."""
        assert tuple(x.shape) == (self.batch, self.cin) + self.in_dims, (tuple(x.shape), self.in_dims)
        self.scratch.zero()
        wait_ev, rec_ev = order if order is not None else (None, None)
        if training and bump_nbt and "__nbt_flat__" in bufs:
            bufs["__nbt_flat__"].add_(self.G)         # every BatchNorm sees G calls
        # every weight re-pack of the stack that depends on the parameters only (forward fragments of the un-folded layers with
        # their biases, data-gradient fragments once the backward exists) in ONE launch -- 40-odd launches and as many bias
        # copies per call otherwise
        O.prep_batch([(l.fwd, params[l.conv_prefix + ".weight"], params[l.conv_prefix + ".bias"]) for l in self.layers
                      if not l.fold and not l.fold_groups and not (self.fused_out and l is self.layers[-1])] +
                     [(l.dgrad, params[l.conv_prefix + ".weight"]) for l in self.layers
                      if l._bwd_ready and getattr(l, "dgrad", None) is not None and l.f8_dgrad is None])
        O.ncdhw_to_cl(x.contiguous(), self.x0, self.dtype)
        if training:
            s0 = self.layers[0].in_sums
            per = s0.numel() // self.G
            for gi in range(self.G):                  # statistics of the stack input, per pass
                O.bn_stats(self.x0[gi * self.gb:(gi + 1) * self.gb], self.dtype, s0[gi * per:(gi + 1) * per])
        h = self.x0
        out = torch.empty((self.batch, self.cout) + self.out_dims, dtype=torch.float32, device=self.device)
        for i, lay in enumerate(self.layers):
            nxt = self.layers[i + 1].in_sums if (training and i + 1 < len(self.layers)) else None
            if wait_ev is not None:
                torch.cuda.current_stream().wait_event(wait_ev[i])
            if self.fused_out and i + 1 == len(self.layers):
                self._out_forward(lay, h, params, bufs, training, out)
            else:
                h = lay.forward(h, params, bufs, training, nxt)
            if rec_ev is not None:
                rec_ev[i].record()
        if not self.fused_out:
            O.cl_to_ncdhw(h, out, self.out_dtype)
        return out

    # ---- the fused output layer (csrc/sp_pwout.hip)
    def _out_coef(self, lay):
        """(pointer, group stride in floats, group batch) of the layer's BatchNorm scale / shift rows"""
        return lay.apply_coef.data_ptr(), 3 * lay.cpi, (self.gb if self.G > 1 else 0)

    def _out_forward(self, lay, x, params, bufs, training, out):
        lay._bn_fwd(params, bufs, training)
        c = lay.conv_prefix
        cp, gs, gb = self._out_coef(lay)
        V = x.numel() // x.shape[-1] // self.batch
        L.call("sp_pwout_fwd", O.ptr(x), self.batch, V, lay.cin, lay.cpi, cp, gs, gb, O.ptr(params[c + ".weight"]), O.ptr(params[c + ".bias"]),
               O.ptr(out), O.stream())

    def _out_backward(self, lay, x, dout, out, params, grads, param_grads):
        """dL/dout, out (NCDHW fp32, one channel) -> g = gradient at the layer's BatchNorm output (bf16, 16 channels), its
        BatchNorm-backward coefficients, and the layer's parameter gradients"""
        c, p = lay.conv_prefix, lay.bn_prefix
        G, dev = self.G, self.device
        if self._out_sums is None:
            self._out_sums = torch.zeros(G * STATS_NREP * 32, dtype=torch.float64, device=dev)
            self._out_g = torch.empty_like(x)
            lay.coef = torch.zeros((G, 3, lay.cpi) if G > 1 else (3, lay.cpi), device=dev)
        self._out_sums.zero_()
        V = x.numel() // x.shape[-1] // self.batch
        cp, gs, gb = self._out_coef(lay)
        w = params[c + ".weight"]
        L.call("sp_pwout_bwd", O.ptr(dout), O.ptr(out), O.ptr(x), self.batch, V, lay.cin, lay.cpi, O.ptr(w), gb, STATS_NREP,
               O.ptr(self._out_g), O.ptr(self._out_sums), O.stream())
        bs = lay.scratch.get(lay.bsums_id)
        L.call("sp_pwout_finish", O.ptr(self._out_sums), STATS_NREP, G, lay.cin, O.ptr(w), cp, gs, O.ptr(bs), STATS_NREP,
               O.ptr(grads[c + ".weight"]) if param_grads else None, O.ptr(grads[c + ".bias"]) if param_grads else None, O.stream())
        if G > 1:
            world = 1
            if SYNC["on"]:
                _allreduce(bs)
                world = SYNC["world"]
            L.call("sp_bn_bwd_finalize_groups", O.ptr(bs), STATS_NREP, float(lay.count * world), O.ptr(params[p + ".weight"]),
                   O.ptr(lay.mean), O.ptr(lay.invstd), lay.cin, lay.cpi, G, O.ptr(grads[p + ".weight"]), O.ptr(grads[p + ".bias"]),
                   O.ptr(lay.coef), 1.0 / world, O.stream())
        else:
            lay._bn_bwd_finalize(bs, params, grads, STATS_NREP)
        return self._out_g, lay.coef

    def prepare(self, params, with_bwd):
        """Pack every weight that depends on the parameters only (forward fragments of the un-folded layers, data-gradient
        fragments) NOW, on the current stream: passes that then run concurrently on other streams find the shared bank
        current and launch no re-pack of their own."""
        for lay in self.layers:
            c = lay.conv_prefix
            if not lay.fold:
                lay.fwd.prep(params[c + ".weight"], params[c + ".bias"])
            if with_bwd:
                lay._init_bwd()
                if getattr(lay, "dgrad", None) is not None and lay.f8_dgrad is None:
                    lay.dgrad.prep(params[c + ".weight"])

    def private_grads(self, names, views):
        """a zeroed gradient buffer of this context shaped like the stack's segment of the flat gradient buffer (concurrent
        passes accumulate privately; Cae3D._StackFn.backward adds the buffers up in stream order)"""
        n = sum(v.numel() for v in views)
        if getattr(self, "_gpriv", None) is None or self._gpriv.numel() != n:
            self._gpriv = torch.empty(n, dtype=torch.float32, device=self.device)
        self._gpriv.zero_()
        out, off = {}, 0
        for k, v in zip(names, views):
            out[k] = self._gpriv[off:off + v.numel()].view(v.shape)
            off += v.numel()
        return self._gpriv, out

    def backward(self, dout, out, params, grads, need_input_grad, param_grads=True):
        """dout = dL/dout (NCDHW fp32).  Accumulates parameter gradients; returns dL/dx (NCDHW fp32) or None.
        param_grads=False: a FROZEN stack (CaePredictionLearner / CaeStepLearner: the gradient passes through the decoder into
        a trainable encoder or into the learned step): data gradients and BatchNorm-backward terms only, `grads` is scratch."""
        dt = self.dtype
        for lay in (self.layers[:-1] if self.fused_out else self.layers):
            lay._init_bwd()
            lay.param_grads = bool(param_grads)
        last = self.layers[-1]
        dout = dout.contiguous()
        gv = (lambda t: (t.numel() // t.shape[-1] // self.G) if self.G > 1 else 0)      # voxels per BatchNorm group of a tensor
        top = len(self.layers) - 1
        if self.fused_out:
            prev = self.layers[-2]
            g, coef = self._out_backward(last, prev.y, dout, out.contiguous(), params, grads, bool(param_grads))
            O.bn_act_bwd(g, prev.y, coef, dt, prev.act, prev.act_param, prev.dz, prev.dbias_sums, group_vox=gv(prev.y), cls=prev.cls_arg())
            top -= 1
        elif self.cout <= 8:
            O.out_grad_to_cl(dout, out, dt, last.act, last.act_param, last.dz, last.dbias_sums)
        else:
            # wide outputs (the latent): channels-last copy of the gradient, then the activation derivative
            if not hasattr(self, "_dy"):
                self._dy = O.alloc_cl(self.batch, self.out_dims, last.cpo, dt, self.device)
            O.ncdhw_to_cl(dout, self._dy, dt)
            O.bn_act_bwd(self._dy, last.y, None, dt, last.act, last.act_param, last.dz, last.dbias_sums)
        for i in range(top, -1, -1):
            lay = self.layers[i]
            x = self.layers[i - 1].y if i > 0 else self.x0
            g, coef = lay.backward(x, params, grads, want_g=(i > 0 or need_input_grad))
            if i > 0:
                prev = self.layers[i - 1]
                # a layer whose weight gradient runs on the raw input needs the border-class sums of its dz from THIS pass: without
                # coefficients (no BatchNorm in front of `lay`) nobody would write them -- no layer table has that case
                assert coef is not None or prev.cls_arg() is None, "raw-input weight gradient of %s: its dz comes without class sums" % prev.name
                O.bn_act_bwd(g, prev.y, coef, dt, prev.act, prev.act_param, prev.dz, prev.dbias_sums, group_vox=gv(prev.y) if coef is not None else 0,
                             cls=prev.cls_arg() if coef is not None else None)
            elif need_input_grad:
                if not hasattr(self, "_dx"):
                    self._dx = O.alloc_cl(self.batch, self.in_dims, lay.cpi, dt, self.device)
                O.bn_act_bwd(g, self.x0, coef, dt, L.ACT_NONE, 0.0, self._dx, None, group_vox=gv(self.x0) if coef is not None else 0)
                dx = torch.empty((self.batch, self.cin) + self.in_dims, dtype=torch.float32, device=self.device)
                O.cl_to_ncdhw(self._dx, dx, dt)
                return dx
        return None


class StackPool:
    """Contexts keyed by (batch, input dims, dtype, device); a context is busy from its forward until its
    backward (or immediately released when no gradient is recorded)."""

    def __init__(self, table, prefix, channels, alpha, last_sigmoid):
        self.table, self.prefix, self.channels, self.alpha, self.last_sigmoid = table, prefix, channels, alpha, last_sigmoid
        self.free = {}
        self.banks = {}

    def acquire(self, batch, in_dims, dtype, device, lane=0, groups=1):
        """lane: index of the pass inside one encoder / decoder call (Cae3D._run_stack_many): pass k always gets the contexts of
        lane k, so the eager warm-up steps build exactly the contexts a captured step replays.  All lanes share the bank of
        packed weights (un-folded fragments depend on the parameters only; ``StackContext.prepare`` fills it once per step
        before concurrent passes fork)."""
        key = (batch, tuple(in_dims), dtype, str(device), lane) + ((groups,) if groups > 1 else ())
        lst = self.free.setdefault(key, [])
        if lst:
            return key, lst.pop()
        return key, StackContext(self.table, self.prefix, self.channels, self.alpha, batch, in_dims, dtype, device,
                                 self.last_sigmoid, bank=self.banks.setdefault((key[:4], groups), {}), groups=groups)

    def release(self, key, ctx):
        lst = self.free.setdefault(key, [])
        if len(lst) < 8:
            lst.append(ctx)

    def clear(self):
        self.free = {}
        self.banks = {}
