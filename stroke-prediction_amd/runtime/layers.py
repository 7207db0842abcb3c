"""One fused network unit: [BatchNorm3d on load] -> Conv3d | ConvTranspose3d -> bias -> activation.

This is the building block of both reference networks (``Block3x3x3`` Unet3D.py:14-27 and the
BN-conv-ELU triples of Cae3D.py:39-76,176-220).  Forward is ONE kernel per sub-convolution
(the BatchNorm is folded into the operand load, bias/activation/next-layer statistics into the
epilogue); backward is wgrad + dgrad + one reduction, with the BatchNorm backward expressed as
per-channel coefficients ``dx = c0*g + c1*x + c2`` that the consumer of ``g`` applies on load.
"""
import ctypes as C
import os

import torch

from . import lib as L
from . import ops as O
from . import plan as P

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# exact data-parallel mode (parallel.DataParallelSync(mode="exact")): BatchNorm sums are all-reduced so every rank
# normalises with the statistics of the GLOBAL batch, as the single-process reference does
SYNC = {"group": None, "world": 1, "on": False, "direct": None}


def _allreduce(t):
    """sum of an accumulator tensor over the ranks, in place, visible to what the current stream runs next.  With a
    communicator of our own (parallel.DirectComm: ``direct``) the collective is an ``sp_allreduce_flat[_f64]`` call on that
    communicator's stream, forked from and joined to the current one -- a node like any other inside a captured step;
    otherwise torch.distributed's (its NCCL collectives run on the process group's stream: eager steps only)."""
    d = SYNC.get("direct")
    if d is not None:
        d.all_reduce_async(t)
        d.wait()
        return
    import torch.distributed as dist
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=SYNC["group"])


STATS_NREP = int(os.environ.get("SP_STATS_NREP", "64"))     # replicas of the conv-epilogue statistics accumulators (spreads same-address fp64 atomics)


class Scratch:
    """A flat fp64 scratch arena for all per-step reduction accumulators: one memset per step."""

    def __init__(self, device):
        self.device = device
        self.sizes = []
        self.buf = None

    def reserve(self, n):
        self.sizes.append(n)
        return len(self.sizes) - 1

    def finalize(self):
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        self.buf = torch.zeros(max(1, self.offsets[-1]), dtype=torch.float64, device=self.device)

    def get(self, idx):
        return self.buf[self.offsets[idx]:self.offsets[idx + 1]]

    def zero(self):
        self.buf.zero_()


class ConvLayer:
    def __init__(self, name, kind, cin, cout, k, stride, pad, in_dims, batch, dtype, device, scratch,
                 bn_prefix=None, conv_prefix=None, act=L.ACT_NONE, act_param=0.0, out_dtype=None,
                 need_input_grad=True, cpi=None, bank=None, split_g=None, pitch=8, groups=1, hl=False, pooled=False):
        """hl (the "bf16x3" precision mode): the FORWARD convolution takes and writes bf16 pairs (SP_HL: x = x_hi + x_lo in two
        tensors, three MFMAs per product); ``self.y`` is the hi half -- the tensor the bf16 mode would have stored -- and the
        backward side below is the bf16 one on the hi tensors, unchanged.
        groups > 1 (the batched passes of the CAE, runtime/cae_engine.py): the batch holds `groups` passes of batch // groups
        samples each; every pass is its own BatchNorm group (own batch statistics, own scale / shift / backward coefficients),
        the convolution, its weight gradient and its data gradient run ONCE over the whole batch.  The normalised input is
        always written out per group (``xhat``), so no kernel needs an affine on its operand load and the weights never
        depend on a pass."""
        self.G = int(groups)
        self.hl = bool(hl)
        assert not hl or (dtype == L.SP_BF16 and kind == "conv" and groups == 1 and bank is None), "bf16 pairs: un-batched bf16-storage convolutions"
        assert batch % self.G == 0
        self.gb = batch // self.G
        self.name, self.kind = name, kind
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.in_dims, self.batch, self.dtype, self.device = tuple(in_dims), batch, dtype, device
        self.bn_prefix, self.conv_prefix = bn_prefix, conv_prefix
        self.act, self.act_param = act, act_param
        self.out_dtype = dtype if out_dtype is None else out_dtype
        self.need_input_grad = need_input_grad
        self.x_planar = False       # set by the engine: the input is a plane-major concat buffer
        self.split_g = split_g      # channel count of the first part of a concatenated input: its gradient and the rest's go to two dense tensors
        # channel pitch of this layer's tensors: multiples of `pitch` (16 lets every 3x3x3 layer use the DMA kernels)
        self.cpi, self.cpo = (cpi or O.cpad(cin, pitch)), O.cpad(cout, pitch)
        mk = P.conv_fwd_op if kind == "conv" else P.convT_fwd_op
        self.fwd_op = mk(cin, cout, k, stride, pad, in_dims, self.cpi, self.cpo, L.SP_HL if hl else dtype)
        self.out_dims = tuple(self.fwd_op.y_dims)
        self.bank = bank            # optional dict shared by the layers of all contexts of one stack (packed weights)
        self.count = float(self.gb * in_dims[0] * in_dims[1] * in_dims[2])       # voxels of ONE BatchNorm group
        # un-padded bf16 convolutions fold the BatchNorm into weights/bias so the tile can be staged by DMA
        pads = pad if isinstance(pad, (tuple, list)) else (pad,) * 3
        self.fold = bool(bn_prefix is not None and kind == "conv" and dtype == L.SP_BF16 and max(pads) == 0
                         and (hl or all(s.tile["dma"] for s in self.fwd_op.subs)) and self.G == 1)
        assert not hl or self.fold or bn_prefix is None, "bf16 pairs: BatchNorm folded into the weights (un-padded convolutions)"
        self.scratch = scratch
        # Padded bf16 convolutions behind a BatchNorm (the CAE): zero padding applies AFTER the normalisation, so the
        # BatchNorm cannot be folded into the weights, and a DMA cannot normalise on load.  The normalised input is
        # therefore written once (one elementwise pass) and both the forward conv and the weight gradient run on the
        # DMA kernels with plain zero-fill (register-staged kernels: 2-2.5x slower per voxel).
        strides = stride if isinstance(stride, (tuple, list)) else (stride,) * 3
        self.materialize = bool(bn_prefix is not None and kind == "conv" and dtype == L.SP_BF16 and not self.fold
                                and max(pads) > 0 and max(pads) <= 2 and max(strides) == 1 and k == 3 and self.cpi % 16 == 0
                                and self.cpo % 16 == 0 and all(s.tile["dma"] for s in self.fwd_op.subs) and O.MATERIALIZE_BN)
        if self.G > 1:
            self.materialize = bn_prefix is not None        # grouped: xhat = s_g x + t_g for EVERY layer behind a BatchNorm
        if hl:
            self.materialize = False
        self.xhat = None
        self.y_lo = None
        # folded layers run without affine-on-load and with plain statistics: candidates for the z-marching kernel; so do the
        # materialised ones (the kernel pads from its zero page and has an ELU epilogue)
        zm_ok = (self.fold and act in (L.ACT_NONE, L.ACT_LEAKY) and bank is None) or \
                (self.materialize and act == L.ACT_ELU and O.ZM_CAE and self.out_dtype == dtype)
        # (folded fragments depend on the BatchNorm statistics of the pass: never shared between the contexts of a bank)
        if self.G > 1:
            zm_ok = bool(self.materialize and kind == "conv" and dtype == L.SP_BF16 and act == L.ACT_ELU and O.ZM_CAE
                         and self.out_dtype == dtype and max(pads) <= 2 and max(strides) == 1 and k == 3
                         and self.cpi % 16 == 0 and self.cpo % 16 == 0)
        # (groups: ONE launch over the whole batch -- the kernel hands its statistics to the rows of a sample's group)
        # pooled: MaxPool3d(2) follows this layer (the second convolution of a down block): the forward kernel keeps the classic tile,
        # the shape its pooling epilogue is written for
        self.fwd = O.ConvRunner(self.fwd_op, device, share=None if (bank is None or self.fold) else bank.setdefault((name, "fwd"), {}),
                                zm_batch=(self.batch if (self.G > 1 and O.ZM_GROUPS) else self.gb) if zm_ok else None,
                                zm_tile="classic" if (pooled and O.FUSE_POOL) else None)
        # batched passes on the z-marching kernel: the BatchNorm folded into per-group weight fragments and a bias table over the
        # border classes of the padded output (sp_conv_prep_folded_groups): the forward reads the RAW input, the normalised copy
        # (still the weight gradient's operand) is written later, on the side stream of the backward
        self.fold_groups = bool(self.G > 1 and O.FOLD_GROUPS and O.ZM_GROUPS and zm_ok and self.fwd.zm is not None and bn_prefix is not None and kind == "conv"
                                and act == L.ACT_ELU and dtype == L.SP_BF16 and (2 * pads[0] + 1) * (2 * pads[1] + 1) * (2 * pads[2] + 1) <= 75)
        self.raw_wgrad = bool(self.fold_groups and O.RAW_WGRAD and O.WGRAD_PARTS and tuple(P._triple(stride)) == (1, 1, 1) and k == 3
                              and self.cpi % 16 == 0 and self.cpo % 16 == 0 and not os.environ.get("SP_WGRAD_ZR") == "0")
        # ... and the 1x1x1 layers (Cae3D.py:214-216): the pointwise kernel applies each group's scale / shift on its operand load
        # (exact: no padding), the weight gradient reads the raw input the same way
        self.raw_pw = bool(self.G > 1 and O.RAW_WGRAD and O.WGRAD_PARTS and O.USE_PW_WGRAD and O.ZM_GROUPS and bn_prefix is not None and kind == "conv" and k == 1
                           and tuple(P._triple(stride)) == (1, 1, 1) and max(pads) == 0 and dtype == L.SP_BF16 and act == L.ACT_ELU
                           and self.out_dtype == dtype and self.fwd.fc is not None and self.fwd.fc["pointwise"]
                           and (self.gb * in_dims[0] * in_dims[1] * in_dims[2]) % 32 == 0)
        if self.raw_pw:
            self.raw_wgrad = True
            self._ncls, self._pads = 1, (0, 0, 0)
        if self.fold_groups:
            z = self.fwd.zm
            self._gfrag_elems = z["nsteps"] * z["NT"] * 64 * 8
            self.gfrag = torch.empty(self.G * self._gfrag_elems, dtype=torch.bfloat16, device=device)
            self._ncls = (2 * pads[0] + 1) * (2 * pads[1] + 1) * (2 * pads[2] + 1)
            self.gtab = torch.empty(self.G * self._ncls * self.cpo, dtype=torch.float32, device=device)
            self._pads = tuple(pads)
        if bn_prefix is not None:
            G = self.G
            self.apply_coef = torch.zeros((G, 3, self.cpi) if G > 1 else (3, self.cpi), device=device)     # (scale, 0, shift): rows 0 and 2 ARE scale / shift
            self.scale = self.apply_coef[..., 0, :] if G > 1 else self.apply_coef[0]
            self.shift = self.apply_coef[..., 2, :] if G > 1 else self.apply_coef[2]
            self.mean = torch.zeros((G, self.cpi) if G > 1 else (self.cpi,), device=device)
            self.invstd = torch.zeros((G, self.cpi) if G > 1 else (self.cpi,), device=device)
            self.in_sums_id = scratch.reserve(G * self.cpi * 2 * STATS_NREP)
        else:
            self.scale = self.shift = None
        # backward side is created lazily (inference never pays for it)
        self._bwd_ready = False
        self.param_grads = True      # False: backward through a FROZEN layer (phase-2 learners): data gradient and BatchNorm-backward
        #                              coefficients only -- no weight-gradient kernel, no gamma / beta / bias gradients kept
        self.y = None
        # fp8 execution (runtime/f8.py; enabled per layer by the engine in the "fp8" precision mode)
        self.f8_fwd = self.f8_dgrad = self.f8_wgrad = None
        self.f8_on = self.f8_wgrad_only = False
        self.x8 = self.y8 = self.dz8 = None
        self.dz8_ready = False      # set by the kernel that formed dz when it also wrote the fp8 copy (dz8_out)
        self.want_y8 = False
        self.store_y = True       # False (engine, fp8 mode): the 16-bit output is not written, its readers take self.y8
        self.f8_grad_scale = 1.0

    def enable_f8(self, grad_scale, fwd=True):
        """fp8 MFMA operands for this layer's forward and data-gradient convolution where the fp8 z-marching kernel has an
        instance for its shape (folded BatchNorm, stride 1, 3x3x3, 32..96 input channels); returns whether the forward runs
        in fp8.  The engine then provides ``self.x8`` (e4m3 plane-major copy of the input) before every forward.
        fwd=False (the "fp8b" mode): the forward stays on the bf16 kernel, data and weight gradient run in fp8."""
        from . import f8 as F8
        self.f8_on = True
        self.f8_grad_scale = float(grad_scale)
        if (fwd and F8.FWD and self.fold and self.kind == "conv" and self.dtype == L.SP_BF16 and self.out_dtype == L.SP_BF16 and self.bank is None
                and self.act in (L.ACT_NONE, L.ACT_LEAKY)):
            if F8.ConvRunnerF8.applicable(self.fwd_op, self.batch):
                self.f8_fwd = F8.ConvRunnerF8(self.fwd_op, self.device, self.batch, F8.E4M3)
            elif F8.ConvRunnerF8Split.applicable(self.fwd_op, self.batch):      # 12 / 16 / 24 input planes: groups of planes
                self.f8_fwd = F8.ConvRunnerF8Split(self.fwd_op, self.device, self.batch, F8.E4M3)
        # no fp8 forward instance (more input planes than the ring holds, too few columns): the weight gradient can still
        # run on fp8 copies of the two tensors -- the engine then fills x8 in training steps only
        self.f8_wgrad_only = bool(self.f8_fwd is None and F8.WGRAD and F8.WGRAD_ONLY and F8.DZ_FMT == F8.E5M2 and self.fold
                                  and self.kind == "conv" and self.dtype == L.SP_BF16 and self.bank is None and self.G == 1
                                  and self.k == 3 and max(P._triple(self.stride)) == 1 and max(P._triple(self.pad)) == 0
                                  and self.cpi % 32 == 0 and self.cpo % 32 == 0 and self.cin == self.cpi and self.cout == self.cpo)
        return self.f8_fwd is not None

    def dz8_out(self):
        """(tensor, format, scale) for the fp8 shadow output of the kernel that forms this layer's dz, or None: the data
        gradient then needs no quantisation pass.  Call after _init_bwd."""
        if (self.f8_dgrad is None and self.f8_wgrad is None) or not self.FUSE_Q8:
            return None
        from . import f8 as F8
        self.dz8_ready = True
        return (self.dz8, F8.DZ_FMT, self.f8_grad_scale)

    def dz_target(self):
        """where the kernel that forms this layer's dz stores the 16-bit tensor: ``self.dz``, or None when it also writes the fp8
        copy (dz8_out) and both convolutions of this layer's backward read that one.  Call after _init_bwd."""
        if (self.SKIP_DZ and self.FUSE_Q8 and self.f8_wgrad is not None
                and (self.f8_dgrad is not None or not self.need_input_grad)):
            return None
        return self.dz

    SKIP_DZ = not os.environ.get("SP_F8_KEEP_DZ")      # (A/B knob: the 16-bit dz is always stored)
    FUSE_Q8 = not os.environ.get("SP_F8_NO_FUSE")      # (A/B knob: every fp8 operand by a separate sp_quantize_f8 pass)

    def y8_capable(self):
        """this layer's forward kernel can write the e4m3 plane-major copy of its output next to the bf16 one"""
        if self.f8_fwd is not None:
            return True
        return bool(self.FUSE_Q8 and self.kind == "conv" and self.G == 1 and not self.materialize and self.out_dtype == L.SP_BF16
                    and self.act in (L.ACT_NONE, L.ACT_LEAKY) and self.fwd.zm_y8_ok())

    def alloc_y8(self):
        from . import f8 as F8
        if self.y8 is None:
            self.y8 = F8.alloc_f8(self.batch, self.out_dims, self.cpo, self.device)
        return self.y8

    # ---------------------------------------------------------------- forward
    @property
    def in_sums(self):
        return self.scratch.get(self.in_sums_id)

    def alloc_out(self):
        if self.y is None:
            if self.hl:      # the pair: one allocation, hi half first
                pair = torch.empty((2, self.batch) + tuple(self.out_dims) + (self.cpo,), dtype=O.TORCH_DT[L.SP_BF16], device=self.device)
                self.y, self.y_lo = pair[0], pair[1]
            else:
                self.y = O.alloc_cl(self.batch, self.out_dims, self.cpo, self.out_dtype, self.device)
        return self.y

    def _bn_fwd(self, params, bufs, training, fused=False):
        """Finalize the input BatchNorm (batch statistics from in_sums, running buffers) into scale/shift.
        fused: the finalize itself runs inside the weight re-pack kernel -- returns its arguments (lib.BnFinArgs) instead of
        launching sp_bn_finalize."""
        p = self.bn_prefix
        world = 1
        if training and SYNC["on"]:
            _allreduce(self.in_sums)
            world = SYNC["world"]
        if fused:
            assert self.G == 1
            f = L.BnFinArgs()
            f.sums = O.ptr(self.in_sums if training else None)
            f.gamma, f.beta = O.ptr(params[p + ".weight"]), O.ptr(params[p + ".bias"])
            f.running_mean, f.running_var = O.ptr(bufs[p + ".running_mean"]), O.ptr(bufs[p + ".running_var"])
            f.scale, f.shift, f.mean, f.invstd = O.ptr(self.scale), O.ptr(self.shift), O.ptr(self.mean), O.ptr(self.invstd)
            f.count, f.momentum, f.eps = float(self.count * world), BN_MOMENTUM, BN_EPS
            f.nrep, f.training, f.C, f.CP = STATS_NREP, int(training), self.cin, self.cpi
            if training and "__nbt_flat__" not in bufs:
                bufs[p + ".num_batches_tracked"].add_(1)
            return f
        if self.G > 1:
            L.call("sp_bn_finalize_groups", O.ptr(self.in_sums if training else None), STATS_NREP, float(self.count * world),
                   O.ptr(params[p + ".weight"]), O.ptr(params[p + ".bias"]), O.ptr(bufs[p + ".running_mean"]),
                   O.ptr(bufs[p + ".running_var"]), BN_MOMENTUM, BN_EPS, int(training), self.cin, self.cpi, self.G, 3 * self.cpi,
                   self.apply_coef.data_ptr(), self.apply_coef.data_ptr() + 8 * self.cpi, O.ptr(self.mean), O.ptr(self.invstd),
                   O.stream())
        else:
            O.bn_finalize(self.in_sums if training else None, self.count * world, params[p + ".weight"], params[p + ".bias"],
                          bufs[p + ".running_mean"], bufs[p + ".running_var"], BN_MOMENTUM, BN_EPS, training,
                          self.cin, self.cpi, self.scale, self.shift, self.mean, self.invstd, nrep=STATS_NREP)
        if training and "__nbt_flat__" not in bufs:      # else: one increment for all BatchNorms (UnetEngine.forward)
            bufs[p + ".num_batches_tracked"].add_(1)

    def can_pool(self):
        """forward(pool=...) applies: the forward kernel has the MaxPool3d(2) epilogue for this layer's shape"""
        return bool(self.G == 1 and self.fold and self.f8_fwd is None and not self.want_y8 and self.store_y and self.kind == "conv"
                    and self.act in (L.ACT_NONE, L.ACT_LEAKY) and self.out_dtype == self.dtype and self.fwd.zm_pool_ok())

    def forward(self, x, params, bufs, training, out_stats=None, x_lo=None, pool=None):
        """x: channels-last input (hl: its hi half, x_lo the lo half); returns the (cached) output tensor (hl: its hi half, the
        lo half is self.y_lo).
        pool = (pooled, pooled_lo | None) (``can_pool()``): the kernel also writes MaxPool3d(2) of the output, and out_stats receives
        the statistics of the pooled tensor."""
        # folded layers: the BatchNorm finalize rides in the re-pack kernel of the weights it is folded into
        bn = None
        if self.bn_prefix is not None:
            fuse = bool(self.fold and self.G == 1 and self.f8_fwd is None and self.fwd.can_fuse_bn())
            bn = self._bn_fwd(params, bufs, training, fused=fuse)
        c = self.conv_prefix
        y = self.alloc_out()
        if self.hl:
            assert self.fold and x_lo is not None
            self.fwd.prep(params[c + ".weight"], params[c + ".bias"], self.scale, self.shift, bn=bn)
            self.fwd.run(x, y, self.batch, None, None, self.act, self.act_param, out_stats, dtype_out=L.SP_HL,
                         stats_nrep=STATS_NREP, x_planar=self.x_planar, x_lo=x_lo, y_lo=self.y_lo, pool=pool)
            return y
        assert pool is None or (self.fold and self.f8_fwd is None), "pool: folded bf16 layers (can_pool)"
        if self.G > 1 and self.fold_groups:
            op, z = self.fwd_op, self.fwd.zm
            L.call("sp_conv_prep_folded_groups", O.ptr(params[c + ".weight"]), op.w_sco, op.w_sci, op.cout, op.cin, O.ptr(z["kmap_d"]), z["nsteps"],
                   z["NT"], O.ptr(self.gfrag), self._gfrag_elems * 2, self.apply_coef.data_ptr(), 3 * self.cpi, self.cpi, self.G,
                   O.ptr(params[c + ".bias"]), self._pads[0], self._pads[1], self._pads[2], O.ptr(self.gtab), self.cpo, O.stream())
            self.fwd.run(x, y, self.batch, None, None, self.act, self.act_param, out_stats, dtype_out=self.out_dtype,
                         stats_nrep=STATS_NREP, group_batch=self.gb, group_fold=(self.gfrag, self._gfrag_elems * 2, self.gtab, self._ncls * self.cpo))
            return y
        if self.G > 1 and self.raw_pw:
            self.fwd.prep(params[c + ".weight"], params[c + ".bias"])
            self.fwd.run(x, y, self.batch, self.apply_coef[:, 0], self.apply_coef[:, 2], self.act, self.act_param, out_stats, dtype_out=self.out_dtype,
                         stats_nrep=STATS_NREP, group_batch=self.gb, coef_gstride=3 * self.cpi)
            return y
        if self.G > 1:
            src = x
            if self.materialize:
                if self.xhat is None:
                    self.xhat = torch.empty_like(x)
                O.bn_act_bwd(x, x, self.apply_coef, self.dtype, L.ACT_NONE, 0.0, self.xhat, None,
                             group_vox=x.numel() // x.shape[-1] // self.G)      # xhat = scale_g * x + shift_g
                src = self.xhat
            self.fwd.prep(params[c + ".weight"], params[c + ".bias"])
            self.fwd.run(src, y, self.batch, None, None, self.act, self.act_param, out_stats, dtype_out=self.out_dtype,
                         stats_nrep=STATS_NREP, group_batch=self.gb)
            return y
        if self.materialize:
            if self.xhat is None:
                self.xhat = torch.empty_like(x)
            O.bn_act_bwd(x, x, self.apply_coef, self.dtype, L.ACT_NONE, 0.0, self.xhat, None)      # xhat = scale*x + shift
            self.fwd.prep(params[c + ".weight"], params[c + ".bias"])
            self.fwd.run(self.xhat, y, self.batch, None, None, self.act, self.act_param, out_stats, dtype_out=self.out_dtype,
                         stats_nrep=STATS_NREP)
            return y
        if self.f8_fwd is not None:       # fp8 operands: x8 = e4m3 copy of x (engine), BatchNorm folded into the e4m3 weights
            self.f8_fwd.prep(params[c + ".weight"], params[c + ".bias"], self.scale, self.shift)
            y8 = self.alloc_y8() if self.want_y8 else None
            if not self.store_y:      # (set by the engine: every reader of this output takes the e4m3 copy)
                assert y8 is not None
                self.f8_fwd.run(self.x8, y, self.act, self.act_param, out_stats, STATS_NREP, y8=y8, store=False)
            else:
                self.f8_fwd.run(self.x8, y, self.act, self.act_param, out_stats, STATS_NREP, y8=y8)
        elif self.fold:
            self.fwd.prep(params[c + ".weight"], params[c + ".bias"], self.scale, self.shift, bn=bn)
            self.fwd.run(x, y, self.batch, None, None, self.act, self.act_param, out_stats, dtype_out=self.out_dtype,
                         stats_nrep=STATS_NREP, x_planar=self.x_planar, y8=self.alloc_y8() if self.want_y8 else None, pool=pool)
        else:
            self.fwd.prep(params[c + ".weight"], params[c + ".bias"])
            self.fwd.run(x, y, self.batch, self.scale, self.shift, self.act, self.act_param, out_stats,
                         dtype_out=self.out_dtype, stats_nrep=STATS_NREP, y8=self.alloc_y8() if self.want_y8 else None)
        return y

    # ---------------------------------------------------------------- backward
    def _init_bwd(self):
        if self._bwd_ready:
            return
        kk = self.k ** 3
        cin, cout, k, s, p = self.cin, self.cout, self.k, self.stride, self.pad
        dev, dt = self.device, self.dtype
        self.dz = O.alloc_cl(self.batch, self.out_dims, self.cpo, dt, dev)
        self.dbias_id = None
        if self.kind == "conv":
            self.wgrad = O.WgradRunner(cin, cout, k, s, p, self.in_dims, self.out_dims, self.cpi, self.cpo,
                                       cin * kk, kk, dt, dev)
            dop = P.conv_dgrad_op(cin, cout, k, s, p, self.in_dims, self.cpo, self.cpi, dt)
        else:
            # roles swap: shifted operand = dz (output grid), fixed operand = normalised input
            self.wgrad = O.WgradRunner(cout, cin, k, s, p, self.out_dims, self.in_dims, self.cpo, self.cpi,
                                       cout * kk, kk, dt, dev)
            dop = P.convT_dgrad_op(cin, cout, k, s, p, self.in_dims, self.cpo, self.cpi, dt)
        # Folded DMA weight-gradient path: the BatchNorm-backward sums (sum g, sum g*x) come out of the raw-input
        # weight-gradient accumulator (sum_v g*x = sum W*acc, sum_v g = sum W*dbias for an un-padded convolution), so
        # the data-gradient convolution needs no statistics epilogue (no second read of x, no atomics) -- and is not
        # run at all for the first layer of a network, whose input gradient nobody wants.
        self.bn_from_wgrad = bool(self.bn_prefix is not None and self.kind == "conv" and self.wgrad.folds(self.scale)
                                  and max(P._triple(p)) == 0 and O.BN_SUMS_FROM_WGRAD and self.G == 1)
        self.split_one = False
        if self.split_g and self.kind == "conv" and self.need_input_grad and self.bn_from_wgrad and self.split_g % 16 == 0 and (cin - self.split_g) % 16 == 0 \
                and self.cpi == cin and dt == L.SP_BF16 and not self.f8_on:
            # ONE launch, two dense tensors (sp_conv3d_zm with y2: round 5) where the z-marching instance exists
            probe = O.ConvRunner(dop, dev, zm_batch=self.batch if self.bank is None else None)
            if probe.zm_split_ok():
                self.dgrad, self.split_one = probe, True
                self.g_parts = [O.alloc_cl(self.batch, self.in_dims, self.split_g, dt, dev), O.alloc_cl(self.batch, self.in_dims, cin - self.split_g, dt, dev)]
                self.g = tuple(self.g_parts)
        if self.split_one:
            pass
        elif self.split_g and self.kind == "conv" and self.need_input_grad and self.bn_from_wgrad and os.environ.get("SP_SPLIT_G"):
            # gradient of a channel-concatenated input as one dense tensor per part: both consumers (upsample backward,
            # pool/skip backward) then read whole lines instead of 64 / 32 bytes of every 96-byte row
            self.dgrad_parts, self.g_parts = [], []
            for ci0, cn in ((0, self.split_g), (self.split_g, cin - self.split_g)):
                pop = P.conv_dgrad_op(cn, cout, k, s, p, self.in_dims, self.cpo, O.cpad(cn), dt, cin_total=cin)
                self.dgrad_parts.append((O.ConvRunner(pop, dev), ci0 * kk))
                self.g_parts.append(O.alloc_cl(self.batch, self.in_dims, O.cpad(cn), dt, dev))
            self.g = tuple(self.g_parts)
        elif self.need_input_grad or (self.bn_prefix is not None and not self.bn_from_wgrad):
            # (BatchNorm sums from the weight gradient: the data gradient is a plain convolution -> z-marching candidate)
            # batched passes behind a BatchNorm (the CAE): the z-marching data gradient with the (sum g, sum g x) epilogue,
            # one launch over all groups -- where that instance exists; else the tiled kernel's stats_mode 1
            zm_grouped = bool(self.G > 1 and O.ZM_GROUPS and O.ZM_CAE and self.kind == "conv" and dt == L.SP_BF16 and self.bn_prefix is not None
                              and O.ConvRunner.zm_plan_bn_bwd_ok(P.zm_plan(dop)))
            self.dgrad = O.ConvRunner(dop, dev, share=None if self.bank is None else self.bank.setdefault((self.name, "dgrad"), {}),
                                      zm_batch=self.batch if ((self.bn_from_wgrad and self.bank is None) or zm_grouped) else None)
            self.g = O.alloc_cl(self.batch, self.in_dims, self.cpi, dt, dev)
            if self.f8_on and self.bn_from_wgrad and self.need_input_grad and self.kind == "conv":
                from . import f8 as F8      # plain data gradient (no statistics epilogue): fp8 candidate, dz as e5m2
                if F8.DGRAD and F8.ConvRunnerF8.applicable(dop, self.batch):
                    self.f8_dgrad = F8.ConvRunnerF8(dop, dev, self.batch, F8.DZ_FMT)
                elif F8.DGRAD and F8.ConvRunnerF8Split.applicable(dop, self.batch):
                    self.f8_dgrad = F8.ConvRunnerF8Split(dop, dev, self.batch, F8.DZ_FMT)
                if self.f8_dgrad is not None:
                    self.dz8 = F8.alloc_f8(self.batch, self.out_dims, self.cpo, dev)
        if (self.f8_fwd is not None or self.f8_wgrad_only) and self.x8 is not None and self.bn_from_wgrad:
            from . import f8 as F8      # weight gradient from the fp8 copies both other convolutions of the layer use
            if F8.WgradRunnerF8.applicable(self.wgrad):
                self.f8_wgrad = F8.WgradRunnerF8(self.wgrad)
                if self.dz8 is None:
                    self.dz8 = F8.alloc_f8(self.batch, self.out_dims, self.cpo, dev)
        if self.bn_prefix is not None:
            self.coef = torch.zeros((self.G, 3, self.cpi) if self.G > 1 else (3, self.cpi), device=dev)
        self._bwd_ready = True

    def reserve_bwd_scratch(self):
        self.dbias_sums_id = self.scratch.reserve(self.cpo * L.SP_REDUCE_ROWS)     # replica rows, see include/stroke_amd.h
        if getattr(self, "raw_wgrad", False):
            self.cls_sums_id = self.scratch.reserve(self.G * self._ncls * self.cpo)      # border-class sums of dz per group
        if self.bn_prefix is not None:
            self.bsums_id = self.scratch.reserve(self.G * self.cpi * 2 * STATS_NREP)

    @property
    def dbias_sums(self):
        return self.scratch.get(self.dbias_sums_id).view(L.SP_REDUCE_ROWS, self.cpo)

    def can_fuse_dz(self, producer):
        """backward(fuse_dz=...) applies: this layer's data gradient runs on a z-marching instance with the BatchNorm / activation
        backward epilogue (sp_conv3d_zm stats_mode 2), its BatchNorm-backward sums come from the weight gradient, and `producer`
        (the layer whose output is this layer's input) takes a plain 16-bit dz"""
        if not (O.FUSE_DZ and self._bwd_ready and producer._bwd_ready):
            return False
        return bool(self.G == 1 and self.param_grads and self.bn_from_wgrad and self.need_input_grad and self.dtype == L.SP_BF16
                    and self.f8_wgrad is None and self.f8_dgrad is None and not getattr(self, "dgrad_parts", None)
                    and getattr(self, "dgrad", None) is not None and self.dgrad.zm_bn_bwd_ok()
                    and producer.store_y and producer.dz_target() is producer.dz and producer.dz8_out() is None
                    and producer.act in (L.ACT_LEAKY, L.ACT_NONE) and producer.y is not None and producer.y.shape[-1] == self.cpi)

    def backward(self, x, params, grads, want_g=True, fuse_dz=None):
        """Given self.dz (gradient at the pre-activation output) and self.dbias_sums already filled by the
        producer of dz: accumulate parameter gradients, return (g, coef) describing the input gradient
        dx = coef0*g + coef1*x + coef2 (coef None: dx = g).
        fuse_dz = (dz_out, dz_sums, act, act_param) (``can_fuse_dz``): the data gradient is not stored -- its kernel's epilogue
        applies this layer's BatchNorm backward and the producer's activation derivative (x is the producer's output) and writes the
        producer's dz and sum dz; returns (None, None).
        want_g=False: nobody reads g (the first layer of a stack whose input needs no gradient) -- honoured where the
        BatchNorm-backward sums do not come from the data gradient (the raw-input weight gradient of the batched CAE layers)."""
        self._want_g = bool(want_g)
        c = self.conv_prefix
        w = params[c + ".weight"]
        if not self.param_grads:
            # frozen parameters, live BatchNorm statistics (CaePredictionLearner.py:27 freezes the CAE, Learner.run_training keeps
            # it in train mode): the input gradient still carries the batch-statistics terms, so the (sum g, sum g x) reductions
            # stay; `grads` is a scratch dictionary whose gamma / beta entries are discarded
            assert getattr(self, "dgrad", None) is not None and not getattr(self, "dgrad_parts", None) and self.f8_dgrad is None
            if self.G > 1:
                return self._backward_grouped(x, w, params, grads, wgrad=False)
            return self._backward_input(x, w, params, grads)
        if self.G > 1:
            return self._backward_grouped(x, w, params, grads)
        # the wgrad finish kernel also adds the bias gradient (sum of dz) and re-zeroes its accumulator
        if self.bn_from_wgrad:
            bs = self.scratch.get(self.bsums_id)
            if self.f8_wgrad is not None:
                self._ensure_dz8()
                run_wgrad = lambda: self.f8_wgrad.run(self.x8, self.dz8, self.batch, self.f8_grad_scale, grads[c + ".weight"],
                                                      self.scale, self.shift, self.dbias_sums, grads[c + ".bias"],
                                                      bn_w=w, bn_sums=bs, bn_nrep=STATS_NREP)
            else:
                run_wgrad = lambda: self.wgrad.run(x, self.dz, self.batch, grads[c + ".weight"], self.scale, self.shift,
                                                   dbias_sums=self.dbias_sums, dbias_grad=grads[c + ".bias"], nbias=self.cout,
                                                   bn_w=w, bn_sums=bs, bn_nrep=STATS_NREP, defer_finish=True, x_planar=self.x_planar)
            if fuse_dz is not None:
                # the data gradient's prologue finalizes the BatchNorm backward from the sums of the finish kernel: one stream, in order
                run_wgrad()()
                world = 1
                if SYNC["on"]:
                    _allreduce(bs)
                    world = SYNC["world"]
                p = self.bn_prefix
                dz_out, dz_sums, act, ap = fuse_dz
                self.dgrad.prep(w)
                self.dgrad.run(self.dz, dz_out, self.batch, None, None, act, ap, None, stats_mode=2, aux=x, dz_sums=dz_sums,
                               bnb=dict(sums=bs, nrep=STATS_NREP, count=self.count * world, gamma=params[p + ".weight"], mean=self.mean,
                                        invstd=self.invstd, C=self.cin, CP=self.cpi, dgamma=grads[p + ".weight"], dbeta=grads[p + ".bias"],
                                        pscale=1.0 / world))
                self.dz8_ready = False
                return None, None
            whole = O.overlap_level() == 2      # the weight-gradient kernel itself runs beside the data gradient
            finish = None if whole else run_wgrad()
            # finish + BatchNorm-backward finalize on the side stream, beside the data-gradient convolution
            f = O.fork()
            with f:
                if whole:
                    finish = run_wgrad()
                finish()
                self._bn_bwd_finalize(bs, params, grads, STATS_NREP)
            if self.need_input_grad:
                self._run_dgrad(w)
            self.dz8_ready = False
            f.join()
            return (self.g, self.coef) if self.need_input_grad else (None, None)
        # every other layer: the weight gradient (kernel + finish) is independent of the data gradient as well -- second
        # stream while a graph is captured (ops.overlap_level)
        f = O.fork() if O.overlap_level() == 2 else None
        if f is not None:
            f.__enter__()
        try:
            if self.kind == "conv" and self.materialize:
                self.wgrad.run(self.xhat, self.dz, self.batch, grads[c + ".weight"], None, None,
                               dbias_sums=self.dbias_sums, dbias_grad=grads[c + ".bias"], nbias=self.cout)
            elif self.kind == "conv":
                self.wgrad.run(x, self.dz, self.batch, grads[c + ".weight"], self.scale, self.shift,
                               dbias_sums=self.dbias_sums, dbias_grad=grads[c + ".bias"], nbias=self.cout)
            else:
                self.wgrad.run(self.dz, x, self.batch, grads[c + ".weight"], None, None, self.scale, self.shift,
                               dbias_sums=self.dbias_sums, dbias_grad=grads[c + ".bias"], nbias=self.cout)
        finally:
            if f is not None:
                f.__exit__(None, None, None)
        try:
            return self._backward_input(x, w, params, grads)
        finally:
            if f is not None:
                f.join()

    def cls_arg(self):
        """what the pass that forms this layer's dz needs when the weight gradient reads the raw input (ops.bn_act_bwd(cls=...))"""
        if not (self.raw_wgrad and self.param_grads):
            return None
        return (self.gb, self._pads, self.scratch.get(self.cls_sums_id))

    def _backward_grouped_raw(self, x, w, params, grads):
        """batched passes, BatchNorm folded per group, padded convolution: the weight gradient on the RAW input into group-pure
        partial blocks; its finish (side stream) applies each group's scale / shift with the border-class sums of dz
        (sp_wgrad_finish_folded_groups) and yields the BatchNorm-backward sums, so the data gradient beside it is a plain one"""
        c, p = self.conv_prefix, self.bn_prefix
        wg = self.wgrad
        wg.groups = self.G
        bs = self.scratch.get(self.bsums_id)
        f = O.fork()
        with f:
            wg.run_raw(x, self.dz, self.batch)
            L.call("sp_wgrad_finish_folded_groups", O.ptr(wg.acc), wg.nparts, wg.ntap, self.G, wg.cot * 16, wg.cit * 16, self.cout, self.cin, wg.w_sco, wg.w_sci,
                   self.apply_coef.data_ptr(), 3 * self.cpi, self.cpi, O.ptr(self.scratch.get(self.cls_sums_id)), self._pads[0], self._pads[1],
                   self._pads[2], O.ptr(w), O.ptr(grads[c + ".weight"]), O.ptr(grads[c + ".bias"]), O.ptr(bs), STATS_NREP, self.cpi, O.stream())
        if getattr(self, "_want_g", True):      # (else: the BatchNorm's own gradients below are all that is left of this layer's input side)
            self.dgrad.prep(w)
            self.dgrad.run(self.dz, self.g, self.batch)
        f.join()
        world = 1
        if SYNC["on"]:
            _allreduce(bs)
            world = SYNC["world"]
        L.call("sp_bn_bwd_finalize_groups", O.ptr(bs), STATS_NREP, float(self.count * world), O.ptr(params[p + ".weight"]),
               O.ptr(self.mean), O.ptr(self.invstd), self.cin, self.cpi, self.G, O.ptr(grads[p + ".weight"]),
               O.ptr(grads[p + ".bias"]), O.ptr(self.coef), 1.0 / world, O.stream())
        return self.g, self.coef

    def _backward_grouped(self, x, w, params, grads, wgrad=True):
        """groups > 1: one weight-gradient launch over all passes (operands: the materialised normalised input and dz -- the
        sum over the batch IS the sum over the passes), one data-gradient launch whose epilogue (or one reduction per group)
        yields the per-group BatchNorm-backward sums, one grouped finalize."""
        c = self.conv_prefix
        if self.raw_wgrad and wgrad and self.param_grads:
            return self._backward_grouped_raw(x, w, params, grads)
        src = self.xhat if self.materialize else x
        f = O.fork() if (O.overlap_level() == 2 and wgrad) else None
        if f is not None:
            f.__enter__()
        try:
            if not wgrad:
                pass
            elif self.kind == "conv":
                if self.fold_groups:        # the weight gradient's operand: x^ = s_g x + t_g, written here (beside the data gradient)
                    if self.xhat is None:
                        self.xhat = torch.empty_like(x)
                    O.bn_act_bwd(x, x, self.apply_coef, self.dtype, L.ACT_NONE, 0.0, self.xhat, None,
                                 group_vox=x.numel() // x.shape[-1] // self.G)
                    src = self.xhat
                self.wgrad.run(src, self.dz, self.batch, grads[c + ".weight"], None, None,
                               dbias_sums=self.dbias_sums, dbias_grad=grads[c + ".bias"], nbias=self.cout)
            else:
                self.wgrad.run(self.dz, src, self.batch, grads[c + ".weight"], None, None, None, None,
                               dbias_sums=self.dbias_sums, dbias_grad=grads[c + ".bias"], nbias=self.cout)
        finally:
            if f is not None:
                f.__exit__(None, None, None)
        try:
            if not (self.need_input_grad or self.bn_prefix is not None):
                return None, None
            self.dgrad.prep(w)
            if self.bn_prefix is None:
                self.dgrad.run(self.dz, self.g, self.batch)
                return self.g, None
            bs = self.scratch.get(self.bsums_id)
            fused = (all(s.tile["dma"] for s in self.dgrad.op.subs) and O.USE_DMA and not self.dgrad.uses_zm() and self.dgrad.fc is None) \
                or self.dgrad.zm_bn_bwd_ok() or self.dgrad.par_ok(self.dtype)
            if fused:
                self.dgrad.run(self.dz, self.g, self.batch, stats=bs, stats_nrep=STATS_NREP, stats_mode=1, aux=x, group_batch=self.gb)
            else:
                self.dgrad.run(self.dz, self.g, self.batch)
                per = self.cpi * 2 * STATS_NREP
                for gi in range(self.G):
                    sl = slice(gi * self.gb, (gi + 1) * self.gb)
                    O.bn_bwd_reduce(self.g[sl], x[sl], self.dtype, bs[gi * per:(gi + 1) * per])
            p = self.bn_prefix
            world = 1
            if SYNC["on"]:
                _allreduce(bs)
                world = SYNC["world"]
            L.call("sp_bn_bwd_finalize_groups", O.ptr(bs), STATS_NREP, float(self.count * world), O.ptr(params[p + ".weight"]),
                   O.ptr(self.mean), O.ptr(self.invstd), self.cin, self.cpi, self.G, O.ptr(grads[p + ".weight"]),
                   O.ptr(grads[p + ".bias"]), O.ptr(self.coef), 1.0 / world, O.stream())
            return self.g, self.coef
        finally:
            if f is not None:
                f.join()

    def _backward_input(self, x, w, params, grads):
        if not (self.need_input_grad or self.bn_prefix is not None):
            return None, None
        self.dgrad.prep(w)
        if self.bn_prefix is None:
            self.dgrad.run(self.dz, self.g, self.batch)
            return self.g, None
        p = self.bn_prefix
        bs = self.scratch.get(self.bsums_id)
        fused = (all(s.tile["dma"] for s in self.dgrad.op.subs) and O.USE_DMA and not self.dgrad.uses_zm() and self.dgrad.fc is None) \
            or self.dgrad.par_ok(self.dtype)
        if fused:      # (sum g, sum g*x) accumulated by the dgrad epilogue
            self.dgrad.run(self.dz, self.g, self.batch, stats=bs, stats_nrep=STATS_NREP, stats_mode=1, aux=x)
        else:
            self.dgrad.run(self.dz, self.g, self.batch)
            O.bn_bwd_reduce(self.g, x, self.dtype, bs)
        self._bn_bwd_finalize(bs, params, grads, STATS_NREP)
        return self.g, self.coef

    def _ensure_dz8(self):
        """dz8 = fp8(S * dz) unless the kernel that formed dz wrote it (dz8_out)"""
        if not self.dz8_ready:
            from . import f8 as F8
            F8.quantize(self.dz, self.dz8, F8.DZ_FMT, self.f8_grad_scale)
            self.dz8_ready = True

    def _run_dgrad(self, w):
        if self.f8_dgrad is not None:
            S = self.f8_grad_scale
            self._ensure_dz8()
            self.f8_dgrad.prep(w, out_scale=1.0 / S)
            self.f8_dgrad.run(self.dz8, self.g)
            return
        if getattr(self, "split_one", False):
            self.dgrad.prep(w)
            self.dgrad.run(self.dz, self.g_parts[0], self.batch, y2=self.g_parts[1], split_nt=self.split_g // 16)
        elif getattr(self, "dgrad_parts", None):
            for (runner, woff), g in zip(self.dgrad_parts, self.g_parts):
                runner.prep(w.view(-1)[woff:])
                runner.run(self.dz, g, self.batch)
        else:
            self.dgrad.prep(w)
            self.dgrad.run(self.dz, self.g, self.batch)

    def _bn_bwd_finalize(self, bs, params, grads, nrep):
        p = self.bn_prefix
        world = 1
        if SYNC["on"]:
            _allreduce(bs)
            world = SYNC["world"]      # sums are global now: every rank holds the full dgamma/dbeta -> scale by 1/world
        O.bn_bwd_finalize(bs, self.count * world, params[p + ".weight"], self.mean, self.invstd, self.cin, self.cpi,
                          grads[p + ".weight"], grads[p + ".bias"], self.coef, nrep=nrep, pscale=1.0 / world)


class FirstConvLayer(ConvLayer):
    """First layer of a network on the packed-K kernels of sp_first.hip: BatchNorm(2) -> Conv3d(2, 16 or 32, 3) -> act read
    straight from the NCDHW fp32 input (no channels-last copy, no padded channels), and a backward that needs no
    data-gradient convolution (the BatchNorm-backward sums come out of the weight gradient).  bf16 storage only."""

    @staticmethod
    def supported(cin, cout, k, stride, pad, dtype, bn):
        return bool(bn and dtype == L.SP_BF16 and stride == 1 and pad == 0 and L.load().sp_first_supported(cin, cout, k))

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        assert FirstConvLayer.supported(self.cin, self.cout, self.k, self.stride, self.pad, self.dtype, self.bn_prefix)
        assert self.cpo == self.cout
        self.wfrag = torch.zeros((self.cout // 16) * 3 * 64 * 8, dtype=torch.bfloat16, device=self.device)
        self.wfrag_lo = torch.zeros_like(self.wfrag) if self.hl else None
        self.bias_f = torch.zeros(self.cout, device=self.device)
        self.flops = 2.0 * self.batch * self.out_dims[0] * self.out_dims[1] * self.out_dims[2] * 27 * self.cin * self.cout
        self.store_y = True      # False (set by the engine, fp8 mode): only the e4m3 copy of the output is written and read

    def input_stats(self, images):
        """Batch statistics of the network input for the first BatchNorm (replaces bn_stats on a channels-last copy)."""
        B, Cc = images.shape[:2]
        L.call("sp_bn_stats_ncdhw_f32" if self.hl else "sp_bn_stats_ncdhw", O.ptr(images), B, Cc, images[0, 0].numel(), self.cpi,
               O.ptr(self.in_sums), STATS_NREP, O.stream())

    def forward(self, images, params, bufs, training, out_stats=None):
        assert images.dtype == torch.float32 and images.is_contiguous()
        bn = self._bn_fwd(params, bufs, training, fused=O.FUSE_BN_FINALIZE)
        c = self.conv_prefix
        y = self.alloc_out()
        st = O.stream()
        D, H, W = self.in_dims
        if bn is not None:      # the BatchNorm finalize inside the re-pack kernel
            L.call("sp_first_prep_bn", O.ptr(params[c + ".weight"]), O.ptr(params[c + ".bias"]), O.ptr(self.wfrag), O.ptr(self.wfrag_lo),
                   O.ptr(self.bias_f), self.cout, C.byref(bn), st)
        if self.hl:      # bf16 pairs: the fp32 input split into hi + lo inside the kernel, hi + lo weight fragments, y as a pair
            if bn is None:
                L.call("sp_first_prep_hl", O.ptr(params[c + ".weight"]), O.ptr(params[c + ".bias"]), O.ptr(self.scale), O.ptr(self.shift),
                       O.ptr(self.wfrag), O.ptr(self.wfrag_lo), O.ptr(self.bias_f), self.cout, st)
            with O._Timed("conv_igemm", self.flops, "%d->%d @%dx%dx%d first x3" % (self.cin, self.cout, D, H, W)):
                L.call("sp_first_conv_fwd_hl", O.ptr(images), self.batch, D, H, W, O.ptr(self.wfrag), O.ptr(self.wfrag_lo), O.ptr(self.bias_f),
                       self.act, self.act_param, O.ptr(y), O.ptr(self.y_lo), O.ptr(out_stats), STATS_NREP, self.cout, st)
            return y
        if bn is None:
            L.call("sp_first_prep_n", O.ptr(params[c + ".weight"]), O.ptr(params[c + ".bias"]), O.ptr(self.scale), O.ptr(self.shift),
                   O.ptr(self.wfrag), O.ptr(self.bias_f), self.cout, st)
        y8 = self.alloc_y8() if self.want_y8 else None      # (fp8 mode: the e4m3 operand of the second layer)
        assert self.store_y or y8 is not None
        with O._Timed("conv_igemm", self.flops, "%d->%d @%dx%dx%d first%s" % (self.cin, self.cout, D, H, W, "" if self.store_y else " (e4m3 only)")):
            L.call("sp_first_conv_fwd_n", O.ptr(images), self.batch, D, H, W, O.ptr(self.wfrag), O.ptr(self.bias_f), self.act,
                   self.act_param, O.ptr(y) if self.store_y else None, O.ptr(out_stats), STATS_NREP, self.cout, O.ptr(y8),
                   0 if y8 is None else y8[0].numel(), st)
        return y

    def y8_capable(self):
        return bool(self.FUSE_Q8)

    def _init_bwd(self):
        if self._bwd_ready:
            return
        self.dz = O.alloc_cl(self.batch, self.out_dims, self.cpo, self.dtype, self.device)
        vox = self.batch * self.out_dims[0] * self.out_dims[1] * self.out_dims[2]
        self.nparts = max(8, min(int(os.environ.get("SP_FIRST_WGRAD_BLOCKS", "1024")), vox // int(os.environ.get("SP_FIRST_WGRAD_MINVOX", "8192"))))
        self.partials = torch.empty(self.nparts * 27 * self.cout * 2, dtype=torch.float32, device=self.device)
        self.tapsrc = torch.arange(27, dtype=torch.int32, device=self.device)
        self.coef = torch.zeros(3, self.cpi, device=self.device)
        self._bwd_ready = True

    def backward(self, images, params, grads, g=None, coef=None):
        """g / coef given: the output gradient is formed inside the weight-gradient kernel (dz = (c0*g + c1*y + c2)*act'(y),
        the caller skips its sp_bn_act_bwd); else self.dz / self.dbias_sums were filled by the caller."""
        c = self.conv_prefix
        st = O.stream()
        D, H, W = self.in_dims
        bs = self.scratch.get(self.bsums_id)
        with O._Timed("conv_wgrad", self.flops, "%d->%d @%dx%dx%d first" % (self.cin, self.cout, D, H, W)):
            if g is not None and not self.store_y:      # y as its e4m3 copy (the 16-bit tensor was not written)
                L.call("sp_first_wgrad_fused_y8", O.ptr(images), O.ptr(g), O.ptr(self.y8), self.y8[0].numel(), O.ptr(coef), self.act,
                       self.act_param, self.batch, D, H, W, O.ptr(self.partials), self.nparts, O.ptr(self.dbias_sums), self.cout, st)
            elif g is not None:
                L.call("sp_first_wgrad_fused_n", O.ptr(images), O.ptr(g), O.ptr(self.y), O.ptr(coef), self.act, self.act_param,
                       self.batch, D, H, W, O.ptr(self.partials), self.nparts, O.ptr(self.dbias_sums), self.cout, st)
            else:
                assert self.store_y, "the un-fused first-layer backward reads the 16-bit output"
                L.call("sp_first_wgrad_n", O.ptr(images), O.ptr(self.dz), self.batch, D, H, W, O.ptr(self.partials), self.nparts,
                       self.cout, st)
        L.call("sp_wgrad_finish_folded", O.ptr(self.partials), self.nparts, O.ptr(self.tapsrc), 27, self.cout, 2, self.cout,
               self.cin, self.cin * 27, 27, O.ptr(self.scale), O.ptr(self.shift), O.ptr(self.dbias_sums),
               O.ptr(grads[c + ".weight"]), O.ptr(grads[c + ".bias"]), O.ptr(params[c + ".weight"]), O.ptr(bs), STATS_NREP,
               self.cpi, self.cpo, st)
        self._bn_bwd_finalize(bs, params, grads, STATS_NREP)
        return None, None
