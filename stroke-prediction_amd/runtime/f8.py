"""fp8 (OCP e4m3 / e5m2) execution of the stride-1 3x3x3 convolutions: drivers of ``sp_conv3d_zm8`` / ``sp_conv_prep_f8`` /
``sp_quantize_f8`` (csrc/sp_conv_zm8.hip; BASELINE.json configs[4]).

Precision recipe (``Unet3D(dtype="fp8")``): the MFMA operands of the forward and the data-gradient convolution of every
3x3x3 layer with 32..128 input channels, and of the weight gradient of every 3x3x3 layer but the first, are fp8 -- activations
e4m3 (plane-major copies written by the producing convolution's epilogue, or by the pooling / concatenation kernel, or by one
quantisation pass), weights e4m3 with a power-of-two scale per output channel, output gradients e5m2 scaled by a power of two --
with fp32 accumulation; BatchNorm statistics, bias, activations, Dice and Adam stay fp32, the stored activations every other
kernel reads stay bf16.  The first layer (2 input channels), the classify head and the forward / data gradient of the layers with
12 or 24 input planes run on the bf16 kernels.

``Unet3D(dtype="fp8b")`` keeps the bf16 forward and runs only the backward (data and weight gradients) on these kernels: the
direction the "fp8" mode's gradients lose against the f32 mode is the e4m3 forward's alone (tools/probes/f8_cos_knobs.sh).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import lib as L
from . import ops as O
from . import plan as P

F8_MIN_PLANES = int(os.environ.get("SP_F8_MIN_PLANES", "128"))      # (column, plane) pairs below which the march is all prologue (measured: 128 -> 128 @46^3 169 -> 110 us, 128 -> 256 @28^3 117 -> 69 us against the bf16 tiled kernel)
E4M3, E5M2 = 0, 1
FWD = bool(int(os.environ.get("SP_F8_FWD", "1")))           # forward convolutions on the fp8 kernel (0: bf16 forward, fp8 backward only)
DGRAD = bool(int(os.environ.get("SP_F8_DGRAD", "1")))       # data-gradient convolutions on the fp8 kernel too (0: forward only)
WGRAD = bool(int(os.environ.get("SP_F8_WGRAD", "1")))       # weight gradients of the fp8 layers on fp8 operands as well (0: from the bf16 tensors)
FUSE_SLICES = bool(int(os.environ.get("SP_F8_FUSE_SLICES", "1")))  # the output-channel slices of an op in one launch (0: one launch each)
FUSE_SLICES_MIN = 4      # ... when every workgroup of that launch gets at least this many (column, plane) pairs
SPLIT = bool(int(os.environ.get("SP_F8_SPLIT", "1")))       # ops with 12 / 16 / 24 input planes as groups of 6 / 8 planes + one finish pass
WGRAD_ONLY = bool(int(os.environ.get("SP_F8_WGRAD_ONLY", "1")))  # ... and those of the layers whose forward has no fp8 instance (their x8 is made for it)
DZ_FMT = E5M2 if os.environ.get("SP_F8_DZ", "e5m2") == "e5m2" else E4M3      # storage format of the quantised output gradients


_F8_ITEM = np.dtype([("w", "<u8"), ("sCo", "<i8"), ("sCi", "<i8"), ("Cout", "<i4"), ("Cin", "<i4"), ("kmap", "<u8"), ("nsteps", "<i4"), ("NT", "<i4"),
                     ("wfrag", "<u8"), ("fold_scale", "<u8"), ("fold_shift", "<u8"), ("bias", "<u8"), ("bias_out", "<u8"), ("winv", "<u8"),
                     ("ntaps", "<i4"), ("out_scale", "<f4")])      # sp_f8_prep_item
_tables = {}


def prep_many(jobs):
    """e4m3 weight fragments of several ConvRunnerF8 (all their output-channel slices) in ONE launch.
    jobs: [(runner, w, b, fold_scale, fold_shift, out_scale)]; jobs whose fragments are current are skipped."""
    todo = []
    for job in jobs:
        r, w, b, fs, fsh, osc = job[:6]
        key = None if fs is not None else (w.data_ptr(), w._version, O.PARAM_EPOCH[0], float(osc))
        if key is None or r._key != key:
            todo.append((r, w, b, fs, fsh, osc, key))
    if not todo:
        return
    # the cached device table holds raw addresses and geometry only: key it on EVERYTHING it contains (as ops.prep_batch does), so
    # an entry can only be replayed for runners that own exactly those buffers -- after Unet3D._engine evicted its engines a new
    # runner may get the old w / bias / winv / wfrag addresses back while its kmap table lands elsewhere
    tkey = tuple((w.data_ptr(), 0 if b is None else b.data_ptr(), O.ptr(fs) or 0, O.ptr(fsh) or 0, float(osc), r.bias.data_ptr(), r.winv.data_ptr(), r.ci0,
                  r.op.w_sco, r.op.w_sci, r.op.cin) +
                 tuple((s["wfrag"].data_ptr(), s["kmap_d"].data_ptr(), s["c0"], s["cn"], s["nsteps"], s["NT"]) for s in r.slices)
                 for r, w, b, fs, fsh, osc, _ in todo)
    tab = _tables.get(tkey)
    if tab is None and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("fp8 weight re-pack: this set of layers has not been re-packed together in an eager step, and its address "
                           "table is a host-to-device copy that a graph capture refuses -- run one more eager step before "
                           "capturing (Learner.GRAPH_WARMUP)")
    if tab is None:
        if len(_tables) > 256:
            _tables.clear()
        items = []
        for r, w, b, fs, fsh, osc, _ in todo:
            op = r.op
            assert w.dtype == torch.float32 and w.is_contiguous()
            ntaps = 27                    # (3 x 3 x 3 only)
            want_bias = b is not None or fsh is not None
            ci0 = r.ci0                   # the runner covers input channels [ci0, ci0 + op.cin) of the weight and of the fold tables
            for s in r.slices:
                c0 = s["c0"]
                items.append((w.data_ptr() + 4 * (c0 * op.w_sco + ci0 * op.w_sci), op.w_sco, op.w_sci, s["cn"], op.cin, s["kmap_d"].data_ptr(), s["nsteps"], s["NT"],
                              s["wfrag"].data_ptr(), (O.ptr(fs) + 4 * ci0) if fs is not None else 0, (O.ptr(fsh) + 4 * ci0) if fsh is not None else 0,
                              0 if b is None else b.data_ptr() + 4 * c0,
                              (r.bias.data_ptr() + 4 * c0) if want_bias else 0, r.winv.data_ptr() + 4 * c0, ntaps, float(osc)))
        arr = np.array(items, dtype=_F8_ITEM)
        dev = torch.from_numpy(arr.view(np.uint8).copy()).to(todo[0][1].device)
        tab = _tables[tkey] = (dev, len(items), max(int(it[7]) * 16 for it in items))
    dev, n, rows = tab
    L.call("sp_conv_prep_f8_batch", dev.data_ptr(), n, rows, O.stream())
    for r, w, b, fs, fsh, osc, key in todo:
        r._key = key
        r.has_bias = b is not None or fsh is not None


def alloc_f8(batch, dims, cp, device):
    """plane-major fp8 tensor [cp/16][B][D][H][W][16 bytes]"""
    assert cp % 16 == 0
    return torch.empty((cp // 16, batch) + tuple(dims) + (16,), dtype=torch.uint8, device=device)


def quantize(src, dst, fmt=E4M3, scale=1.0, src_planar=False):
    """dst (plane-major fp8, see alloc_f8) = fp8(scale * src); src bf16 channels-last (B, D, H, W, CP) or the same shape stored
    plane-major (the concat buffers)."""
    B, D, H, W, CP = src.shape
    nvox = B * D * H * W
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.uint8 and tuple(dst.shape) == (CP // 16, B, D, H, W, 16), \
        (tuple(src.shape), tuple(dst.shape))
    with O._Timed("quantize_f8", 0.0, "%s %dx%dx%d x%d" % ("e5m2" if fmt else "e4m3", D, H, W, CP)):
        L.call("sp_quantize_f8", O.ptr(src), CP, nvox * 16 if src_planar else 0, O.ptr(dst), nvox * 16, nvox, fmt, float(scale),
               O.stream())


class ConvRunnerF8:
    """One stride-1 3x3x3 op (``plan.ConvOp``: a forward convolution or a data gradient) on the fp8 z-marching kernel, one
    launch per slice of <= 32 output channels."""

    @staticmethod
    def applicable(op, batch):
        sl = P.zm8_slices(op)
        if sl is None or op.cin > op.cpi:
            return False
        z = P.zm8_plan(sl[0][2])
        sub = op.subs[0]
        cols = -(-sub.out_dims[1] // z["TH"]) * -(-sub.out_dims[2] // 16)
        return batch * cols * sub.out_dims[0] >= F8_MIN_PLANES

    def __init__(self, op, device, batch, bin_fmt=E4M3, ci0=0):
        assert ConvRunnerF8.applicable(op, batch)
        self.op, self.device, self.batch, self.bin = op, device, batch, bin_fmt
        self.ci0 = ci0                # first input channel (of the weight tensor and the fold tables) this runner's op starts at
        self.slices = []
        sl = [(c0, cn, P.zm8_plan(sub_op)) for c0, cn, sub_op in P.zm8_slices(op)]
        # equal slices (same tile count, consecutive channel ranges): their fragments share one buffer and ONE launch runs them
        # all (sp_conv_args.nslices: the slices' pieces of an output row then leave the L2 as whole lines)
        nt0, n0 = sl[0][2]["NT"], sl[0][2]["nsteps"]
        uniform = bool(FUSE_SLICES and len(sl) >= 2 and all(z["NT"] == nt0 and z["nsteps"] == n0 and c0 == i * nt0 * 16 for i, (c0, _, z) in enumerate(sl)))
        # slices per launch: a divisor m <= 16 of the slice count; the 32 workgroups of an XCD form floor(32 / m) teams of m -- the
        # largest m that leaves at most 10 % of them idle (24 slices: three launches of 8), else the best filling one
        self.fuse_m = 0
        if uniform:
            cands = [m for m in range(2, min(16, len(sl)) + 1) if len(sl) % m == 0]
            good = [m for m in cands if (32 // m) * m >= 29]
            if cands:
                self.fuse_m = max(good) if good else max(cands, key=lambda m: ((32 // m) * m, m))
        self.wstride = n0 * nt0 * 2048
        pool = torch.empty(len(sl) * self.wstride, dtype=torch.uint8, device=device) if self.fuse_m else None
        for i, (c0, cn, z) in enumerate(sl):
            wf = pool[i * self.wstride:(i + 1) * self.wstride] if self.fuse_m else torch.empty(z["nsteps"] * z["NT"] * 2048, dtype=torch.uint8, device=device)
            self.slices.append(dict(z, c0=c0, cn=cn, ktab_d=O._dev_i32(z["ktab"], device), kmap_d=O._dev_i32(z["kmap"], device), wfrag=wf))
        if self.fuse_m:     # enough (column, plane) pairs for every team of a launch
            z = self.slices[0]
            sub = op.subs[0]
            pairs = batch * -(-sub.out_dims[1] // z["TH"]) * -(-sub.out_dims[2] // 16) * sub.out_dims[0]
            if pairs < FUSE_SLICES_MIN * 8 * (32 // self.fuse_m):
                self.fuse_m = 0
        self.fused = self.fuse_m > 0
        cpad = -(-op.cout // 16) * 16
        self.bias = torch.zeros(cpad, dtype=torch.float32, device=device)
        self.winv = torch.ones(cpad, dtype=torch.float32, device=device)
        self.has_bias = False
        self._key = None

    def prep_jobs(self, w, out_scale):
        """prep_many jobs of an un-folded re-pack (data gradients: the engine batches those of all layers into one launch)"""
        return [(self, w, None, None, None, out_scale)]

    def prep(self, w, b=None, fold_scale=None, fold_shift=None, out_scale=1.0):
        """e4m3 fragments of w (x fold_scale per input channel), folded bias, per-channel dequantisation multipliers
        (x out_scale: the reciprocal of the scale the B operand was quantised with)."""
        prep_many([(self, w, b, fold_scale, fold_shift, out_scale)])      # one launch for all slices

    def run(self, x8, y, act=L.ACT_NONE, act_param=0.0, stats=None, stats_nrep=1, y8=None, y8_scale=1.0, store=True):
        """y bf16: the finished output; y fp32: this op's partial sums (one input-channel group of a larger convolution --
        no bias, activation, statistics; ConvRunnerF8Split adds the groups up).  store=False (with y8): the 16-bit output is not
        written (y only gives the shape) -- every reader takes the e4m3 copy"""
        op, batch = self.op, self.batch
        assert store or (y8 is not None and y.dtype == torch.bfloat16)
        sub = op.subs[0]
        partial = y.dtype == torch.float32
        assert x8.dtype == torch.uint8 and tuple(x8.shape) == (op.cpi // 16, batch) + tuple(op.in_dims) + (16,), \
            (tuple(x8.shape), op.cpi, op.in_dims)
        assert (partial or y.dtype == torch.bfloat16) and tuple(y.shape[:4]) == (batch,) + tuple(op.y_dims) and y.shape[4] >= op.cpo
        assert not partial or (act == L.ACT_NONE and stats is None and y8 is None)
        if y8 is not None:
            assert self.bin == E4M3 and y8.dtype == torch.uint8 and tuple(y8.shape) == (y.shape[4] // 16, batch) + tuple(op.y_dims) + (16,)
        a = L.ConvArgs()
        a.x = O.ptr(x8)
        a.dtype_in, a.dtype_out = L.SP_BF16, (L.SP_F32 if partial else L.SP_BF16)
        a.B = batch
        a.Di, a.Hi, a.Wi = op.in_dims
        a.CPi = op.cpi
        a.x_plane = batch * int(np.prod(op.in_dims)) * 16
        a.YD, a.YH, a.YW = op.y_dims
        a.CPo = y.shape[4]
        a.Do, a.Ho, a.Wo = sub.out_dims
        a.osD = a.osH = a.osW = 1
        a.sD = a.sH = a.sW = 1
        a.o0D, a.o0H, a.o0W = sub.o0
        a.act, a.act_param = act, act_param
        a.stats_nrep = stats_nrep
        a.f8_bin = self.bin
        a.y8_scale = float(y8_scale)
        plane8 = batch * int(np.prod(op.y_dims)) * 16
        a.y8_plane = plane8
        st = O.stream()
        a.nslices, a.slice_wfrag_stride = (self.fuse_m, self.wstride) if self.fused else (0, 0)
        nl = len(self.slices) // self.fuse_m if self.fused else 0
        for s in (self.slices[::self.fuse_m] if self.fused else self.slices):
            c0 = s["c0"]
            a.y = (y.data_ptr() + y.element_size() * c0) if store else None
            a.bias = (self.bias.data_ptr() + 4 * c0) if (self.has_bias and not partial) else None
            a.f8_wscale = self.winv.data_ptr() + 4 * c0
            a.stats = None if stats is None else stats.data_ptr() + 16 * c0
            a.y8 = None if y8 is None else y8.data_ptr() + (c0 // 16) * plane8
            a.wfrag_hi, a.ktab = O.ptr(s["wfrag"]), O.ptr(s["ktab_d"])
            a.MT, a.NT, a.NTtot = s["MT"], s["NT"], s["NT"]
            a.Cout = s["NT"] * 16
            with O._Timed("conv_igemm", op.flops(batch) * (1.0 / nl if self.fused else s["cn"] / op.cout),
                          "%d->%d @%s zm8 %s%s%s" % (op.cin, op.cout, "x".join(map(str, op.in_dims)), "e5m2" if self.bin else "e4m3",
                                                     (" %d slices in one" % self.fuse_m) if self.fused else (" slices" if len(self.slices) > 1 else ""),
                                                     " +stats" if stats is not None else "")):
                L.call("sp_conv3d_zm8", C.byref(a), O.ptr(O.zero_page(self.device)), st)


class ConvRunnerF8Split:
    """A stride-1 3x3x3 op with more input planes than an fp8 instance holds (12, 16, 24: the layers behind the concatenations and
    the 256-channel bottleneck of the 4-scale network): one ConvRunnerF8 per group of 6 or 8 input planes writes fp32 partial sums,
    ``sp_conv_partial_finish`` adds the groups (and their folded biases), applies the activation and takes the statistics.
    Same interface as ConvRunnerF8 (the e4m3 copy of the output comes from the finish pass)."""

    @staticmethod
    def groups(op):
        P_ = op.cpi // 16
        if op.cin != op.cpi or op.cpi % 16:
            return None
        for gp in (6, 8):
            if P_ > 8 and P_ % gp == 0:
                return gp, P_ // gp
        return None

    @staticmethod
    def _sub_op(op, gp):
        import dataclasses
        return dataclasses.replace(op, cin=gp * 16, cpi=gp * 16)

    @staticmethod
    def applicable(op, batch):
        g = ConvRunnerF8Split.groups(op)
        return bool(SPLIT and g is not None and ConvRunnerF8.applicable(ConvRunnerF8Split._sub_op(op, g[0]), batch))

    def __init__(self, op, device, batch, bin_fmt=E4M3):
        assert ConvRunnerF8Split.applicable(op, batch)
        self.op, self.device, self.batch, self.bin = op, device, batch, bin_fmt
        self.gp, self.G = ConvRunnerF8Split.groups(op)
        sub = ConvRunnerF8Split._sub_op(op, self.gp)
        self.runners = [ConvRunnerF8(sub, device, batch, bin_fmt, ci0=g * self.gp * 16) for g in range(self.G)]
        self.cpad = self.runners[0].bias.numel()
        self.bias_all = torch.zeros(self.G, self.cpad, dtype=torch.float32, device=device)
        for g, r in enumerate(self.runners):
            r.bias = self.bias_all[g]
        self.partial = None
        self.has_bias = False

    def prep(self, w, b=None, fold_scale=None, fold_shift=None, out_scale=1.0):
        prep_many([(r, w, b if g == 0 else None, fold_scale, fold_shift, out_scale) for g, r in enumerate(self.runners)])
        self.has_bias = b is not None or fold_shift is not None

    def prep_jobs(self, w, out_scale):
        return [(r, w, None, None, None, out_scale) for r in self.runners]

    def run(self, x8, y, act=L.ACT_NONE, act_param=0.0, stats=None, stats_nrep=1, y8=None, y8_scale=1.0, store=True):
        op, batch = self.op, self.batch
        assert store or y8 is not None
        assert y.dtype == torch.bfloat16 and y.shape[4] == self.cpad == op.cpo, (tuple(y.shape), self.cpad, op.cpo)
        assert y8 is None or (self.bin == E4M3 and y8_scale == 1.0 and tuple(y8.shape) == (self.cpad // 16, batch) + tuple(op.y_dims) + (16,))
        assert tuple(x8.shape) == (op.cpi // 16, batch) + tuple(op.in_dims) + (16,), (tuple(x8.shape), op.cpi)
        nvox = batch * int(np.prod(op.y_dims))
        if self.partial is None:
            self.partial = torch.empty((self.G, batch) + tuple(op.y_dims) + (self.cpad,), dtype=torch.float32, device=self.device)
        for g, r in enumerate(self.runners):
            r.run(x8[g * self.gp:(g + 1) * self.gp], self.partial[g])
        with O._Timed("conv_partial_finish", 0.0, "%d->%d @%s x%d groups" % (op.cin, op.cout, "x".join(map(str, op.in_dims)), self.G)):
            L.call("sp_conv_partial_finish", O.ptr(self.partial), self.G, nvox, self.cpad, O.ptr(self.bias_all) if self.has_bias else None,
                   self.cpad, act, act_param, O.ptr(y) if store else None, O.ptr(stats), stats_nrep, O.ptr(y8), nvox * 16, O.stream())


class WgradRunnerF8:
    """Weight gradient of a stride-1 un-padded 3x3x3 convolution on the fp8 operands the forward (x8: e4m3 copy of the raw
    input) and the data gradient (dz8: e5m2 copy of S * dz) already hold (``sp_conv3d_wgrad_f8``).  Rides on the layer's bf16
    ``ops.WgradRunner`` for everything but the kernel: the finish step is the folded one (BatchNorm of the input applied to
    the accumulator, BatchNorm-backward sums out of it), called with acc_scale = 1 / S."""

    BLOCK_UNITS = int(os.environ.get("SP_F8_WGRAD_UNITS", "8"))      # (column, plane) pairs a persistent workgroup wants at least

    @staticmethod
    def applicable(wg):
        return bool(WGRAD and DZ_FMT == E5M2 and wg.dma and wg.unpadded and wg.cot % 2 == 0 and wg.cit % 2 == 0)

    def __init__(self, wg):
        assert WgradRunnerF8.applicable(wg)
        self.wg = wg
        self.acc = None
        self.acc_batch = None
        self.args = L.WgradF8Args()

    def _alloc(self, batch):
        wg, a = self.wg, self.args
        w = wg.args
        yz = (wg.cot // 2) * (wg.cit // 2)
        units = batch * -(-w.Ho // 4) * -(-w.Wo // 32) * w.Do
        nb = max(8, min(512 // yz, units // self.BLOCK_UNITS)) // 8 * 8
        a.nblocks = self.nparts = int(os.environ.get("SP_F8_WGRAD_BLOCKS", nb))
        self.acc = torch.empty(self.nparts * wg.ntap * wg.cot * 16 * wg.cit * 16, dtype=torch.float32, device=wg.device)
        self.acc_batch = batch

    def run(self, x8, dz8, batch, grad_scale, dw, in_scale, in_shift, dbias_sums, dbias_grad, bn_w=None, bn_sums=None, bn_nrep=1):
        """launches the kernel; returns the finish step (a callable, as ``WgradRunner.run(defer_finish=True)``)"""
        wg, a = self.wg, self.args
        w = wg.args
        assert x8.dtype == torch.uint8 and tuple(x8.shape) == (wg.cit, batch, w.Di, w.Hi, w.Wi, 16), (tuple(x8.shape), wg.cit)
        assert dz8.dtype == torch.uint8 and tuple(dz8.shape) == (wg.cot, batch, w.Do, w.Ho, w.Wo, 16), (tuple(dz8.shape), wg.cot)
        assert in_scale is not None and dbias_sums is not None
        if self.acc is None or self.acc_batch != batch:
            self._alloc(batch)
        a.x, a.dz, a.dw_acc = O.ptr(x8), O.ptr(dz8), O.ptr(self.acc)
        a.B = batch
        a.Di, a.Hi, a.Wi, a.Do, a.Ho, a.Wo = w.Di, w.Hi, w.Wi, w.Do, w.Ho, w.Wo
        a.CoT, a.CiT = wg.cot, wg.cit
        a.x_plane = batch * w.Di * w.Hi * w.Wi * 16
        a.dz_plane = batch * w.Do * w.Ho * w.Wo * 16
        with O._Timed("conv_wgrad", 2 * batch * w.Do * w.Ho * w.Wo * wg.ntap * wg.cin * wg.cout,
                      "%d->%d @%dx%dx%d f8" % (wg.cin, wg.cout, w.Di, w.Hi, w.Wi)):
            L.call("sp_conv3d_wgrad_f8", C.byref(a), O.stream())

        def finish():
            L.call("sp_wgrad_finish_folded_scaled", O.ptr(self.acc), self.nparts, O.ptr(wg.tapsrc), wg.ntap, wg.cot * 16, wg.cit * 16,
                   wg.cout, wg.cin, wg.w_sco, wg.w_sci, O.ptr(in_scale), O.ptr(in_shift), O.ptr(dbias_sums), O.ptr(dw),
                   O.ptr(dbias_grad), O.ptr(bn_w), O.ptr(bn_sums), bn_nrep, 0, O._dbias_stride(dbias_sums), 1.0 / float(grad_scale),
                   O.stream())
        return finish


def grad_scale_for(n_out_voxels):
    """power of two the output gradients are multiplied with before they are rounded to e5m2: a mean-type loss over
    n_out_voxels has per-voxel gradients ~ 1/n; 64 n puts them in the middle of e5m2's 30 binades."""
    return float(2.0 ** math.ceil(math.log2(64.0 * max(1, n_out_voxels))))
