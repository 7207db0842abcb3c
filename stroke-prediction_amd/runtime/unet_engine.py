"""Forward / backward of the 3-D U-Net (``Unet3D.forward`` Unet3D.py:56-79; 4-scale topology Unet3D.py:95-146) on HIP kernels.

Data flow (channels-last tensors, BatchNorm folded into the consuming convolution's load):

  x0 -b1c1-> y11 -b1c2-> y12 -pool-> p1 -b2c1-> y21 -b2c2-> y22 -pool-> p2 -b3c1-> y31 -b3c2-> y32
  cat4 = [up(y32) | crop(y22)] -b4c1-> y41 -b4c2-> y42 ; cat5 = [up(y42) | crop(y12)] -b5c1-> y51 -b5c2-> y52
  y52 -1x1,lrelu-> h -1x1,sigmoid-> seg
(three scales; the engine is written for S scales: blocks 1..S down, S+1..2S-1 up -- S = 4 is the LargeUnet3D topology)

Batch statistics of every BatchNorm input are produced by the kernel that writes that tensor
(conv / pool / upsample / crop epilogues); only the network input needs a stand-alone pass.
"""
import math
import os

import torch

from . import lib as L
from . import ops as O
from .layers import ConvLayer, FirstConvLayer, Scratch

F8_Y2_E4M3 = bool(int(os.environ.get("SP_F8_Y2_E4M3", "1")))      # ... and the down blocks' second activations (pooling / skip crop / pool backward read the copy)
F8_Y1_E4M3 = bool(int(os.environ.get("SP_F8_Y1_E4M3", "1")))      # fp8 mode: the first layer's output lives as its e4m3 copy only (0: A/B)

LEAKY = 0.01


def unet_out_dims(dims, scales=3):
    """Appendix B of SURVEY.md: per axis, every block loses 4 voxels (two valid 3x3x3 convolutions), pooling halves
    (floor), upsampling doubles.  scales = 3: 128 -> 88; scales = 4: 256 -> 164.  Raises for inputs too small."""
    out = []
    for n in dims:
        m = n
        for _ in range(scales - 1):
            m = (m - 4) // 2
        m -= 4
        if m < 1:
            raise ValueError("Unet3D: spatial size %s is too small for %d scales of valid 3x3x3 convolutions" % (tuple(dims), scales))
        for _ in range(scales - 1):
            m = 2 * m - 4
        if m < 1:
            raise ValueError("Unet3D: spatial size %s is too small for %d scales of valid 3x3x3 convolutions" % (tuple(dims), scales))
        out.append(m)
    return tuple(out)


class UnetEngine:
    """Bound to one (batch, spatial size, dtype); the model keeps a small cache of engines.

    channels = [n_in, b_1 .. b_{2S-1}, b_C, n_classes] for S scales: S = 3 is ``Unet3D`` (Unet3D.py:30-84, 8 numbers),
    S = 4 the topology of ``LargeUnet3D`` (Unet3D.py:87-146, 10 numbers).  Blocks 1..S go down (pool between them),
    blocks S+1..2S-1 go up; up block u concatenates the upsampled output of block u-1 with the centre crop of down
    block 2S-u."""

    LOSS_SCALE_GROWTH_STEPS = 200      # f16 build: clean steps after which a halved loss scale doubles again

    def __init__(self, channels, batch, dims, dtype, device, f8=False, variant="", hl=False, f8_fwd=True):
        """f8: the "fp8" precision mode -- storage stays bf16 (dtype), the 3x3x3 layers the fp8 kernel has an instance for
        run their forward and data-gradient MFMAs on e4m3 / e5m2 operands (runtime/f8.py).  f8_fwd=False ("fp8b"): only the
        backward does -- every forward convolution is the bf16 one (its epilogue, or the pooling / concatenation kernel, writes
        the e4m3 copy the weight gradient reads), data and weight gradients run on the fp8 kernels.
        hl: the "bf16x3" precision mode -- every activation of the FORWARD pass is a bf16 pair (hi + lo tensors, ~17 bits), the
        forward convolutions run three MFMAs per product (hi*hi + hi*lo + lo*hi with hi / lo weight fragments), pooling /
        concatenation / head work on the pair values; the BACKWARD pass is the bf16 one on the hi tensors, which are exactly the
        tensors the bf16 mode stores."""
        O.require_gpu()
        L.load()
        assert not f8 or dtype == L.SP_BF16
        assert not hl or (dtype == L.SP_BF16 and not f8)
        self.hl = bool(hl)
        self.variant = variant       # build of the library this engine's tensors belong to (lib.use): "" = bf16, "f16" = IEEE half
        assert L.current_variant() == variant, "construct and run an engine inside lib.use(engine.variant)"
        assert len(channels) >= 8 and len(channels) % 2 == 0, "channels: n_in, 2S-1 block widths, head width, classes"
        S = self.scales = (len(channels) - 2) // 2
        n_in, bch, bc, ncls = channels[0], list(channels[1:2 * S]), channels[-2], channels[-1]
        assert ncls <= 8, "the classify head supports up to 8 classes"
        self.channels, self.batch, self.dims, self.dtype, self.device = list(channels), batch, tuple(dims), dtype, device
        try:
            unet_out_dims(dims, S)
        except ValueError:
            raise ValueError("Unet3D: spatial size %s is too small for %s scales of valid 3x3x3 "
                             "convolutions (minimum %d per axis)" % (tuple(dims), {3: "three", 4: "four"}.get(S, S),
                                                                    {3: 44, 4: 92}.get(S, 0)))
        self.scratch = sc = Scratch(device)
        mk = lambda name, ci, co, d, bn=True, k=3, act=L.ACT_LEAKY, ap=LEAKY, out_dtype=None, blk=None, idx=None, \
            need_g=True, cpi=None, split=None, pooled=False: ConvLayer(name, "conv", ci, co, k, 1, 0, d, batch, dtype, device, sc, pooled=pooled,
                                             bn_prefix=("%s.bn_conv_relu_2x.%d" % (blk, idx)) if bn else None,
                                             conv_prefix=("%s.bn_conv_relu_2x.%d" % (blk, idx + 1)) if bn else name,
                                             act=act, act_param=ap, out_dtype=out_dtype, need_input_grad=need_g, cpi=cpi,
                                             split_g=split, hl=(hl and bn))
        sub = lambda d, k: tuple(x - k for x in d)
        half = lambda d: tuple(x // 2 for x in d)
        dbl = lambda d: tuple(2 * x for x in d)
        # conv[i] = (first, second) convolution of block i (1-based); dims_in[i] = input dims of block i
        self.conv, self.dims_in = {}, {}
        d = self.dims
        b1 = bch[0]
        # the network input gets a 16-channel pitch (not 8): every 3x3x3 layer then meets the 16-channel plane
        # granularity of the DMA weight-gradient kernel
        self.first_packed = FirstConvLayer.supported(n_in, b1, 3, 1, 0, dtype, True) and not os.environ.get("SP_GENERIC_FIRST")
        for i in range(1, S + 1):                       # ---- down path
            ci = n_in if i == 1 else bch[i - 2]
            co = bch[i - 1]
            self.dims_in[i] = d
            if i == 1 and self.first_packed:   # two-channel input: packed-K kernels reading the NCDHW fp32 input directly
                c1 = FirstConvLayer("b1c1", "conv", n_in, b1, 3, 1, 0, d, batch, dtype, device, sc,
                                    bn_prefix="block1.bn_conv_relu_2x.0", conv_prefix="block1.bn_conv_relu_2x.1",
                                    act=L.ACT_LEAKY, act_param=LEAKY, need_input_grad=False, cpi=O.cpad(n_in, 16), hl=hl)
            else:
                c1 = mk("b%dc1" % i, ci, co, d, blk="block%d" % i, idx=0, need_g=(i > 1), cpi=O.cpad(ci, 16) if i == 1 else None)
            c2 = mk("b%dc2" % i, co, co, sub(d, 2), blk="block%d" % i, idx=3, pooled=(i < S))
            self.conv[i] = (c1, c2)
            d = sub(d, 4)
            if i < S:
                d = half(d)
        # gradient of a concatenated input as two dense tensors: from ONE data-gradient launch where that instance exists (the layer
        # decides: ConvLayer.split_one); two launches cost more (+45 us) than the dense reads save -> SP_SPLIT_G opt-in only
        split_ok = lambda cu, cs: cu if (256 % (cu // 8) == 0 and cu % 16 == 0 and cs % 16 == 0 and not hl_or_f8) else None
        hl_or_f8 = bool(f8)
        for u in range(S + 1, 2 * S):                   # ---- up path: d = output dims of block u-1
            cu, cs, co = bch[u - 2], bch[2 * S - u - 1], bch[u - 1]
            assert cu % 8 == 0, "up-path channel counts must be multiples of 8"
            d = dbl(d)
            self.dims_in[u] = d
            c1 = mk("b%dc1" % u, cu + cs, co, d, blk="block%d" % u, idx=0, split=split_ok(cu, O.cpad(cs)))
            c2 = mk("b%dc2" % u, co, co, sub(d, 2), blk="block%d" % u, idx=3)
            self.conv[u] = (c1, c2)
            d = sub(d, 4)
        self.out_dims = d
        blast = bch[-1]
        self.h0 = mk("classify.0", blast, bc, d, bn=False, k=1)
        self.h2 = mk("classify.2", bc, ncls, d, bn=False, k=1, act=L.ACT_SIGMOID, ap=0.0, out_dtype=L.SP_F32)
        # fused pointwise head where a kernel exists for (C, CH, NC); the two generic 1x1 layers otherwise
        self.fused_head = bool(L.load().sp_head_supported_dtype(blast, bc, ncls, dtype)) and blast % 8 == 0 and not os.environ.get("SP_GENERIC_HEAD")
        if hl and not (self.first_packed and self.fused_head and blast == 16):
            raise NotImplementedError("Unet3D(dtype='bf16x3'): the bf16-pair forward needs the packed first layer (2 input channels, "
                                      "16 or 32 outputs) and the fused classify head (16 -> 16 | 32 -> <= 2 classes)")
        self.layers = [c for i in range(1, 2 * S) for c in self.conv[i]] + [self.h0, self.h2]
        for l in self.layers:
            l.reserve_bwd_scratch()
        sc.finalize()
        dt = dtype
        c11 = self.conv[1][0]
        self.x0 = None if self.first_packed else O.alloc_cl(batch, self.dims, c11.cpi, dt, device)
        self.pooled_lo, self.cat_lo = {}, {}
        if hl:      # pairs: one allocation each, hi half first (the hi halves are what the backward reads)
            pair = lambda d, cp: torch.empty((2, batch) + tuple(d) + (cp,), dtype=O.TORCH_DT[dt], device=device)
            self.pooled, self.cat = {}, {}
            for i in range(1, S):
                t = pair(self.dims_in[i + 1], O.cpad(bch[i - 1]))
                self.pooled[i], self.pooled_lo[i] = t[0], t[1]
            for u in range(S + 1, 2 * S):
                t = pair(self.dims_in[u], bch[u - 2] + O.cpad(bch[2 * S - u - 1]))
                self.cat[u], self.cat_lo[u] = t[0], t[1]
        else:
            self.pooled = {i: O.alloc_cl(batch, self.dims_in[i + 1], O.cpad(bch[i - 1]), dt, device) for i in range(1, S)}
            self.cat = {u: O.alloc_cl(batch, self.dims_in[u], bch[u - 2] + O.cpad(bch[2 * S - u - 1]), dt, device) for u in range(S + 1, 2 * S)}
        for u in range(S + 1, 2 * S):
            assert self.cat[u].shape[-1] == self.conv[u][0].cpi
        self.ncls = ncls
        # plane-major concat buffers where both consumers of one are the DMA kernels (bf16, folded BatchNorm, channel
        # counts within the DMA weight-gradient kernel's tile limits): decided per up block
        self.cat_planar = {}
        for u in range(S + 1, 2 * S):
            c = self.conv[u][0]
            self.cat_planar[u] = bool(O.CAT_PLANAR and O.BN_SUMS_FROM_WGRAD and O.USE_DMA and dtype == L.SP_BF16 and c.fold
                                      and self.cat[u].shape[-1] % 16 == 0 and O.wgrad_dma_ok(c.cpi, c.cpo, dtype)
                                      and all(sb.tile["opp"] == 2 for sb in c.fwd_op.subs))
            c.x_planar = self.cat_planar[u]
        self.generation = 0         # bumped by every forward: a backward checks that its pass is still the resident one
        # IEEE-half storage: output gradients of a mean-type loss over n voxels are ~1/n, below half's subnormals for the volumes
        # this network sees -- the backward runs on S * gradients (S a power of two ~ n; everything in it is linear) into a
        # private buffer, which is added to the parameter gradients as 1/S of itself
        nvox_out = batch * self.out_dims[0] * self.out_dims[1] * self.out_dims[2]
        self.loss_scale = float(2.0 ** math.ceil(math.log2(max(2, nvox_out)))) if variant == "f16" else 1.0
        self._gpriv = None
        self._loss_scale_t = self._overflows = None
        # ---- fp8 mode: which layers run on the fp8 kernel, and where each one's e4m3 input comes from
        self.f8 = bool(f8)
        self._f8_fused = set()
        self._f8_train_only = set()
        self._training = True
        self.f8_src = {}            # layer -> ("y8", producer layer) | ("quant", bf16 source getter, plane-major?)
        if self.f8:
            from . import f8 as F8
            gs = F8.grad_scale_for(batch * self.out_dims[0] * self.out_dims[1] * self.out_dims[2])
            for i in range(1, 2 * S):
                c1, c2 = self.conv[i]
                for lay in (c1, c2):
                    if not isinstance(lay, FirstConvLayer):
                        lay.enable_f8(gs, fwd=f8_fwd)
                if c2.f8_fwd is not None:
                    if c1.y8_capable():      # (an fp8 layer, or the bf16 z-marching first layer: its epilogue writes the copy)
                        c1.want_y8 = True
                        self.f8_src[c2] = ("y8", c1)
                    else:
                        c2.x8 = F8.alloc_f8(batch, c2.in_dims, c2.cpi, device)
                        self.f8_src[c2] = ("quant", c1, False)
                elif c2.f8_wgrad_only:      # x8 only feeds the weight gradient: made in training steps only
                    if c1.y8_capable():
                        c1.want_y8 = True
                        self.f8_src[c2] = ("y8", c1)
                    else:
                        c2.x8 = F8.alloc_f8(batch, c2.in_dims, c2.cpi, device)
                        self.f8_src[c2] = ("quant", c1, False)
                        self._f8_train_only.add(c2)
                if c1.f8_fwd is not None or c1.f8_wgrad_only:
                    c1.x8 = F8.alloc_f8(batch, c1.in_dims, c1.cpi, device)
                    self.f8_src[c1] = ("quant", None, bool(self.cat_planar.get(i, False)))
                    if c1.f8_fwd is None:
                        self._f8_train_only.add(c1)

    # legacy names of the 3-scale engine (tools/)
    def __getattr__(self, name):
        if len(name) == 3 and name[0] == "c" and name[1:].isdigit() and "conv" in self.__dict__:
            i, j = int(name[1]), int(name[2])
            if i in self.conv and j in (1, 2):
                return self.conv[i][j - 1]
        raise AttributeError(name)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, images, params, bufs, training):
        """images: (B, n_in, D, H, W) fp32 on the device.  Returns seg (B, n_classes, D', H', W') fp32."""
        B, dt, S = self.batch, self.dtype, self.scales
        assert tuple(images.shape) == (B, self.channels[0]) + self.dims and images.dtype == torch.float32
        images = images.contiguous()
        self.generation += 1
        self._training = bool(training)
        self.scratch.zero()
        if training and "__nbt_flat__" in bufs:
            bufs["__nbt_flat__"].add_(1)
        st = (lambda l: l.in_sums) if training else (lambda l: None)
        c11 = self.conv[1][0]
        if self.first_packed:
            self.x0 = images                  # the packed first-layer kernels read the NCDHW fp32 input itself
            if training:
                c11.input_stats(images)
        else:
            O.ncdhw_to_cl(images, self.x0, dt)
            if training:
                O.bn_stats(self.x0, dt, c11.in_sums)
        x = self.x0
        if self.hl:
            return self._forward_hl(images, params, bufs, training, st)
        for i in range(1, S + 1):
            c1, c2 = self.conv[i]
            self._f8_input(c1, x)
            if i > 1:
                self._f8_e4m3_only(c1, c2, training)
            if i == 1 and self.f8 and self.first_packed and F8_Y1_E4M3 and c2.f8_fwd is not None and c1.want_y8 and c1.cpo in (16, 32):
                # fp8 mode: every reader of the first layer's output takes its e4m3 copy -- the second layer's forward and (fp8) weight
                # gradient as their operand, the first layer's own weight-gradient kernel for act'(y) and the BatchNorm-backward term
                # (sp_first_wgrad_fused_y8) -- so the 16-bit tensor, the largest of the step, is not written
                if training:
                    c2.x8 = c1.alloc_y8()          # (what _f8_input(c2, y1) does after the first layer ran; the backward plan looks at it)
                    c2._init_bwd()
                c1.store_y = bool(training and c2.f8_wgrad is None)
            y1 = c1.forward(x, params, bufs, training, st(c2))
            self._f8_input(c2, y1)
            self._f8_y2_e4m3_only(i, training)
            if i < S and training and not self.f8 and c2.can_pool():
                # MaxPool3d(2) in the convolution's epilogue: the pooled tensor and ITS statistics (the next BatchNorm's) from one kernel
                y2 = c2.forward(y1, params, bufs, training, st(self.conv[i + 1][0]), pool=(self.pooled[i], None))
                x = self.pooled[i]
                continue
            y2 = c2.forward(y1, params, bufs, training)
            if i < S:
                O.maxpool2_fwd(y2, self.pooled[i], dt, st(self.conv[i + 1][0]), q8=self._f8_fused_input(self.conv[i + 1][0]),
                               x8=None if c2.store_y else c2.y8)
                x = self.pooled[i]
        low = y2
        for u in range(S + 1, 2 * S):
            c1, c2 = self.conv[u]
            q8 = self._f8_fused_input(c1) if self.cat_planar[u] and low.shape[-1] % 16 == 0 else None
            # fp8 forward AND fp8 weight gradient: nobody reads the 16-bit concat buffer (the backward kernels of its two
            # producers recompute their half from y) -- it is not written
            keep = True
            if q8 is not None and c1.f8_fwd is not None and ConvLayer.SKIP_DZ:
                if training:
                    c1._init_bwd()
                keep = training and c1.f8_wgrad is None
            sk = self.conv[2 * S - u][1]
            assert sk.store_y or q8 is not None
            O.upsample2_crop_cat_fwd(low, sk.y, self.cat[u], dt, st(c1), planar=self.cat_planar[u], q8=q8, store=keep,
                                     skip8=None if sk.store_y else sk.y8)
            self._f8_input(c1, self.cat[u])
            self._f8_e4m3_only(c1, c2, training)
            y1 = c1.forward(self.cat[u], params, bufs, training, st(c2))
            self._f8_input(c2, y1)
            low = c2.forward(y1, params, bufs, training)
        seg = torch.empty((B, self.ncls) + self.out_dims, dtype=torch.float32, device=self.device)
        if self.fused_head:
            nv = self.out_dims[0] * self.out_dims[1] * self.out_dims[2]
            L.call("sp_head_fwd", O.ptr(low), dt, nv, B, low.shape[-1], self.channels[-3], O.ptr(params["classify.0.weight"]),
                   O.ptr(params["classify.0.bias"]), self.channels[-2], O.ptr(params["classify.2.weight"]),
                   O.ptr(params["classify.2.bias"]), self.ncls, LEAKY, O.ptr(seg), O.stream())
            return seg
        h = self.h0.forward(low, params, bufs, training)
        o = self.h2.forward(h, params, bufs, training)
        O.cl_to_ncdhw(o, seg, L.SP_F32)
        return seg

    def _forward_hl(self, images, params, bufs, training, st):
        """the forward pass on bf16 pairs ("bf16x3"): same data flow, every tensor a (hi, lo) pair; hi halves land where the
        bf16 backward expects its activations"""
        B, S = self.batch, self.scales
        lod = lambda hi, lo: lo.data_ptr() - hi.data_ptr()
        x, x_lo = images, None
        for i in range(1, S + 1):
            c1, c2 = self.conv[i]
            y1 = c1.forward(x, params, bufs, training, st(c2)) if i == 1 else c1.forward(x, params, bufs, training, st(c2), x_lo=x_lo)
            if i < S and training and c2.can_pool():
                p, p_lo = self.pooled[i], self.pooled_lo[i]
                y2 = c2.forward(y1, params, bufs, training, st(self.conv[i + 1][0]), x_lo=c1.y_lo, pool=(p, p_lo))
                x, x_lo = p, p_lo
                continue
            y2 = c2.forward(y1, params, bufs, training, x_lo=c1.y_lo)
            if i < S:
                p, p_lo = self.pooled[i], self.pooled_lo[i]
                _, D, H, W, CP = y2.shape
                L.call("sp_maxpool2_fwd_hl", O.ptr(y2), lod(y2, c2.y_lo), O.ptr(p), lod(p, p_lo), B, D, H, W, CP,
                       O.ptr(st(self.conv[i + 1][0])), O.stream())
                x, x_lo = p, p_lo
        low, low_lo = y2, c2.y_lo
        for u in range(S + 1, 2 * S):
            c1, c2 = self.conv[u]
            skip = self.conv[2 * S - u][1]
            cat, cat_lo = self.cat[u], self.cat_lo[u]
            _, D, H, W, CPu = low.shape
            _, Ds, Hs, Ws, CPs = skip.y.shape
            L.call("sp_upsample2_crop_cat_fwd_hl", O.ptr(low), lod(low, low_lo), CPu, O.ptr(skip.y), lod(skip.y, skip.y_lo), CPs,
                   O.ptr(cat), lod(cat, cat_lo), CPu + CPs, B, D, H, W, Ds, Hs, Ws,
                   (B * 8 * D * H * W * 16) if self.cat_planar[u] else 0, O.ptr(st(c1)), O.stream())
            y1 = c1.forward(cat, params, bufs, training, st(c2), x_lo=cat_lo)
            low = c2.forward(y1, params, bufs, training, x_lo=c1.y_lo)
            low_lo = c2.y_lo
        seg = torch.empty((B, self.ncls) + self.out_dims, dtype=torch.float32, device=self.device)
        nv = self.out_dims[0] * self.out_dims[1] * self.out_dims[2]
        L.call("sp_head_fwd_hl", O.ptr(low), lod(low, low_lo), nv, B, low.shape[-1], self.channels[-3], O.ptr(params["classify.0.weight"]),
               O.ptr(params["classify.0.bias"]), self.channels[-2], O.ptr(params["classify.2.weight"]),
               O.ptr(params["classify.2.bias"]), self.ncls, LEAKY, O.ptr(seg), O.stream())
        return seg

    def _f8_e4m3_only(self, c1, c2, training):
        """fp8 mode: the first convolution of a block whose output is read by fp8 kernels only -- the second convolution's forward
        and weight gradient take its e4m3 copy as their operand, sp_bn_act_bwd_y8 takes it for act'(y) and the BatchNorm-backward
        term -- does not write the 16-bit tensor (``c1.store_y = False``)"""
        from . import f8 as F8
        c1.store_y = True
        if not (self.f8 and F8_Y1_E4M3 and isinstance(c1.f8_fwd, (F8.ConvRunnerF8, F8.ConvRunnerF8Split)) and c1.want_y8 and c2.f8_fwd is not None
                and self.f8_src.get(c2, (None,))[0] == "y8"):
            return
        if training:
            c2.x8 = c1.alloc_y8()
            c2._init_bwd()
            if c2.f8_wgrad is None:
                return
        c1.store_y = False

    def _f8_y2_e4m3_only(self, i, training):
        """fp8 mode, down block i: the second convolution's output is read by the pooling kernel, by the concatenation of up block
        2S - i (its skip half) and by sp_pool_skip_act_bwd -- all three take the e4m3 copy (sp_maxpool2_fwd_x8,
        sp_upsample2_crop_cat_fwd_q8s8, sp_pool_skip_act_bwd_y8), so the 16-bit tensor is not written"""
        from . import f8 as F8
        S = self.scales
        c2 = self.conv[i][1]
        c2.store_y = True
        if not (self.f8 and F8_Y2_E4M3 and i < S and isinstance(c2.f8_fwd, (F8.ConvRunnerF8, F8.ConvRunnerF8Split)) and c2.cpo % 16 == 0):
            return
        u = 2 * S - i
        cu1 = self.conv[u][0]
        src = self.f8_src.get(cu1)
        if not (self.cat_planar.get(u) and self.conv[u - 1][1].cpo % 16 == 0 and src is not None and src[0] == "quant" and ConvLayer.FUSE_Q8
                and (training or cu1 not in self._f8_train_only)):      # (the row-ordered concat kernel with its e4m3 output: _f8_fused_input(cu1))
            return
        c2.want_y8 = True
        c2.store_y = False

    def _f8_fused_input(self, lay):
        """fp8 mode: (lay.x8, e4m3, 1.0) when the pooling / concatenation kernel that writes lay's input should write its e4m3
        copy as well (the separate quantisation pass is then skipped once), else None"""
        if self.f8_src.get(lay) is None or self.f8_src[lay][0] != "quant" or not ConvLayer.FUSE_Q8:
            return None
        if lay in self._f8_train_only and not self._training:
            return None
        from . import f8 as F8
        self._f8_fused.add(lay)
        return (lay.x8, F8.E4M3, 1.0)

    def _f8_input(self, lay, x):
        """fp8 mode: make ``lay.x8`` the e4m3 plane-major copy of its input x -- written by the producing fp8 convolution's
        epilogue or by the pooling / concatenation kernel, else by one quantisation pass over the bf16 tensor"""
        src = self.f8_src.get(lay)
        if src is None or (lay in self._f8_train_only and not self._training):
            return
        if lay in self._f8_fused:
            self._f8_fused.discard(lay)
            return
        if src[0] == "y8":
            lay.x8 = src[1].alloc_y8()
        else:
            from . import f8 as F8
            F8.quantize(x, lay.x8, F8.E4M3, 1.0, src_planar=src[2])

    def _up_bwd(self, low, cat, g, coef, planar):
        """gradient of the upsampled half of a concat input -> dz of the low-resolution producer `low`"""
        dt = self.dtype
        # fp8 mode: the ring / tiled kernels write the e5m2 copy themselves (not the gather fallback, which needs `cat`)
        cp = low.y.shape[-1]
        q8 = low.dz8_out() if ((planar or isinstance(g, tuple)) and cp % 16 == 0 and 256 % (cp // 8) == 0 and cp <= 256
                               and dt == L.SP_BF16 and min(low.y.shape[1:4]) >= 2) else None
        dz = low.dz_target() if q8 is not None else low.dz
        if isinstance(g, tuple):      # dense per-part gradient tensors (ConvLayer split_g)
            O.upsample2_act_bwd(low.y, None, g[0], coef, dt, L.ACT_LEAKY, LEAKY, dz, low.dbias_sums, coef_stride=cat.shape[-1], q8=q8)
        else:
            O.upsample2_act_bwd(low.y, None if planar else cat, g, coef, dt, L.ACT_LEAKY, LEAKY, dz, low.dbias_sums, q8=q8)

    def _skip_bwd(self, prod, gp, coefp, cat, g, coef, c_up):
        """pool gradient + skip half of the concat gradient -> dz of the block output `prod`"""
        dt = self.dtype
        if isinstance(g, tuple):
            O.pool_skip_act_bwd(prod.y, gp, coefp, None, g[1], coef, 0, dt, L.ACT_LEAKY, LEAKY, prod.dz_target(), prod.dbias_sums,
                                coef_c0=c_up, coef_stride=cat.shape[-1], q8=prod.dz8_out(), y8=None if prod.store_y else prod.y8)
        else:
            O.pool_skip_act_bwd(prod.y, gp, coefp, cat, g, coef, c_up, dt, L.ACT_LEAKY, LEAKY, prod.dz_target(), prod.dbias_sums,
                                q8=prod.dz8_out(), y8=None if prod.store_y else prod.y8)

    def _block_inner_bwd(self, c1, c2, params, grads):
        """backward of a block's second convolution down to the dz of its first one (Unet3D.py:18-24: BatchNorm -> conv -> LeakyReLU
        -> BatchNorm -> conv): where the data gradient's kernel has the epilogue, g never reaches memory (ConvLayer.can_fuse_dz)"""
        if c2.can_fuse_dz(c1):
            c2.backward(c1.y, params, grads, fuse_dz=(c1.dz, c1.dbias_sums, L.ACT_LEAKY, LEAKY))
            return
        g, coef = c2.backward(c1.y, params, grads)
        O.bn_act_bwd(g, c1.y, coef, self.dtype, L.ACT_LEAKY, LEAKY, c1.dz_target(), c1.dbias_sums, q8=c1.dz8_out(), y8=None if c1.store_y else c1.y8)

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, dseg, seg, params, grads, ready=None):
        if self.loss_scale == 1.0:
            return self._backward(dseg, seg, params, grads, ready)
        # IEEE-half build: the backward runs on S * gradients into a private buffer, added to the parameter gradients as 1/S of
        # itself.  S is DYNAMIC and lives on the device (capturable: a replayed hipGraph adapts it too): a step whose scaled
        # gradients are not all finite (an output gradient or a BatchNorm-backward coefficient pushed a 16-bit dz / g past
        # 65504) contributes nothing and halves S for the following steps; ``overflow_steps`` counts them.
        if self._loss_scale_t is None:
            self._loss_scale_t = torch.full((), self.loss_scale, dtype=torch.float32, device=self.device)
            self._overflows = torch.zeros((), dtype=torch.int64, device=self.device)
            self._clean_steps = torch.zeros((), dtype=torch.int64, device=self.device)
        S = self._loss_scale_t
        names = list(grads)
        n = sum(grads[k].numel() for k in names)
        if self._gpriv is None or self._gpriv.numel() != n:
            self._gpriv = torch.empty(n, dtype=torch.float32, device=self.device)
        self._gpriv.zero_()
        priv, off = {}, 0
        for k in names:
            priv[k] = self._gpriv[off:off + grads[k].numel()].view(grads[k].shape)
            off += grads[k].numel()
        self._backward(dseg * S, seg, params, priv, None)
        ok = torch.isfinite(self._gpriv).all()
        inv = ok.to(torch.float32) / S                      # 0 for a step that overflowed
        gsafe = torch.where(ok, self._gpriv, torch.zeros((), dtype=torch.float32, device=self.device))
        first = grads[names[0]]
        flat_ok = all(grads[k].is_contiguous() for k in names) and \
            all(grads[b].data_ptr() == grads[a].data_ptr() + 4 * grads[a].numel() for a, b in zip(names, names[1:]))
        if flat_ok:      # the views of one flat buffer (runtime/flat.py): one fused multiply-add
            torch.as_strided(first, (n,), (1,)).addcmul_(gsafe, inv)
        else:
            off = 0
            for k in names:
                grads[k].addcmul_(gsafe[off:off + grads[k].numel()].view(grads[k].shape), inv)
                off += grads[k].numel()
        self._overflows.add_((~ok).to(torch.int64))
        # halve on overflow; after LOSS_SCALE_GROWTH_STEPS clean steps in a row double again, up to the initial scale (a
        # scale that only shrinks gives up gradient range for good after a few early overflows).  All on the device: capturable.
        self._clean_steps = torch.where(ok, self._clean_steps + 1, torch.zeros_like(self._clean_steps))
        grow = self._clean_steps >= self.LOSS_SCALE_GROWTH_STEPS
        before = S.clone()
        S.mul_(torch.where(ok, torch.where(grow, 2.0, 1.0), 0.5).to(torch.float32)).clamp_(min=1.0)
        # (growth stops at the initial scale; a scale above it -- set by hand -- only shrinks)
        S.copy_(torch.where(grow & ok, torch.minimum(S, before.clamp(min=self.loss_scale)), S))
        self._clean_steps = torch.where(grow, torch.zeros_like(self._clean_steps), self._clean_steps)
        if ready is not None:
            ready("block1.")         # every gradient is final only now: one exchange

    @property
    def overflow_steps(self):
        """f16 build: training steps whose scaled gradients overflowed (skipped, loss scale halved); host read = one sync"""
        return 0 if self._overflows is None else int(self._overflows)

    def _backward(self, dseg, seg, params, grads, ready=None):
        """dseg: dL/dseg (NCDHW fp32).  Accumulates into ``grads[name]`` (fp32 tensors, parameter layout).
        Must follow a training-mode forward on the same engine (activations are kept in the layers).
        ready(prefix): called when every gradient from the first parameter named ``prefix*`` to the end of the flat
        buffer is final (data-parallel buckets, runtime/flat.py)."""
        dt, S = self.dtype, self.scales
        for l in self.layers:
            if not (self.fused_head and l in (self.h0, self.h2)):
                l._init_bwd()
        dseg = dseg.contiguous()
        last = self.conv[2 * S - 1][1]
        # all data-gradient weight re-packs depend on the parameters only: side stream, beside the head's backward
        pre = O.fork()
        with pre:      # ... and ONE launch for all of them
            O.prep_batch([(l.dgrad, params[l.conv_prefix + ".weight"]) for l in self.layers
                          if getattr(l, "dgrad", None) is not None and l.need_input_grad and l.f8_dgrad is None])
            f8l = [l for l in self.layers if l.f8_dgrad is not None]
            if f8l:      # ... and ONE for the e4m3 fragments of every fp8 data gradient
                from . import f8 as F8
                F8.prep_many([j for l in f8l for j in l.f8_dgrad.prep_jobs(params[l.conv_prefix + ".weight"], 1.0 / l.f8_grad_scale)])
        if self.fused_head:
            nv = self.out_dims[0] * self.out_dims[1] * self.out_dims[2]
            b5, bc, ncls = self.channels[-3], self.channels[-2], self.ncls
            lib = L.load()
            rows = lib.sp_head_bwd_rows(self.batch * nv)
            if getattr(self, "_hpart", None) is None:
                self._hpart = torch.empty(rows * lib.sp_head_row_floats(b5, bc, ncls), dtype=torch.float32, device=self.device)
            q8 = last.dz8_out() if last.y.shape[-1] == b5 else None      # fp8 mode: the e5m2 copy of dz straight from this kernel
            if q8 is not None:
                L.call("sp_head_bwd_q8", O.ptr(last.y), dt, nv, self.batch, last.y.shape[-1], b5,
                       O.ptr(params["classify.0.weight"]), O.ptr(params["classify.0.bias"]), bc,
                       O.ptr(params["classify.2.weight"]), ncls, LEAKY, O.ptr(seg), O.ptr(dseg), L.ACT_LEAKY, LEAKY,
                       O.ptr(last.dz_target()), O.ptr(self._hpart), *O._q8_args(q8, self.batch * nv), O.stream())
            else:
                L.call("sp_head_bwd", O.ptr(last.y), dt, nv, self.batch, last.y.shape[-1], b5,
                       O.ptr(params["classify.0.weight"]), O.ptr(params["classify.0.bias"]), bc,
                       O.ptr(params["classify.2.weight"]), ncls, LEAKY, O.ptr(seg), O.ptr(dseg), L.ACT_LEAKY, LEAKY,
                       O.ptr(last.dz), O.ptr(self._hpart), O.stream())
            L.call("sp_head_grad_finish", O.ptr(self._hpart), rows, b5, bc, ncls, O.ptr(grads["classify.0.weight"]),
                   O.ptr(grads["classify.0.bias"]), O.ptr(grads["classify.2.weight"]), O.ptr(grads["classify.2.bias"]),
                   O.ptr(last.dbias_sums), O.stream())
        else:
            # output side: dz of the last 1x1 conv = dseg * sigmoid'(seg)
            O.out_grad_to_cl(dseg, seg, dt, L.ACT_SIGMOID, 0.0, self.h2.dz, self.h2.dbias_sums)
            g, _ = self.h2.backward(self.h0.y, params, grads)
            O.bn_act_bwd(g, self.h0.y, None, dt, L.ACT_LEAKY, LEAKY, self.h0.dz, self.h0.dbias_sums)
            g, _ = self.h0.backward(last.y, params, grads)
            O.bn_act_bwd(g, last.y, None, dt, L.ACT_LEAKY, LEAKY, last.dz_target(), last.dbias_sums, q8=last.dz8_out())
        pre.join()
        skip = {}                     # down block index -> (concat buffer, its gradient, coefficients, channels of the upsampled part)
        for u in range(2 * S - 1, S, -1):
            c1, c2 = self.conv[u]
            self._block_inner_bwd(c1, c2, params, grads)
            gu, coefu = c1.backward(self.cat[u], params, grads)
            if ready is not None and u == S + 1:     # every up block and the head are final: their all-reduce bucket may start
                ready("block%d." % (S + 1))
            self._up_bwd(self.conv[u - 1][1], self.cat[u], gu, coefu, self.cat_planar[u])
            skip[2 * S - u] = (self.cat[u], gu, coefu, self.channels[u - 1])
        for i in range(S, 0, -1):
            c1, c2 = self.conv[i]
            if i > 1:
                self._block_inner_bwd(c1, c2, params, grads)
                gp, coefp = c1.backward(self.pooled[i - 1], params, grads)
                if ready is not None and i == 2:
                    ready("block2.")
                cat, gu, coefu, c_up = skip[i - 1]
                self._skip_bwd(self.conv[i - 1][1], gp, coefp, cat, gu, coefu, c_up)
                continue
            g, coef = c2.backward(c1.y, params, grads)
            if self.first_packed and coef is not None and c1.cpo in (16, 32):
                c1.backward(self.x0, params, grads, g=g, coef=coef)     # dz formed inside the weight-gradient kernel
            else:
                O.bn_act_bwd(g, c1.y, coef, dt, L.ACT_LEAKY, LEAKY, c1.dz, c1.dbias_sums)
                c1.backward(self.x0, params, grads)       # only the first BatchNorm's gamma/beta need this dgrad
