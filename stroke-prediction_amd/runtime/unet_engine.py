"""Forward / backward of the 3-scale 3-D U-Net (``Unet3D.forward`` Unet3D.py:56-79) on HIP kernels.

Data flow (channels-last tensors, BatchNorm folded into the consuming convolution's load):

  x0 -b1c1-> y11 -b1c2-> y12 -pool-> p1 -b2c1-> y21 -b2c2-> y22 -pool-> p2 -b3c1-> y31 -b3c2-> y32
  cat4 = [up(y32) | crop(y22)] -b4c1-> y41 -b4c2-> y42 ; cat5 = [up(y42) | crop(y12)] -b5c1-> y51 -b5c2-> y52
  y52 -1x1,lrelu-> h -1x1,sigmoid-> seg

Batch statistics of every BatchNorm input are produced by the kernel that writes that tensor
(conv / pool / upsample / crop epilogues); only the network input needs a stand-alone pass.
"""
import os

import torch

from . import lib as L
from . import ops as O
from .layers import ConvLayer, FirstConvLayer, Scratch

LEAKY = 0.01


def unet_out_dims(dims):
    out = []
    for n in dims:
        b3 = ((n - 4) // 2 - 4) // 2 - 4
        out.append(4 * b3 - 12)
    return tuple(out)


class UnetEngine:
    """Bound to one (batch, spatial size, dtype); the model keeps a small cache of engines."""

    def __init__(self, channels, batch, dims, dtype, device):
        O.require_gpu()
        L.load()
        n_in, b1, b2, b3, b4, b5, bc, ncls = channels
        assert ncls <= 8, "the classify head supports up to 8 classes"
        self.channels, self.batch, self.dims, self.dtype, self.device = list(channels), batch, tuple(dims), dtype, device
        for n in dims:
            b3d = ((n - 4) // 2 - 4) // 2 - 4
            if b3d < 1 or 4 * b3d - 12 < 1:
                raise ValueError("Unet3D: spatial size %s is too small for three scales of valid 3x3x3 "
                                 "convolutions (minimum 44 per axis)" % (tuple(dims),))
        self.scratch = sc = Scratch(device)
        mk = lambda name, ci, co, d, bn=True, k=3, act=L.ACT_LEAKY, ap=LEAKY, out_dtype=None, blk=None, idx=None, \
            need_g=True, cpi=None, split=None: ConvLayer(name, "conv", ci, co, k, 1, 0, d, batch, dtype, device, sc,
                                             bn_prefix=("%s.bn_conv_relu_2x.%d" % (blk, idx)) if bn else None,
                                             conv_prefix=("%s.bn_conv_relu_2x.%d" % (blk, idx + 1)) if bn else name,
                                             act=act, act_param=ap, out_dtype=out_dtype, need_input_grad=need_g, cpi=cpi,
                                             split_g=split)
        sub = lambda d, k: tuple(x - k for x in d)
        half = lambda d: tuple(x // 2 for x in d)
        dbl = lambda d: tuple(2 * x for x in d)
        d0 = self.dims
        # the network input gets a 16-channel pitch (not 8): every 3x3x3 layer then meets the 16-channel plane
        # granularity of the DMA weight-gradient kernel
        self.first_packed = FirstConvLayer.supported(n_in, b1, 3, 1, 0, dtype, True) and not os.environ.get("SP_GENERIC_FIRST")
        if self.first_packed:   # two-channel input: packed-K kernels reading the NCDHW fp32 input directly
            self.c11 = FirstConvLayer("b1c1", "conv", n_in, b1, 3, 1, 0, d0, batch, dtype, device, sc,
                                      bn_prefix="block1.bn_conv_relu_2x.0", conv_prefix="block1.bn_conv_relu_2x.1",
                                      act=L.ACT_LEAKY, act_param=LEAKY, need_input_grad=False, cpi=O.cpad(n_in, 16))
        else:
            self.c11 = mk("b1c1", n_in, b1, d0, blk="block1", idx=0, need_g=False, cpi=O.cpad(n_in, 16))
        self.c12 = mk("b1c2", b1, b1, sub(d0, 2), blk="block1", idx=3)
        d12 = sub(d0, 4)
        dp1 = half(d12)
        self.c21 = mk("b2c1", b1, b2, dp1, blk="block2", idx=0)
        self.c22 = mk("b2c2", b2, b2, sub(dp1, 2), blk="block2", idx=3)
        d22 = sub(dp1, 4)
        dp2 = half(d22)
        self.c31 = mk("b3c1", b2, b3, dp2, blk="block3", idx=0)
        self.c32 = mk("b3c2", b3, b3, sub(dp2, 2), blk="block3", idx=3)
        d32 = sub(dp2, 4)
        dc4 = dbl(d32)
        assert b3 % 8 == 0 and b4 % 8 == 0, "up-path channel counts must be multiples of 8"
        split_ok = lambda cu, cs: cu if (256 % (cu // 8) == 0 and cu % 16 == 0 and cs % 16 == 0 and os.environ.get("SP_SPLIT_G")) else None   # measured: two data-gradient launches cost more (+45 us) than the dense reads save -> opt-in
        self.c41 = mk("b4c1", b3 + b2, b4, dc4, blk="block4", idx=0, split=split_ok(b3, O.cpad(b2)))
        self.c42 = mk("b4c2", b4, b4, sub(dc4, 2), blk="block4", idx=3)
        d42 = sub(dc4, 4)
        dc5 = dbl(d42)
        self.c51 = mk("b5c1", b4 + b1, b5, dc5, blk="block5", idx=0, split=split_ok(b4, O.cpad(b1)))
        self.c52 = mk("b5c2", b5, b5, sub(dc5, 2), blk="block5", idx=3)
        d52 = sub(dc5, 4)
        self.h0 = mk("classify.0", b5, bc, d52, bn=False, k=1)
        self.h2 = mk("classify.2", bc, ncls, d52, bn=False, k=1, act=L.ACT_SIGMOID, ap=0.0, out_dtype=L.SP_F32)
        self.out_dims = d52
        # fused pointwise head where a kernel exists for (C, CH, NC); the two generic 1x1 layers otherwise
        self.fused_head = bool(L.load().sp_head_supported(b5, bc, ncls)) and b5 % 8 == 0 and not os.environ.get("SP_GENERIC_HEAD")
        self.d12, self.dp1, self.d22, self.dp2, self.d32, self.dc4, self.d42, self.dc5 = d12, dp1, d22, dp2, d32, dc4, d42, dc5
        self.layers = [self.c11, self.c12, self.c21, self.c22, self.c31, self.c32, self.c41, self.c42, self.c51,
                       self.c52, self.h0, self.h2]
        for l in self.layers:
            l.reserve_bwd_scratch()
        sc.finalize()
        dt = dtype
        self.x0 = None if self.first_packed else O.alloc_cl(batch, d0, self.c11.cpi, dt, device)
        self.p1 = O.alloc_cl(batch, dp1, O.cpad(b1), dt, device)
        self.p2 = O.alloc_cl(batch, dp2, O.cpad(b2), dt, device)
        self.cat4 = O.alloc_cl(batch, dc4, b3 + O.cpad(b2), dt, device)
        self.cat5 = O.alloc_cl(batch, dc5, b4 + O.cpad(b1), dt, device)
        assert self.cat4.shape[-1] == self.c41.cpi and self.cat5.shape[-1] == self.c51.cpi
        self.ncls = ncls
        # plane-major concat buffers when both consumers of each are the DMA kernels (bf16, folded BatchNorm)
        self.cat_planar = bool(O.CAT_PLANAR and O.BN_SUMS_FROM_WGRAD and O.USE_DMA and dtype == L.SP_BF16 and self.c41.fold and self.c51.fold
                               and self.cat4.shape[-1] % 16 == 0 and self.cat5.shape[-1] % 16 == 0)
        self.c41.x_planar = self.c51.x_planar = self.cat_planar

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, images, params, bufs, training):
        """images: (B, n_in, D, H, W) fp32 on the device.  Returns seg (B, n_classes, D', H', W') fp32."""
        B, dt = self.batch, self.dtype
        assert tuple(images.shape) == (B, self.channels[0]) + self.dims and images.dtype == torch.float32
        images = images.contiguous()
        self.scratch.zero()
        if training and "__nbt_flat__" in bufs:
            bufs["__nbt_flat__"].add_(1)
        st = (lambda l: l.in_sums) if training else (lambda l: None)
        if self.first_packed:
            self.x0 = images                  # the packed first-layer kernels read the NCDHW fp32 input itself
            if training:
                self.c11.input_stats(images)
        else:
            O.ncdhw_to_cl(images, self.x0, dt)
            if training:
                O.bn_stats(self.x0, dt, self.c11.in_sums)
        y11 = self.c11.forward(self.x0, params, bufs, training, st(self.c12))
        y12 = self.c12.forward(y11, params, bufs, training)
        O.maxpool2_fwd(y12, self.p1, dt, st(self.c21))
        y21 = self.c21.forward(self.p1, params, bufs, training, st(self.c22))
        y22 = self.c22.forward(y21, params, bufs, training)
        O.maxpool2_fwd(y22, self.p2, dt, st(self.c31))
        y31 = self.c31.forward(self.p2, params, bufs, training, st(self.c32))
        y32 = self.c32.forward(y31, params, bufs, training)
        c3 = self.channels[3]
        s4 = st(self.c41)
        O.upsample2_crop_cat_fwd(y32, y22, self.cat4, dt, s4, planar=self.cat_planar)
        y41 = self.c41.forward(self.cat4, params, bufs, training, st(self.c42))
        y42 = self.c42.forward(y41, params, bufs, training)
        c4 = self.channels[4]
        s5 = st(self.c51)
        O.upsample2_crop_cat_fwd(y42, y12, self.cat5, dt, s5, planar=self.cat_planar)
        y51 = self.c51.forward(self.cat5, params, bufs, training, st(self.c52))
        y52 = self.c52.forward(y51, params, bufs, training)
        seg = torch.empty((B, self.ncls) + self.out_dims, dtype=torch.float32, device=self.device)
        if self.fused_head:
            nv = self.out_dims[0] * self.out_dims[1] * self.out_dims[2]
            L.call("sp_head_fwd", O.ptr(y52), dt, nv, B, y52.shape[-1], self.channels[5], O.ptr(params["classify.0.weight"]),
                   O.ptr(params["classify.0.bias"]), self.channels[6], O.ptr(params["classify.2.weight"]),
                   O.ptr(params["classify.2.bias"]), self.ncls, LEAKY, O.ptr(seg), O.stream())
            return seg
        h = self.h0.forward(y52, params, bufs, training)
        o = self.h2.forward(h, params, bufs, training)
        O.cl_to_ncdhw(o, seg, L.SP_F32)
        return seg

    def _up_bwd(self, low, cat, g, coef):
        """gradient of the upsampled half of a concat input -> dz of the low-resolution producer `low`"""
        dt = self.dtype
        if isinstance(g, tuple):      # dense per-part gradient tensors (ConvLayer split_g)
            O.upsample2_act_bwd(low.y, None, g[0], coef, dt, L.ACT_LEAKY, LEAKY, low.dz, low.dbias_sums, coef_stride=cat.shape[-1])
        else:
            O.upsample2_act_bwd(low.y, None if self.cat_planar else cat, g, coef, dt, L.ACT_LEAKY, LEAKY, low.dz, low.dbias_sums)

    def _skip_bwd(self, prod, gp, coefp, cat, g, coef, c_up):
        """pool gradient + skip half of the concat gradient -> dz of the block output `prod`"""
        dt = self.dtype
        if isinstance(g, tuple):
            O.pool_skip_act_bwd(prod.y, gp, coefp, None, g[1], coef, 0, dt, L.ACT_LEAKY, LEAKY, prod.dz, prod.dbias_sums,
                                coef_c0=c_up, coef_stride=cat.shape[-1])
        else:
            O.pool_skip_act_bwd(prod.y, gp, coefp, cat, g, coef, c_up, dt, L.ACT_LEAKY, LEAKY, prod.dz, prod.dbias_sums)

    # ------------------------------------------------------------------------------------------ backward
    def backward(self, dseg, seg, params, grads):
        """dseg: dL/dseg (NCDHW fp32).  Accumulates into ``grads[name]`` (fp32 tensors, parameter layout).
        Must follow a training-mode forward on the same engine (activations are kept in the layers)."""
        dt = self.dtype
        for l in self.layers:
            if not (self.fused_head and l in (self.h0, self.h2)):
                l._init_bwd()
        c = self
        dseg = dseg.contiguous()
        # all data-gradient weight re-packs depend on the parameters only: side stream, beside the head's backward
        pre = O.fork()
        with pre:      # ... and ONE launch for all of them
            O.prep_batch([(l.dgrad, params[l.conv_prefix + ".weight"]) for l in self.layers
                          if getattr(l, "dgrad", None) is not None and l.need_input_grad])
        if self.fused_head:
            nv = self.out_dims[0] * self.out_dims[1] * self.out_dims[2]
            b5, bc, ncls = self.channels[5], self.channels[6], self.ncls
            lib = L.load()
            rows = lib.sp_head_bwd_rows(self.batch * nv)
            if getattr(self, "_hpart", None) is None:
                self._hpart = torch.empty(rows * lib.sp_head_row_floats(b5, bc, ncls), dtype=torch.float32, device=self.device)
            L.call("sp_head_bwd", O.ptr(c.c52.y), dt, nv, self.batch, c.c52.y.shape[-1], b5,
                   O.ptr(params["classify.0.weight"]), O.ptr(params["classify.0.bias"]), bc,
                   O.ptr(params["classify.2.weight"]), ncls, LEAKY, O.ptr(seg), O.ptr(dseg), L.ACT_LEAKY, LEAKY,
                   O.ptr(c.c52.dz), O.ptr(self._hpart), O.stream())
            L.call("sp_head_grad_finish", O.ptr(self._hpart), rows, b5, bc, ncls, O.ptr(grads["classify.0.weight"]),
                   O.ptr(grads["classify.0.bias"]), O.ptr(grads["classify.2.weight"]), O.ptr(grads["classify.2.bias"]),
                   O.ptr(c.c52.dbias_sums), O.stream())
        else:
            # output side: dz of the last 1x1 conv = dseg * sigmoid'(seg)
            O.out_grad_to_cl(dseg, seg, dt, L.ACT_SIGMOID, 0.0, c.h2.dz, c.h2.dbias_sums)
            g, _ = c.h2.backward(c.h0.y, params, grads)
            O.bn_act_bwd(g, c.h0.y, None, dt, L.ACT_LEAKY, LEAKY, c.h0.dz, c.h0.dbias_sums)
            g, _ = c.h0.backward(c.c52.y, params, grads)
            O.bn_act_bwd(g, c.c52.y, None, dt, L.ACT_LEAKY, LEAKY, c.c52.dz, c.c52.dbias_sums)
        pre.join()
        g, coef = c.c52.backward(c.c51.y, params, grads)
        O.bn_act_bwd(g, c.c51.y, coef, dt, L.ACT_LEAKY, LEAKY, c.c51.dz, c.c51.dbias_sums)
        g5, coef5 = c.c51.backward(c.cat5, params, grads)
        self._up_bwd(c.c42, c.cat5, g5, coef5)
        g, coef = c.c42.backward(c.c41.y, params, grads)
        O.bn_act_bwd(g, c.c41.y, coef, dt, L.ACT_LEAKY, LEAKY, c.c41.dz, c.c41.dbias_sums)
        g4, coef4 = c.c41.backward(c.cat4, params, grads)
        self._up_bwd(c.c32, c.cat4, g4, coef4)
        g, coef = c.c32.backward(c.c31.y, params, grads)
        O.bn_act_bwd(g, c.c31.y, coef, dt, L.ACT_LEAKY, LEAKY, c.c31.dz, c.c31.dbias_sums)
        gp2, coefp2 = c.c31.backward(c.p2, params, grads)
        self._skip_bwd(c.c22, gp2, coefp2, c.cat4, g4, coef4, self.channels[3])
        g, coef = c.c22.backward(c.c21.y, params, grads)
        O.bn_act_bwd(g, c.c21.y, coef, dt, L.ACT_LEAKY, LEAKY, c.c21.dz, c.c21.dbias_sums)
        gp1, coefp1 = c.c21.backward(c.p1, params, grads)
        self._skip_bwd(c.c12, gp1, coefp1, c.cat5, g5, coef5, self.channels[4])
        g, coef = c.c12.backward(c.c11.y, params, grads)
        if self.first_packed and coef is not None and c.c11.cpo == 16:
            c.c11.backward(c.x0, params, grads, g=g, coef=coef)     # dz formed inside the weight-gradient kernel
        else:
            O.bn_act_bwd(g, c.c11.y, coef, dt, L.ACT_LEAKY, LEAKY, c.c11.dz, c.c11.dbias_sums)
            c.c11.backward(c.x0, params, grads)       # only the first BatchNorm's gamma/beta need this dgrad
