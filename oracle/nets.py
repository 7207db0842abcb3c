"""Functional CPU restatement of the reference hot path (test infrastructure).

All functions take a flat ``state_dict``-style mapping of tensors (same keys as
the reference modules) so that autograd works on leaf tensors directly.  Every
function cites the reference lines it restates.  Parity status: pinned by
``tests/golden`` (see ``oracle/__init__.py``).
"""
import math

import torch
import torch.nn.functional as F

from .weights import ENC_LAYERS, DEC_LAYERS

BN_EPS = 1e-5        # torch.nn.BatchNorm3d default, used at Unet3D.py:18,21 / Cae3D.py:40...
BN_MOMENTUM = 0.1
LEAKY = 0.01         # Unet3D.py:20,23,51


class _RoundBf16(torch.autograd.Function):
    """Round to bf16 storage precision in the forward, identity in the backward."""

    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def round_bf16(t):
    return _RoundBf16.apply(t)


class _RoundF16(torch.autograd.Function):
    """Round to IEEE-half storage precision in the forward, identity in the backward (the "f16" mode of the HIP path)."""

    @staticmethod
    def forward(ctx, t):
        return t.half().to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def round_f16(t):
    return _RoundF16.apply(t)


def _ident(t):
    return t


# ---- fp8 emulation of the "fp8" precision mode of the HIP path (stroke-prediction_amd/runtime/f8.py, csrc/sp_conv_zm8.hip):
# the MFMA operands of the forward and data-gradient convolution are OCP e4m3 / e5m2, everything else as in the bf16 mode.
def round_e4m3(t):
    """round to nearest even, saturating at +-448 (what v_cvt_pk_fp8_f32 behind a clamp does)"""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype)


def round_e5m2(t):
    return t.clamp(-57344.0, 57344.0).to(torch.float8_e5m2).to(t.dtype)


def quant_weights_e4m3(wf):
    """e4m3 weights with one power-of-two scale per output channel: 2^k = largest power of two with max|row| * 2^k <= 224
    (sp_conv_prep_f8); returns the de-quantised weights"""
    amax = wf.detach().abs().flatten(1).max(dim=1).values.clamp_min(1e-30)
    sc = torch.exp2(torch.floor(torch.log2(224.0 / amax))).view(-1, 1, 1, 1, 1)
    return round_e4m3(wf * sc) / sc


class _F8Conv(torch.autograd.Function):
    """y = conv(e4m3(x), e4m3(w)) + b;  dx = conv^T(e5m2(S dy) / S, e4m3(w));  dw from the un-quantised x and dy (the bf16
    weight-gradient kernels) or, with wgrad8, from the same fp8 copies: dw = sum e5m2(S dy) / S * e4m3(x)
    (csrc/sp_wgrad_f8.hip) -- through the folded weights this is also where the BatchNorm-backward sums of x come from."""

    @staticmethod
    def forward(ctx, x, wf, bf, grad_scale, wgrad8=False, fwd8=True, dgrad8=True, q=None):
        """fwd8 / dgrad8 False: that convolution stays the 16-bit one (q = the storage rounding of its weights) -- the "fp8b"
        mode keeps the bf16 forward and runs the data and weight gradients in fp8"""
        ctx.save_for_backward(x, wf)
        ctx.S = grad_scale
        ctx.wgrad8, ctx.dgrad8, ctx.q = bool(wgrad8), bool(dgrad8), q
        if fwd8:
            return F.conv3d(round_e4m3(x), quant_weights_e4m3(wf), bf)
        return F.conv3d(x, q(wf) if q is not None else wf, bf)

    @staticmethod
    def backward(ctx, g):
        x, wf = ctx.saved_tensors
        gq = round_e5m2(g * ctx.S) / ctx.S
        if ctx.dgrad8:
            # the data gradient packs the weights with one scale per INPUT channel of the convolution (its output channels)
            wq = quant_weights_e4m3(wf.transpose(0, 1).contiguous()).transpose(0, 1).contiguous()
            gx = torch.nn.grad.conv3d_input(x.shape, wq, gq)
        else:
            gx = torch.nn.grad.conv3d_input(x.shape, ctx.q(wf) if ctx.q is not None else wf, g)
        if ctx.wgrad8:
            gw = torch.nn.grad.conv3d_weight(round_e4m3(x), wf.shape, gq)
        else:
            gw = torch.nn.grad.conv3d_weight(x, wf.shape, g)
        return gx, gw, g.sum(dim=(0, 2, 3, 4)), None, None, None, None, None


class _SteE4m3(torch.autograd.Function):
    """a tensor that exists as its e4m3 copy only (the fp8 mode's block outputs that pooling / the skip crop read): every reader
    sees the rounded values -- BatchNorm statistics included --, the gradient passes straight through (the backward kernels form
    act'(y) and the pooling argmax from the same rounded values)"""

    @staticmethod
    def forward(ctx, x):
        return round_e4m3(x)

    @staticmethod
    def backward(ctx, g):
        return g


def f8_grad_scale(n_out_voxels):
    """runtime/f8.py:grad_scale_for"""
    return float(2.0 ** math.ceil(math.log2(64.0 * max(1, n_out_voxels))))


def _bn(sd, prefix, x, training):
    """``nn.BatchNorm3d`` call sites; training mode normalises with biased batch
    variance and updates running stats with the unbiased one."""
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training, BN_MOMENTUM, BN_EPS)


def center_crop(t, like, dims=(2, 3, 4)):
    """``crop`` Unet3D.py:6-11 -- offset (in-out)//2 per cropped dim."""
    for d in dims:
        t = t.narrow(d, (t.size(d) - like.size(d)) // 2, like.size(d))
    return t


def _bn_folded_conv(sd, bn_p, cv_p, x, training, q, f8=None):
    """The HIP bf16 path folds the BatchNorm of an un-padded convolution into weights and bias
    (conv(s*x+t) = conv_{W*s}(x) + sum W*t) so the raw bf16 input can be staged by LDS-DMA: the weights are
    rounded AFTER the fold, the input is not re-rounded.  Same function as BN -> conv, emulated here so that the
    bf16 parity test compares like with like."""
    w, b = sd[cv_p + ".weight"], sd[cv_p + ".bias"]
    gamma, beta = sd[bn_p + ".weight"], sd[bn_p + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3, 4))
        var = x.var(dim=(0, 2, 3, 4), unbiased=False)
        with torch.no_grad():
            n = x.numel() / x.shape[1]
            sd[bn_p + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            sd[bn_p + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var * n / (n - 1))
            sd[bn_p + ".num_batches_tracked"] += 1
    else:
        mean, var = sd[bn_p + ".running_mean"], sd[bn_p + ".running_var"]
    scale = gamma / torch.sqrt(var + BN_EPS)
    shift = beta - mean * scale
    bf = b + (w * shift.view(1, -1, 1, 1, 1)).sum(dim=(1, 2, 3, 4))
    if f8 is not None and cv_p in f8["layers"]:
        return _F8Conv.apply(x, w * scale.view(1, -1, 1, 1, 1), bf, f8["grad_scale"], cv_p in f8.get("wgrad", ()))
    if f8 is not None and (cv_p in f8.get("dgrad", ()) or cv_p in f8.get("wgrad", ())):      # "fp8b": fp8 backward only
        return _F8Conv.apply(x, w * scale.view(1, -1, 1, 1, 1), bf, f8["grad_scale"], cv_p in f8.get("wgrad", ()), False,
                             cv_p in f8.get("dgrad", ()), q)
    wf = q(w * scale.view(1, -1, 1, 1, 1))
    return F.conv3d(x, wf, bf)


def unet_block(sd, p, x, training, q=_ident, f8=None):
    """``Block3x3x3`` Unet3D.py:14-27: BN-conv(3,p0)-lrelu twice.
    ``q`` models the storage rounding of the HIP bf16 path (identity for the reference semantics):
    it is applied where that path rounds -- the normalised conv operand, the weights, the stored output."""
    for bn_i, cv_i in ((0, 1), (3, 4)):
        bn_p, cv_p = "%s.bn_conv_relu_2x.%d" % (p, bn_i), "%s.bn_conv_relu_2x.%d" % (p, cv_i)
        if q is not _ident:
            x = _bn_folded_conv(sd, bn_p, cv_p, x, training, q, f8)
        else:
            x = F.conv3d(_bn(sd, bn_p, x, training), sd[cv_p + ".weight"], sd[cv_p + ".bias"])
        x = q(F.leaky_relu(x, LEAKY))
    return x


def upsample2(x, align_corners=False):
    """``nn.Upsample(scale_factor=2, mode='trilinear')`` Unet3D.py:44,46.
    torch>=0.4 semantics (align_corners=False), as the importable reference."""
    return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=align_corners)


def unet_forward(sd, x, training=True, return_all=False, q=_ident, f8=None):
    """``Unet3D.forward`` Unet3D.py:56-79 (three scales) and ``LargeUnet3D.forward`` Unet3D.py:118-146 (four): the number
    of scales S follows from the block count of the state dict (2S - 1 blocks).  Returns sigmoid probs (B,2,...).
    ``q=round_bf16`` emulates the bf16 storage points of the HIP fast path (see ``unet_block``); ``f8=dict(layers={conv
    prefixes}, grad_scale=S)`` additionally runs those convolutions with fp8 operands (``_F8Conv``; needs q=round_bf16); the keys
    ``dgrad`` / ``wgrad`` name the layers whose data / weight gradient alone runs on fp8 operands (the "fp8b" mode: ``layers``
    empty, the forward stays the bf16 one); ``y2_e4m3`` = the down blocks whose output is stored as e4m3 only."""
    nblocks = len({k.split(".")[0] for k in sd if k.startswith("block")})
    S = (nblocks + 1) // 2
    outs = {}
    h = q(x)
    for i in range(1, S + 1):                              # down: block, pool (Unet3D.py:57-63 / :119-128)
        outs[i] = unet_block(sd, "block%d" % i, h, training, q, f8)
        if f8 is not None and i in f8.get("y2_e4m3", ()):      # this block's output lives as its e4m3 copy only (pool and skip crop read that)
            outs[i] = _SteE4m3.apply(outs[i])
        if i < S:
            h = F.max_pool3d(outs[i], 2, 2)
    low = outs[S]
    for u in range(S + 1, 2 * S):                          # up: upsample, crop + cat, block (Unet3D.py:64-73 / :129-141)
        up = q(upsample2(low))
        outs[u] = low = unet_block(sd, "block%d" % u, torch.cat((up, center_crop(outs[2 * S - u], up)), dim=1), training, q, f8)
    # (the HIP path fuses the classify head: weights enter as hi + lo bf16 pairs = fp32 accuracy; the hidden layer is
    # never stored but is rounded to bf16 as the operand of the second matrix product)
    h = q(F.leaky_relu(F.conv3d(low, sd["classify.0.weight"], sd["classify.0.bias"]), LEAKY))
    seg = torch.sigmoid(F.conv3d(h, sd["classify.2.weight"], sd["classify.2.bias"]))
    if return_all:
        return seg, {"b%d" % i: v for i, v in outs.items()}
    return seg


def batch_dice_loss(outputs, targets, label_weights=(1.0,), eps=1e-7, dim=1):
    """``BatchDiceLoss.forward`` metrics.py:16-28 (sums over the whole batch)."""
    assert targets.shape[dim] == len(label_weights)
    acc = 0.0
    for lab, w in enumerate(label_weights):
        o = outputs.narrow(dim, lab, 1).reshape(-1)
        t = targets.narrow(dim, lab, 1).reshape(-1)
        acc = acc + w * (2.0 * (o * t).sum() + eps) / ((o * o).sum() + (t * t).sum() + eps)
    return 1.0 - acc


def unet_loss(seg, labels):
    """``UnetSegmentationLearner.loss_step`` :21-28 -- mean of two Dice terms."""
    return (batch_dice_loss(seg[:, 0:1], labels[:, 0:1]) + batch_dice_loss(seg[:, 1:2], labels[:, 1:2])) / 2


# ----------------------------------------------------------------------------- CAE

def _cae_stack(sd, prefix, layers, x, alpha, training, last_sigmoid):
    n = len(layers)
    for i, (kind, _ci, _co, _k, s, p) in enumerate(layers):
        x = _bn(sd, "%s.%d" % (prefix, 3 * i), x, training)
        w, b = sd["%s.%d.weight" % (prefix, 3 * i + 1)], sd["%s.%d.bias" % (prefix, 3 * i + 1)]
        if kind == "conv":
            x = F.conv3d(x, w, b, stride=s, padding=p)
        else:
            x = F.conv_transpose3d(x, w, b, stride=s, padding=p)
        if last_sigmoid and i == n - 1:
            x = torch.sigmoid(x)
        else:
            x = F.elu(x, alpha)
    return x


def enc_forward(sd, x, alpha=1.0, training=True, prefix="encoder"):
    """``Enc3D.encoder`` Cae3D.py:39-76."""
    return _cae_stack(sd, prefix, ENC_LAYERS, x, alpha, training, False)


def dec_forward(sd, z, alpha=1.0, training=True, prefix="decoder"):
    """``Dec3D.decoder`` Cae3D.py:176-220."""
    return _cae_stack(sd, prefix, DEC_LAYERS, z, alpha, training, True)


def time_to_treatment(clinical, normalization_hours_penumbra=10):
    """``CaeInference.get_time_to_treatment`` CaeInference.py:18-31 (step=None):
    tA->tR / (10 - tO->tA), shape (B,1,1,1,1) float32."""
    c = clinical.float()
    return (c[:, 1] / (normalization_hours_penumbra - c[:, 0])).reshape(-1, 1, 1, 1, 1)


def cae_forward(sd, core, penu, lesion, step, alpha=1.0, training=True):
    """``Cae3D.forward`` Cae3D.py:248-251 = ``Enc3D.forward`` :100-118 (three
    encoder passes in the order core, penu, lesion; lerp :78-89) then
    ``Dec3D.forward`` :227-239 (four decoder passes core, penu, lesion, interp)."""
    lat = {}
    for k, v in (("core", core), ("penu", penu), ("lesion", lesion)):
        lat[k] = enc_forward(sd, v, alpha, training, "enc.encoder")
    lat["interpolation"] = lat["core"] + step * (lat["penu"] - lat["core"])
    rec = {}
    for k in ("core", "penu", "lesion", "interpolation"):
        rec[k] = dec_forward(sd, lat[k], alpha, training, "dec.decoder")
    return lat, rec


def cae_loss(lat, rec, core, penu, lesion, epoch):
    """``CaeReconstructionLearner.loss_step`` CaeReconstructionLearner.py:52-70."""
    factor = min(0.04 * max(0, epoch - 25), 1)
    d_pf = rec["penu"] - rec["interpolation"]
    d_pc = rec["penu"] - rec["core"]
    loss = torch.mean(torch.abs(d_pf) - d_pf) + torch.mean(torch.abs(d_pc) - d_pc)
    loss = loss + batch_dice_loss(rec["core"], core) + batch_dice_loss(rec["penu"], penu) \
        + batch_dice_loss(rec["lesion"], lesion)
    loss = loss + factor * torch.mean(torch.abs(lat["interpolation"] - lat["lesion"]))
    return loss / (5 + factor)


def cae_prediction_forward(sd_cae, sd_enc, unet_core, unet_penu, core, penu, lesion, step, alpha=1.0, training=True):
    """``CaeEncInference.inference_step`` CaeEncInference.py:29-42 with the branch selector the reference MEANS (it writes
    ``dto.mode``, the models read ``dto.flag``: as written the second call trips Cae3D.py:110): the new encoder on the two U-Net
    segmentations + lerp, the CAE's decoder on those three latents (``inputs`` branch); then the whole CAE on the manual masks
    (``gtruth`` branch).  Returns (latents_inputs, reconstructions_inputs, latents_gtruth, reconstructions_gtruth)."""
    lat_in = {"core": enc_forward(sd_enc, unet_core, alpha, training, "encoder"),
              "penu": enc_forward(sd_enc, unet_penu, alpha, training, "encoder")}
    lat_in["interpolation"] = lat_in["core"] + step * (lat_in["penu"] - lat_in["core"])
    rec_in = {k: dec_forward(sd_cae, lat_in[k], alpha, training, "dec.decoder") for k in ("core", "penu", "interpolation")}
    lat_gt, rec_gt = cae_forward(sd_cae, core, penu, lesion, step, alpha, training)
    return lat_in, rec_in, lat_gt, rec_gt


def cae_prediction_loss(lat_in, rec_in, lat_gt, lesion):
    """``CaePredictionLearner.loss_step`` CaePredictionLearner.py:42-57."""
    d_pf = rec_in["penu"] - rec_in["interpolation"]
    d_pc = rec_in["penu"] - rec_in["core"]
    loss = torch.mean(torch.abs(d_pf) - d_pf) + torch.mean(torch.abs(d_pc) - d_pc)
    loss = loss + batch_dice_loss(rec_in["interpolation"], lesion)
    for k in ("interpolation", "core", "penu"):
        loss = loss + torch.mean(torch.abs(lat_gt[k] - lat_in[k]))
    return loss / 6


def enc_step(sd, globals_, alpha=1.0, prefix="enc."):
    """``Enc3DStep._get_step`` Cae3D.py:137-141: sigmoid(step(reduce(globals))), reduce = conv1x1 -> ELU -> conv1x1 -> ELU."""
    h = F.elu(F.conv3d(globals_, sd[prefix + "reduce.0.weight"], sd[prefix + "reduce.0.bias"]), alpha)
    h = F.elu(F.conv3d(h, sd[prefix + "reduce.2.weight"], sd[prefix + "reduce.2.bias"]), alpha)
    return torch.sigmoid(F.conv3d(h, sd[prefix + "step.weight"], sd[prefix + "step.bias"]))


def cae_step_loss(rec, lesion):
    """``CaeStepLearner.loss_step`` CaeStepLearner.py:15-21."""
    d_pf = rec["penu"] - rec["interpolation"]
    return (torch.mean(torch.abs(d_pf) - d_pf) + batch_dice_loss(rec["interpolation"], lesion)) / 2


def cae_beta1(epoch, base=0.9, n_adapt=4):
    """``CaeReconstructionLearner.adapt_betas`` :28-40."""
    return base - 0.1 * (n_adapt - epoch) if epoch < n_adapt else base


# ----------------------------------------------------------------------------- optimiser

def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One ``torch.optim.Adam`` step (L2-coupled weight decay, no amsgrad) as used
    at train_unet_segmentation.py:32 / train_shape_reconstruction.py:40.
    ``step`` is the 1-based step count.  In-place on params / moments."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        if weight_decay != 0:
            g = g + weight_decay * p
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def trainable(sd):
    """Names that carry gradients (everything but BN buffers)."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]
