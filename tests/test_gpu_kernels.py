"""Kernel-level parity: each C-ABI entry point against plain torch (CPU, fp32/fp64) on the same
seeded inputs.  SP_F32 (split-bf16 x3 MFMA) is held to ~1e-4; SP_BF16 to bf16 rounding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from stroke_prediction_amd.runtime import lib as L
from stroke_prediction_amd.runtime import ops as O
from stroke_prediction_amd.runtime import plan as P

DEV = "cuda:0"
O.USE_PERSIST = 2          # small test volumes: take the persistent conv variant wherever it is eligible
BIAS_ATOL = {L.SP_F32: 1e-2, L.SP_BF16: 0.3}   # sums of O(1000) values; bf16 inputs carry 2^-9 relative noise
TOL = {L.SP_F32: dict(rtol=2e-4, atol=2e-4), L.SP_BF16: dict(rtol=3e-2, atol=3e-2)}


def to_cl(x, cp, dtype):
    """NCDHW fp32 (cpu) -> channels-last device tensor via the HIP kernel."""
    B, Cc = x.shape[:2]
    dst = O.alloc_cl(B, x.shape[2:], cp, dtype, DEV)
    O.ncdhw_to_cl(x.contiguous().to(DEV), dst, dtype)
    return dst


def from_cl(t, c, dtype):
    B = t.shape[0]
    out = torch.empty((B, c) + tuple(t.shape[1:4]), dtype=torch.float32, device=DEV)
    O.cl_to_ncdhw(t, out, dtype)
    return out.cpu()


def rnd(dtype, x):
    """round test inputs to what the storage dtype can hold, so both sides see the same numbers"""
    return x.bfloat16().float() if dtype == L.SP_BF16 else x


def test_layout_roundtrip():
    x = torch.randn(2, 3, 5, 6, 7)
    for dt in (L.SP_F32, L.SP_BF16):
        y = from_cl(to_cl(x, 8, dt), 3, dt)
        torch.testing.assert_close(y, rnd(dt, x), rtol=0, atol=0)


CONV_CASES = [
    # cin, cout, k, stride, pad, dims, batch
    (2, 16, 3, 1, (0, 0, 0), (12, 13, 37), 2),
    (16, 16, 3, 1, (0, 0, 0), (14, 20, 40), 2),
    (16, 16, 3, 1, (0, 0, 0), (23, 37, 50), 3),      # many tiles per persistent workgroup, ragged borders
    (16, 32, 3, 1, (0, 0, 0), (9, 11, 21), 1),
    (32, 64, 3, 1, (0, 0, 0), (8, 9, 19), 1),
    (96, 32, 3, 1, (0, 0, 0), (7, 10, 18), 1),
    (48, 16, 3, 1, (0, 0, 0), (7, 12, 35), 1),
    (16, 16, 3, 1, (1, 0, 0), (5, 12, 36), 2),
    (16, 24, 3, 2, (1, 1, 1), (8, 22, 38), 2),
    (32, 100, 3, 2, (0, 0, 0), (7, 25, 25), 1),
    (32, 24, 3, 1, (1, 2, 2), (4, 9, 17), 1),
    (16, 32, 1, 1, (0, 0, 0), (6, 7, 33), 2),
    (32, 2, 1, 1, (0, 0, 0), (6, 7, 33), 2),
]


@pytest.mark.parametrize("dtype", [L.SP_F32, L.SP_BF16])
@pytest.mark.parametrize("cin,cout,k,s,p,dims,B", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dtype, cin, cout, k, s, p, dims, B):
    g = torch.Generator().manual_seed(cin * 131 + cout)
    x = rnd(dtype, torch.randn(B, cin, *dims, generator=g))
    w = torch.randn(cout, cin, k, k, k, generator=g) / np.sqrt(cin * k ** 3)
    b = torch.randn(cout, generator=g) * 0.1
    scale = torch.rand(cin, generator=g) + 0.5
    shift = torch.randn(cin, generator=g) * 0.2
    cpi, cpo = O.cpad(cin), O.cpad(cout)
    wq = rnd(dtype, w)   # the bf16 path rounds weights to bf16 inside prep; compare against that
    xr = (x * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)).requires_grad_(True)
    if dtype == L.SP_BF16:
        xr_q = rnd(dtype, xr.detach()).requires_grad_(True)   # kernel rounds the normalised input to bf16
    else:
        xr_q = xr
    wr = wq.clone().requires_grad_(True)
    zref = F.conv3d(xr_q, wr, b, stride=s, padding=p)
    yref = F.leaky_relu(zref, 0.01)

    op = P.conv_fwd_op(cin, cout, k, s, p, dims, cpi, cpo, dtype)
    run = O.ConvRunner(op, DEV)
    run.prep(w.to(DEV), b.to(DEV))
    xs = to_cl(x, cpi, dtype)
    sc = torch.zeros(cpi, device=DEV); sc[:cin] = scale.to(DEV)
    sh = torch.zeros(cpi, device=DEV); sh[:cin] = shift.to(DEV)
    y = O.alloc_cl(B, op.y_dims, cpo, dtype, DEV)
    stats = torch.zeros(cpo, 2, dtype=torch.float64, device=DEV)
    run.run(xs, y, B, sc, sh, L.ACT_LEAKY, 0.01, stats)
    got = from_cl(y, cout, dtype)
    torch.testing.assert_close(got, yref.detach(), **TOL[dtype])
    # fused statistics describe what was stored
    st = stats.cpu()
    torch.testing.assert_close(st[:cout, 0], got.double().sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(st[:cout, 1], (got.double() ** 2).sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-3)
    if cpo > cout:
        assert float(y[..., cout:].float().abs().max()) == 0.0

    # ---- un-padded convolutions: BatchNorm folded into weights/bias + LDS-DMA staging (bf16 fast path)
    if dtype == L.SP_BF16 and max(p) == 0 and all(sub.tile["dma"] for sub in op.subs):
        run.prep(w.to(DEV), b.to(DEV), sc, sh)
        y2 = O.alloc_cl(B, op.y_dims, cpo, dtype, DEV)
        stats2 = torch.zeros(cpo, 2, dtype=torch.float64, device=DEV)
        run.run(xs, y2, B, None, None, L.ACT_LEAKY, 0.01, stats2)
        ref2 = F.leaky_relu(F.conv3d(x, rnd(dtype, w * scale.view(1, -1, 1, 1, 1)), None, stride=s, padding=p)
                            + (b + (w * shift.view(1, -1, 1, 1, 1)).sum(dim=(1, 2, 3, 4))).view(1, -1, 1, 1, 1), 0.01)
        got2 = from_cl(y2, cout, dtype)
        torch.testing.assert_close(got2, ref2, **TOL[dtype])
        torch.testing.assert_close(got2, yref.detach(), rtol=5e-2, atol=5e-2)     # and it is the same function
        torch.testing.assert_close(stats2.cpu()[:cout, 0], got2.double().sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-3)
        run.prep(w.to(DEV), b.to(DEV))

    # ---- data gradient
    dz = rnd(dtype, torch.randn(zref.shape, generator=g))
    gx_ref, gw_ref = torch.autograd.grad(zref, (xr_q, wr), dz)
    dop = P.conv_dgrad_op(cin, cout, k, s, p, dims, cpo, cpi, dtype)
    drun = O.ConvRunner(dop, DEV)
    drun.prep(w.to(DEV))
    dzs = to_cl(dz, cpo, dtype)
    gbuf = O.alloc_cl(B, dims, cpi, dtype, DEV, zero=True)
    drun.run(dzs, gbuf, B)
    torch.testing.assert_close(from_cl(gbuf, cin, dtype), gx_ref, **TOL[dtype])

    # ---- weight gradient (BatchNorm applied on load)
    wg = O.WgradRunner(cin, cout, k, s, p, dims, op.y_dims, cpi, cpo, cin * k ** 3, k ** 3, dtype, DEV)
    dw = torch.zeros_like(w, device=DEV)
    dbs = torch.zeros(cpo, dtype=torch.float64, device=DEV)
    dbs[:cout] = dz.double().sum(dim=(0, 2, 3, 4)).to(DEV)
    wg.run(xs, dzs, B, dw, sc, sh, dbias_sums=dbs)
    if wg.dma:   # folded BatchNorm: the reference sees the un-rounded normalised input
        gw_ref = torch.autograd.grad(F.conv3d(xr, wr, b, stride=s, padding=p), wr, dz)[0]
    scale_w = float(gw_ref.abs().max())
    torch.testing.assert_close(dw.cpu(), gw_ref, rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * max(1.0, scale_w))


CONVT_CASES = [
    (800, 100, 3, 1, 0, (1, 10, 10), 2),
    (104, 32, 3, 2, 0, (3, 12, 12), 2),
    (24, 24, 2, 2, 0, (7, 29, 29), 1),
    (16, 16, 2, 2, 0, (6, 20, 22), 2),
]


@pytest.mark.parametrize("dtype", [L.SP_F32, L.SP_BF16])
@pytest.mark.parametrize("cin,cout,k,s,p,dims,B", CONVT_CASES)
def test_convT_fwd_dgrad_wgrad(dtype, cin, cout, k, s, p, dims, B):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = rnd(dtype, torch.randn(B, cin, *dims, generator=g))
    w = torch.randn(cin, cout, k, k, k, generator=g) / np.sqrt(cin * k ** 3 / s ** 3)
    b = torch.randn(cout, generator=g) * 0.1
    cpi, cpo = O.cpad(cin), O.cpad(cout)
    xr = x.clone().requires_grad_(True)
    wr = rnd(dtype, w).requires_grad_(True)
    zref = F.conv_transpose3d(xr, wr, b, stride=s, padding=p)
    yref = F.elu(zref, 1.0)
    op = P.convT_fwd_op(cin, cout, k, s, p, dims, cpi, cpo, dtype)
    run = O.ConvRunner(op, DEV)
    run.prep(w.to(DEV), b.to(DEV))
    xs = to_cl(x, cpi, dtype)
    y = O.alloc_cl(B, op.y_dims, cpo, dtype, DEV)
    run.run(xs, y, B, None, None, L.ACT_ELU, 1.0, None)
    torch.testing.assert_close(from_cl(y, cout, dtype), yref.detach(), **TOL[dtype])
    dz = rnd(dtype, torch.randn(zref.shape, generator=g))
    gx_ref, gw_ref = torch.autograd.grad(zref, (xr, wr), dz)
    dop = P.convT_dgrad_op(cin, cout, k, s, p, dims, cpo, cpi, dtype)
    drun = O.ConvRunner(dop, DEV)
    drun.prep(w.to(DEV))
    dzs = to_cl(dz, cpo, dtype)
    gbuf = O.alloc_cl(B, dims, cpi, dtype, DEV, zero=True)
    drun.run(dzs, gbuf, B)
    torch.testing.assert_close(from_cl(gbuf, cin, dtype), gx_ref, **TOL[dtype])
    # weight gradient with swapped roles: shifted operand = dz (convT output grid), fixed operand = x
    kk = k ** 3
    wg = O.WgradRunner(cout, cin, k, s, p, op.y_dims, dims, cpo, cpi, cout * kk, kk, dtype, DEV)
    dw = torch.zeros_like(w, device=DEV)
    wg.run(dzs, xs, B, dw)
    scale_w = float(gw_ref.abs().max())
    torch.testing.assert_close(dw.cpu(), gw_ref, rtol=TOL[dtype]["rtol"], atol=TOL[dtype]["atol"] * max(1.0, scale_w))


@pytest.mark.parametrize("dtype", [L.SP_F32, L.SP_BF16])
def test_bn_pieces(dtype):
    g = torch.Generator().manual_seed(5)
    B, Cc, dims = 2, 24, (5, 6, 17)
    x = rnd(dtype, torch.randn(B, Cc, *dims, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    rm, rv = torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    cp = O.cpad(Cc)
    xs = to_cl(x, cp, dtype)
    sums = O.reduce_rows(cp, 2, DEV)         # SP_REDUCE_ROWS replica rows, added up by the finalize kernel
    O.bn_stats(xs, dtype, sums)
    n = B * int(np.prod(dims))
    scale, shift, mean, invstd = (torch.empty(cp, device=DEV) for _ in range(4))
    rm_d, rv_d = rm.clone().to(DEV), rv.clone().to(DEV)
    O.bn_finalize(sums, n, gamma.to(DEV), beta.to(DEV), rm_d, rv_d, 0.1, 1e-5, True, Cc, cp, scale, shift, mean, invstd,
                  nrep=L.SP_REDUCE_ROWS)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yref = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    y = x * scale[:Cc].cpu().view(1, -1, 1, 1, 1) + shift[:Cc].cpu().view(1, -1, 1, 1, 1)
    torch.testing.assert_close(y, yref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rm_d.cpu(), rm_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv_d.cpu(), rv_ref, rtol=1e-5, atol=1e-6)
    # backward: dx = coef0*g + coef1*x + coef2 ; then * act'(x) as the previous layer's dz
    gy = rnd(dtype, torch.randn(x.shape, generator=g))
    gx_ref, gg_ref, gb_ref = torch.autograd.grad(yref, (xr, gr, br), gy)
    gs = to_cl(gy, cp, dtype)
    bsums = O.reduce_rows(cp, 2, DEV)
    O.bn_bwd_reduce(gs, xs, dtype, bsums)
    dgam, dbet = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    coef = torch.empty(3, cp, device=DEV)
    O.bn_bwd_finalize(bsums, n, gamma.to(DEV), mean, invstd, Cc, cp, dgam, dbet, coef, nrep=L.SP_REDUCE_ROWS)
    torch.testing.assert_close(dgam.cpu(), gg_ref, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(dbet.cpu(), gb_ref, rtol=1e-3, atol=1e-2)
    dz = O.alloc_cl(B, dims, cp, dtype, DEV)
    dbias = O.reduce_rows(cp, 1, DEV)
    O.bn_act_bwd(gs, xs, coef, dtype, L.ACT_ELU, 1.0, dz, dbias)
    elu_d = torch.where(x > 0, torch.ones_like(x), x + 1.0)
    ref = gx_ref * elu_d
    got = from_cl(dz, Cc, dtype)
    torch.testing.assert_close(got, ref, **TOL[dtype])
    # the bias-gradient sums are taken before the storage rounding: compare with the exact reference
    torch.testing.assert_close(dbias.sum(0)[:Cc].cpu().float(), ref.sum(dim=(0, 2, 3, 4)), rtol=1e-3, atol=BIAS_ATOL[dtype])


@pytest.mark.parametrize("dtype", [L.SP_F32, L.SP_BF16])
def test_pool_upsample_crop_fwd_bwd(dtype):
    """The U-Net wiring kernels against autograd on the same small graph (Unet3D.py:59-72)."""
    g = torch.Generator().manual_seed(9)
    B, C1, dims = 2, 16, (12, 10, 14)          # block output y (post-LeakyReLU)
    ypre = rnd(dtype, torch.randn(B, C1, *dims, generator=g))
    ypre = torch.where(ypre > 0, ypre, 0.01 * ypre)
    ypre = rnd(dtype, ypre)
    y = ypre.clone().requires_grad_(True)
    ys = to_cl(ypre, C1, dtype)
    # forward: pool
    p_ref = F.max_pool3d(y, 2, 2)
    ps = O.alloc_cl(B, p_ref.shape[2:], C1, dtype, DEV)
    st = O.reduce_rows(C1, 2, DEV)
    O.maxpool2_fwd(ys, ps, dtype, st)
    st = st.sum(0)
    torch.testing.assert_close(from_cl(ps, C1, dtype), p_ref.detach(), rtol=0, atol=0)
    torch.testing.assert_close(st[:, 0].cpu(), p_ref.detach().double().sum(dim=(0, 2, 3, 4)), rtol=1e-5, atol=1e-4)
    # forward: low-res tensor upsampled into a concat buffer next to the cropped skip
    C0, ldims = 8, (4, 3, 5)
    low = rnd(dtype, torch.randn(B, C0, *ldims, generator=g))
    lowr = low.clone().requires_grad_(True)
    up_ref = F.interpolate(lowr, scale_factor=2, mode="trilinear", align_corners=False)
    cdims = tuple(up_ref.shape[2:])            # (8, 6, 10) <= dims
    off = [(dims[a] - cdims[a]) // 2 for a in range(3)]
    crop_ref = y[:, :, off[0]:off[0] + cdims[0], off[1]:off[1] + cdims[1], off[2]:off[2] + cdims[2]]
    cat_ref = torch.cat((up_ref, crop_ref), 1)
    cat = O.alloc_cl(B, cdims, C0 + C1, dtype, DEV)
    lows = to_cl(low, C0, dtype)
    O.upsample2_fwd(lows, cat, dtype)
    O.crop_copy(ys, cat, C0, dtype)
    torch.testing.assert_close(from_cl(cat, C0 + C1, dtype), cat_ref.detach(), **TOL[dtype])
    # the one-pass variant writes the same buffer bit for bit and accumulates the statistics of all its channels
    cat2 = torch.full_like(cat, 3.0)
    st2 = O.reduce_rows(C0 + C1, 2, DEV)
    O.upsample2_crop_cat_fwd(lows, ys, cat2, dtype, st2)
    st2 = st2.sum(0)
    torch.testing.assert_close(cat2.float(), cat.float(), **TOL[dtype])      # separable evaluation: last-bit differences
    cf = cat.double()
    torch.testing.assert_close(st2[:, 0].cpu(), cf.sum(dim=(0, 1, 2, 3)).cpu(), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(st2[:, 1].cpu(), (cf * cf).sum(dim=(0, 1, 2, 3)).cpu(), rtol=1e-5, atol=1e-3)
    # backward: arbitrary affine "BN backward" forms on both consumers
    cp_cat = C0 + C1
    coefs = torch.randn(3, cp_cat, generator=g) * 0.5
    coefp = torch.randn(3, C1, generator=g) * 0.5
    g_cat = rnd(dtype, torch.randn(cat_ref.shape, generator=g))
    g_p = rnd(dtype, torch.randn(p_ref.shape, generator=g))
    v = lambda t: t.view(1, -1, 1, 1, 1)
    catq = rnd(dtype, cat_ref.detach())
    d_cat = v(coefs[0]) * g_cat + v(coefs[1]) * catq + v(coefs[2])
    d_p = v(coefp[0]) * g_p + v(coefp[1]) * p_ref.detach() + v(coefp[2])
    gy_ref, glow_ref = torch.autograd.grad([cat_ref, p_ref], (y, lowr), [d_cat, d_p])
    lrelu_d = torch.where(ypre > 0, torch.ones_like(ypre), torch.full_like(ypre, 0.01))
    dz = O.alloc_cl(B, dims, C1, dtype, DEV)
    dbias = O.reduce_rows(C1, 1, DEV)
    gcs, gps = to_cl(g_cat, cp_cat, dtype), to_cl(g_p, C1, dtype)
    O.pool_skip_act_bwd(ys, gps, coefp.to(DEV), cat, gcs, coefs.to(DEV), C0, dtype, L.ACT_LEAKY, 0.01, dz, dbias)
    got = from_cl(dz, C1, dtype)
    torch.testing.assert_close(got, gy_ref * lrelu_d, **TOL[dtype])
    torch.testing.assert_close(dbias.sum(0).cpu().float(), (gy_ref * lrelu_d).sum(dim=(0, 2, 3, 4)), rtol=1e-3,
                               atol=BIAS_ATOL[dtype])
    # upsample backward lands on the low-res producer (here with ELU as its activation)
    dzl = O.alloc_cl(B, ldims, C0, dtype, DEV)
    O.upsample2_act_bwd(lows, cat, gcs, coefs.to(DEV), dtype, L.ACT_ELU, 1.0, dzl, None)
    elu_d = torch.where(low > 0, torch.ones_like(low), low + 1.0)
    torch.testing.assert_close(from_cl(dzl, C0, dtype), glow_ref * elu_d, **TOL[dtype])
    # the same two kernels fed with one DENSE gradient tensor per concat part (coefficients indexed separately)
    g_up, g_skip = to_cl(g_cat[:, :C0], C0, dtype), to_cl(g_cat[:, C0:], C1, dtype)
    dz2, dzl2 = torch.zeros_like(dz), torch.zeros_like(dzl)
    O.pool_skip_act_bwd(ys, gps, coefp.to(DEV), None, g_skip, coefs.to(DEV), 0, dtype, L.ACT_LEAKY, 0.01, dz2, None,
                        coef_c0=C0, coef_stride=cp_cat)
    O.upsample2_act_bwd(lows, None, g_up, coefs.to(DEV), dtype, L.ACT_ELU, 1.0, dzl2, None, coef_stride=cp_cat)
    assert torch.equal(dz2, dz) and torch.equal(dzl2, dzl)


def test_dice_and_output_grad():
    from stroke_prediction_amd.runtime import lib
    g = torch.Generator().manual_seed(3)
    B, Cc, dims = 2, 2, (6, 7, 9)
    o = torch.rand(B, Cc, *dims, generator=g)
    t = (torch.rand(B, Cc, *dims, generator=g) > 0.7).float()
    sums = torch.zeros(L.SP_REDUCE_ROWS, 16, dtype=torch.float64, device=DEV)      # replica rows of SP_DICE_PITCH(C) doubles
    od, td = o.to(DEV), t.to(DEV)
    dhw = int(np.prod(dims))
    lib.call("sp_dice_sums", O.ptr(od), Cc * dhw, O.ptr(td), Cc * dhw, B, Cc, dhw, O.ptr(sums), O.stream())
    ref = torch.stack([(o * t).sum(dim=(0, 2, 3, 4)), (o * o).sum(dim=(0, 2, 3, 4)), (t * t).sum(dim=(0, 2, 3, 4))], 1)
    torch.testing.assert_close(sums.sum(0)[:3 * Cc].view(Cc, 3).cpu().float(), ref, rtol=1e-5, atol=1e-4)
    # channel-slice views (batch stride = all channels) are read in place
    s1 = torch.zeros(L.SP_REDUCE_ROWS, 16, dtype=torch.float64, device=DEV)
    lib.call("sp_dice_sums", O.ptr(od[:, 1:2]), Cc * dhw, O.ptr(td[:, 1:2]), Cc * dhw, B, 1, dhw, O.ptr(s1), O.stream())
    torch.testing.assert_close(s1.sum(0)[:3].view(1, 3).cpu().float(), ref[1:2], rtol=1e-5, atol=1e-4)
    # finalize: loss and backward coefficients
    w = torch.tensor([0.3, 0.7])
    loss, coef = torch.empty((), device=DEV), torch.empty(2 * Cc, device=DEV)
    wd = w.to(DEV)
    lib.call("sp_dice_finalize", O.ptr(sums), O.ptr(wd), 1e-7, Cc, O.ptr(loss), O.ptr(coef), O.stream())
    num, den = 2 * ref[:, 0].double() + 1e-7, ref[:, 1].double() + ref[:, 2].double() + 1e-7
    assert abs(float(loss) - float(1 - (w.double() * num / den).sum())) < 1e-6
    ca, cb = (-2 * w.double() / den).float(), (2 * w.double() * num / den ** 2).float()
    torch.testing.assert_close(coef.cpu().view(Cc, 2), torch.stack([ca, cb], 1), rtol=1e-5, atol=1e-9)
    d = torch.empty_like(od)
    up = torch.tensor(0.5, device=DEV)
    lib.call("sp_dice_bwd", O.ptr(od), Cc * dhw, O.ptr(td), Cc * dhw, O.ptr(coef), O.ptr(up), B, Cc, dhw, O.ptr(d), O.stream())
    torch.testing.assert_close(d.cpu(), 0.5 * (ca.view(1, -1, 1, 1, 1) * t + cb.view(1, -1, 1, 1, 1) * o), rtol=1e-5, atol=1e-9)
    dz = O.alloc_cl(B, dims, 8, L.SP_F32, DEV)
    dbias = O.reduce_rows(8, 1, DEV)
    O.out_grad_to_cl(d, od, L.SP_F32, L.ACT_SIGMOID, 0.0, dz, dbias)
    ref = d.cpu() * o * (1 - o)
    torch.testing.assert_close(from_cl(dz, Cc, L.SP_F32), ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dbias.sum(0)[:Cc].cpu().float(), ref.sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-4)


def test_mean_of_channel_losses_fused_equals_literal():
    """(Dice(core) + Dice(penu)) / 2 on channel-slice views (UnetSegmentationLearner.py:21-28): the one-launch evaluation
    on the base tensors against the literal two calls, value and gradient."""
    from stroke_prediction_amd.common.metrics import BatchDiceLoss, mean_of_channel_losses, _stacked_base
    g = torch.Generator().manual_seed(5)
    B, dims = 3, (5, 6, 7)
    seg = torch.rand(B, 2, *dims, generator=g).to(DEV).requires_grad_(True)
    lab = (torch.rand(B, 2, *dims, generator=g) > 0.6).float().to(DEV)
    crit = BatchDiceLoss([1.0])

    def views(s):
        return s[:, 0, :, :, :].unsqueeze(1), s[:, 1, :, :, :].unsqueeze(1)       # Unet3D.forward :76-77
    s2 = seg * 1.0                                   # non-leaf, like the network output
    outs, tgts = views(s2), (lab[:, 0:1], lab[:, 1:2])
    assert _stacked_base(outs) is s2 and _stacked_base(tgts) is lab
    fused = mean_of_channel_losses(crit, outs, tgts)
    gf, = torch.autograd.grad(fused, seg)
    s3 = seg * 1.0
    o3 = views(s3)
    lit = (crit(o3[0], tgts[0]) + crit(o3[1], tgts[1])) / 2
    gl, = torch.autograd.grad(lit, seg)
    assert abs(float(fused) - float(lit)) < 1e-6
    torch.testing.assert_close(gf, gl, rtol=1e-5, atol=1e-9)
    # not slices of one tensor: falls back to the literal sum
    other = torch.rand(B, 1, *dims, generator=g).to(DEV)
    fb = mean_of_channel_losses(crit, (outs[0], other), tgts)
    assert abs(float(fb) - float((crit(outs[0], tgts[0]) + crit(other, tgts[1])) / 2)) < 1e-6


@pytest.mark.parametrize("shape", [(2, 1, 12, 14, 10), (3, 1, 9, 9, 9), (20, 17, 9), (1, 2, 6, 7, 8), (40,)])
def test_surface_distances_match_scipy(shape):
    """sp_surface_distances (Hausdorff / ASSD, medpy semantics incl. the 5-D structure the reference's call implies) against
    the oracle's restatement of medpy (oracle/measures.py, pinned by tests/test_measures_oracle.py)."""
    from stroke_prediction_amd.common import metrics as M
    from oracle import measures as OM
    g = torch.Generator().manual_seed(sum(shape))
    for kind in ("blobs", "noise", "single"):
        if kind == "noise":
            a, b = torch.rand(shape, generator=g), torch.rand(shape, generator=g)
        elif kind == "single":
            a, b = torch.zeros(shape), torch.zeros(shape)
            a.view(-1)[3] = 1.0
            b.view(-1)[a.numel() - 2] = 1.0
        else:                                      # smooth blobs: thick objects with real interiors
            a, b = torch.rand(shape, generator=g), torch.rand(shape, generator=g)
            for _ in range(2):
                for d in range(len(shape)):
                    if shape[d] > 2:
                        a = (a + a.roll(1, d) + a.roll(-1, d)) / 3
                        b = (b + b.roll(1, d) + b.roll(-1, d)) / 3
            a, b = (a - a.mean()) * 20 + 0.5, (b - b.mean()) * 20 + 0.5
        an, bn = a.numpy() > 0.5, b.numpy() > 0.5
        if not (an.any() and bn.any()):
            continue
        hd_ref, assd_ref = OM.hd(an, bn), OM.assd(an, bn)
        hd, assd = M._surface_metrics_device(a.to(DEV).contiguous(), b.to(DEV).contiguous(), 0.5)
        assert abs(hd - hd_ref) <= 1e-5 * max(1.0, hd_ref), (kind, hd, hd_ref)
        assert abs(assd - assd_ref) <= 1e-5 * max(1.0, assd_ref), (kind, assd, assd_ref)
    # through the public entry point (metrics.py:48-62): identical measures, distances included
    r = torch.rand((2, 1, 10, 11, 12), generator=g)
    t = (torch.rand((2, 1, 10, 11, 12), generator=g) > 0.6).float()
    dev = M.binary_measures_torch(r.to(DEV), t.to(DEV), True, distances=True)
    ref = OM.binary_measures(r.numpy(), t.numpy())
    for f in ("dc", "hd", "assd", "precision", "sensitivity", "specificity"):
        assert abs(getattr(dev, f) - ref[f]) <= 1e-5 * max(1.0, abs(ref[f])), f
    # host arrays through the reference's numpy entry point: uploaded, same device kernels
    up = M.binary_measures_numpy(r.numpy(), t.numpy())
    assert up.dc == dev.dc and up.hd == dev.hd
    # the hand-computed known answers of tests/test_measures_oracle.py on the device
    import math
    a = torch.zeros(1, 1, 7, 7, 7); b = torch.zeros(1, 1, 7, 7, 7)
    a[0, 0, 3, 3, 3] = 1; b[0, 0, 2:5, 2:5, 2:5] = 1
    hd, assd = M._surface_metrics_device(a.to(DEV), b.to(DEV), 0.5)
    assert abs(hd - math.sqrt(3)) < 1e-6 and abs(assd - 0.5 * (6 + 12 * math.sqrt(2) + 8 * math.sqrt(3)) / 27) < 1e-6
    hd, assd = M._surface_metrics_device(a[0, 0].contiguous().to(DEV), b[0, 0].contiguous().to(DEV), 0.5)
    assert abs(hd - math.sqrt(3)) < 1e-6 and abs(assd - 0.5 * (1 + (6 + 12 * math.sqrt(2) + 8 * math.sqrt(3)) / 26)) < 1e-6


def test_adam_matches_torch():
    g = torch.Generator().manual_seed(4)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 5):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        O.adam_step_flat(p, gr.to(DEV), m, v, 1e-3, 0.99, 0.999, 1e-8, 1e-5, step)
    torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [L.SP_F32, L.SP_BF16])
@pytest.mark.parametrize("C,CH,NC,CP", [(16, 32, 2, 16), (16, 16, 2, 16), (16, 32, 1, 16), (32, 32, 2, 32), (32, 32, 2, 48)])
def test_fused_head(dtype, C, CH, NC, CP):
    """sp_head_fwd / sp_head_bwd against autograd of the classify Sequential (Unet3D.py:49-54); 32 input channels (the head
    of the 4-scale network) on the MFMA kernels only."""
    if C == 32 and dtype == L.SP_F32:
        assert L.load().sp_head_supported_dtype(C, CH, NC, dtype) == 0
        return
    assert L.load().sp_head_supported_dtype(C, CH, NC, dtype) == 1
    assert L.load().sp_head_supported(24, 32, 2) == 0
    torch.manual_seed(5)
    B, dims = 2, (5, 9, 13)                      # 585 voxels / sample: ragged against the 256-voxel blocks
    nv = dims[0] * dims[1] * dims[2]
    y = rnd(dtype, F.leaky_relu(torch.randn(B, C, *dims), 0.01)).requires_grad_(True)   # the producing conv's output
    w1 = (torch.randn(CH, C) * 0.3).requires_grad_(True)
    b1 = (torch.randn(CH) * 0.1).requires_grad_(True)
    w2 = (torch.randn(NC, CH) * 0.3).requires_grad_(True)
    b2 = (torch.randn(NC) * 0.1).requires_grad_(True)
    h = F.leaky_relu(F.conv3d(y, w1.view(CH, C, 1, 1, 1), b1), 0.01)
    seg_ref = torch.sigmoid(F.conv3d(h, w2.view(NC, CH, 1, 1, 1), b2))
    dseg = torch.randn_like(seg_ref)
    seg_ref.backward(dseg)
    # reference dz: dL/dy times LeakyReLU'(pre-activation), recovered from the sign of y
    dz_ref = y.grad * torch.where(y.detach() > 0, torch.tensor(1.0), torch.tensor(0.01))

    x_cl = to_cl(y.detach(), CP, dtype)
    dv = lambda t: t.detach().to(DEV).contiguous()
    W1, B1, W2, B2 = dv(w1), dv(b1), dv(w2), dv(b2)
    seg = torch.empty((B, NC) + dims, dtype=torch.float32, device=DEV)
    L.call("sp_head_fwd", O.ptr(x_cl), dtype, nv, B, CP, C, O.ptr(W1), O.ptr(B1), CH, O.ptr(W2), O.ptr(B2), NC, 0.01,
           O.ptr(seg), O.stream())
    # bf16 storage: the hidden layer is the bf16 operand of the second matrix product (2^-9 relative per element)
    torch.testing.assert_close(seg.cpu(), seg_ref.detach(), rtol=1e-5, atol=1e-5 if dtype == L.SP_F32 else (3e-3 if C == 16 else 5e-3))

    dz = torch.full_like(x_cl, 7.0)
    dbs = torch.zeros(CP, dtype=torch.float64, device=DEV)
    lib = L.load()
    rows, nq = lib.sp_head_bwd_rows(B * nv), lib.sp_head_row_floats(C, CH, NC)
    assert nq == CH * C + CH + NC * CH + NC + C and rows == (B * nv + 255) // 256
    part = torch.full((rows * nq,), float("nan"), dtype=torch.float32, device=DEV)     # must be fully overwritten
    DS = dv(dseg)
    L.call("sp_head_bwd", O.ptr(x_cl), dtype, nv, B, CP, C, O.ptr(W1), O.ptr(B1), CH, O.ptr(W2), NC, 0.01, O.ptr(seg),
           O.ptr(DS), L.ACT_LEAKY, 0.01, O.ptr(dz), O.ptr(part), O.stream())
    gs = [torch.ones(CH * C, device=DEV), torch.ones(CH, device=DEV), torch.ones(NC * CH, device=DEV), torch.ones(NC, device=DEV)]
    L.call("sp_head_grad_finish", O.ptr(part), rows, C, CH, NC, O.ptr(gs[0]), O.ptr(gs[1]), O.ptr(gs[2]), O.ptr(gs[3]),
           O.ptr(dbs), O.stream())
    hg = torch.cat(gs) - 1.0                       # the finish kernel accumulates (+=) into the gradients
    torch.testing.assert_close(from_cl(dz, C, dtype), dz_ref, **TOL[dtype])
    torch.testing.assert_close(dbs[:C].cpu().float(), dz_ref.sum((0, 2, 3, 4)), rtol=1e-3, atol=BIAS_ATOL[dtype])
    hg = hg.cpu().float()
    o = 0
    for ref in (w1.grad, b1.grad, w2.grad, b2.grad):
        n = ref.numel()
        # bf16 storage: the in-kernel GEMMs see bf16-rounded dh/h/do factors (2^-9 each, random sign over 1170 voxels)
        tol = dict(rtol=1e-4, atol=1e-3) if dtype == L.SP_F32 else dict(rtol=1e-2, atol=0.02 * float(ref.abs().max()) + 1e-3)
        torch.testing.assert_close(hg[o:o + n].view(ref.shape), ref, **tol)
        o += n


def test_bn_bwd_sums_from_wgrad():
    """First-layer shortcut: sum_v g and sum_v g*x (g = conv_transpose(dz, W)) out of the weight-gradient accumulator
    (sp_wgrad_finish_folded with bn_sums) against the explicit data-gradient convolution."""
    torch.manual_seed(11)
    B, cin, cout, dims = 2, 2, 16, (12, 14, 40)
    od = tuple(d - 2 for d in dims)
    dt = L.SP_BF16
    x = rnd(dt, torch.randn(B, cin, *dims))
    dz = rnd(dt, torch.randn(B, cout, *od))
    w = torch.randn(cout, cin, 3, 3, 3) * 0.2
    scale, shift = torch.rand(16) + 0.5, torch.randn(16) * 0.1
    g = F.conv_transpose3d(dz.double(), w.double())
    ref = torch.stack([g.sum((0, 2, 3, 4)), (g * x.double()).sum((0, 2, 3, 4))], 1)          # [cin][2]
    xn = x * scale[:cin].view(1, -1, 1, 1, 1) + shift[:cin].view(1, -1, 1, 1, 1)
    wr = w.clone().requires_grad_(True)
    F.conv3d(xn, wr).backward(dz)
    x_cl, dz_cl = to_cl(x, 16, dt), to_cl(dz, 16, dt)
    wg = O.WgradRunner(cin, cout, 3, 1, 0, dims, od, 16, 16, cin * 27, 27, dt, DEV)
    assert wg.dma and wg.folds(scale)
    dw = torch.zeros_like(w, device=DEV)
    db = torch.zeros(cout, device=DEV)
    dbs = torch.zeros(16, dtype=torch.float64, device=DEV)
    dbs[:cout] = dz.double().sum((0, 2, 3, 4)).to(DEV)
    nrep = 64
    bs = torch.zeros(nrep, 16, 2, dtype=torch.float64, device=DEV)
    wd = w.to(DEV)
    wg.run(x_cl, dz_cl, B, dw, scale.to(DEV), shift.to(DEV), dbias_sums=dbs, dbias_grad=db, nbias=cout,
           bn_w=wd, bn_sums=bs, bn_nrep=nrep)
    got = bs.sum(0).cpu()
    torch.testing.assert_close(got[:cin], ref, rtol=1e-4, atol=1e-2)
    assert float(got[cin:].abs().max()) == 0.0
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(db.cpu(), dz.sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [L.SP_F32, L.SP_BF16])
@pytest.mark.parametrize("C0,ldims", [(32, (9, 7, 21)), (64, (5, 6, 9)), (16, (3, 17, 18)), (128, (6, 5, 19)), (256, (4, 3, 17))])
def test_upsample2_act_bwd_tiled(dtype, C0, ldims):
    """sp_upsample2_act_bwd (tiled z-marching kernel; several patches, z chunks and ragged edges; bf16: the LDS-DMA ring
    kernel, 128 / 256 channels as 2 / 4 groups of 64) against autograd of
    F.interpolate(trilinear x2) composed with the BatchNorm-backward affine form (Unet3D.py:67-72 backward)."""
    g = torch.Generator().manual_seed(21)
    B, C1 = 2, 16
    low = rnd(dtype, torch.randn(B, C0, *ldims, generator=g))
    lowr = low.clone().requires_grad_(True)
    up_ref = F.interpolate(lowr, scale_factor=2, mode="trilinear", align_corners=False)
    cdims = tuple(up_ref.shape[2:])
    cp_cat = C0 + C1
    coefs = torch.randn(3, cp_cat, generator=g) * 0.5
    g_cat = rnd(dtype, torch.randn(B, cp_cat, *cdims, generator=g))
    v = lambda t: t.view(1, -1, 1, 1, 1)
    d_up = v(coefs[0, :C0]) * g_cat[:, :C0] + v(coefs[1, :C0]) * up_ref.detach() + v(coefs[2, :C0])
    (glow_ref,) = torch.autograd.grad([up_ref], (lowr,), [d_up])
    lows = to_cl(low, C0, dtype)
    cat = O.alloc_cl(B, cdims, cp_cat, dtype, DEV)
    O.upsample2_fwd(lows, cat, dtype)
    gcs = to_cl(g_cat, cp_cat, dtype)
    dzl = O.alloc_cl(B, ldims, C0, dtype, DEV)
    dbias = O.reduce_rows(C0, 1, DEV)
    O.upsample2_act_bwd(lows, cat, gcs, coefs.to(DEV), dtype, L.ACT_ELU, 1.0, dzl, dbias)
    elu_d = torch.where(low > 0, torch.ones_like(low), low + 1.0)
    want = glow_ref * elu_d
    torch.testing.assert_close(from_cl(dzl, C0, dtype), want, **TOL[dtype])
    torch.testing.assert_close(dbias.sum(0).cpu().float(), want.sum(dim=(0, 2, 3, 4)), rtol=1e-3, atol=BIAS_ATOL[dtype])


@pytest.mark.parametrize("CO", [16, 32])
def test_first_layer_packed(CO):
    """sp_first.hip: BatchNorm-folded Conv3d(2, 16 | 32, 3)+LeakyReLU, its statistics (and the e4m3 copy of the output), the
    input statistics and the weight gradient (+ BatchNorm-backward sums) straight from the NCDHW input, against torch on
    bf16-rounded operands."""
    torch.manual_seed(17)
    lib = L.load()
    assert lib.sp_first_supported(2, CO, 3) == 1 and lib.sp_first_supported(3, 16, 3) == 0 and lib.sp_first_supported(2, 48, 3) == 0
    B, dims = 2, (9, 11, 70)                      # ragged against the 2 x 4 x 64 tile, two x tiles
    od = tuple(d - 2 for d in dims)
    x = torch.randn(B, 2, *dims) * 1.5 + 0.3
    xq = x.bfloat16().float()
    w = torch.randn(CO, 2, 3, 3, 3) * 0.2
    b = torch.randn(CO) * 0.1
    scale, shift = torch.rand(16) + 0.5, torch.randn(16) * 0.1
    xd = x.to(DEV).contiguous()
    nrep = 64
    # input statistics
    sums = torch.zeros(nrep, 16, 2, dtype=torch.float64, device=DEV)
    L.call("sp_bn_stats_ncdhw", O.ptr(xd), B, 2, dims[0] * dims[1] * dims[2], 16, O.ptr(sums), nrep, O.stream())
    got = sums.sum(0).cpu()
    ref = torch.stack([xq.double().sum((0, 2, 3, 4)), (xq.double() ** 2).sum((0, 2, 3, 4))], 1)
    torch.testing.assert_close(got[:2], ref, rtol=1e-6, atol=1e-3)
    # forward
    wfrag = torch.zeros((CO // 16) * 3 * 64 * 8, dtype=torch.bfloat16, device=DEV)
    bias_f = torch.zeros(CO, device=DEV)
    wd, bd, sc, sh = w.to(DEV), b.to(DEV), scale.to(DEV), shift.to(DEV)
    L.call("sp_first_prep_n", O.ptr(wd), O.ptr(bd), O.ptr(sc), O.ptr(sh), O.ptr(wfrag), O.ptr(bias_f), CO, O.stream())
    y = O.alloc_cl(B, od, CO, L.SP_BF16, DEV)
    st = torch.zeros(nrep, CO, 2, dtype=torch.float64, device=DEV)
    from stroke_prediction_amd.runtime import f8 as F8
    y8 = F8.alloc_f8(B, od, CO, DEV)
    L.call("sp_first_conv_fwd_n", O.ptr(xd), B, dims[0], dims[1], dims[2], O.ptr(wfrag), O.ptr(bias_f), L.ACT_LEAKY, 0.01,
           O.ptr(y), O.ptr(st), nrep, CO, O.ptr(y8), y8[0].numel(), O.stream())
    r8 = F8.alloc_f8(B, od, CO, DEV)
    F8.quantize(y, r8, F8.E4M3, 1.0)
    assert torch.equal(y8, r8)                     # the e4m3 copy == sp_quantize_f8 of the stored output
    wf = (w * scale[:2].view(1, 2, 1, 1, 1)).bfloat16().float()
    bf = b + (w * shift[:2].view(1, 2, 1, 1, 1)).sum((1, 2, 3, 4))
    y_ref = F.leaky_relu(F.conv3d(xq, wf, bf), 0.01)
    yg = from_cl(y, CO, L.SP_BF16)
    torch.testing.assert_close(yg, y_ref, rtol=1e-2, atol=1e-2)
    sg = st.sum(0).cpu()
    torch.testing.assert_close(sg[:, 0].float(), yg.sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(sg[:, 1].float(), (yg ** 2).sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-2)
    # weight gradient + BatchNorm-backward sums
    dz = rnd(L.SP_BF16, torch.randn(B, CO, *od))
    dz_cl = to_cl(dz, CO, L.SP_BF16)
    nparts = 24
    part = torch.full((nparts * 27 * CO * 2,), float("nan"), device=DEV)
    L.call("sp_first_wgrad_n", O.ptr(xd), O.ptr(dz_cl), B, dims[0], dims[1], dims[2], O.ptr(part), nparts, CO, O.stream())
    dw = torch.zeros(CO, 2, 3, 3, 3, device=DEV)
    db = torch.zeros(CO, device=DEV)
    dbs = torch.zeros(CO, dtype=torch.float64, device=DEV)
    dbs[:] = dz.double().sum((0, 2, 3, 4)).to(DEV)
    bs = torch.zeros(nrep, 16, 2, dtype=torch.float64, device=DEV)
    tapsrc = torch.arange(27, dtype=torch.int32, device=DEV)
    L.call("sp_wgrad_finish_folded", O.ptr(part), nparts, O.ptr(tapsrc), 27, CO, 2, CO, 2, 54, 27, O.ptr(sc), O.ptr(sh),
           O.ptr(dbs), O.ptr(dw), O.ptr(db), O.ptr(wd), O.ptr(bs), nrep, 16, 0, O.stream())
    xn = xq * scale[:2].view(1, 2, 1, 1, 1) + shift[:2].view(1, 2, 1, 1, 1)
    wr = w.clone().requires_grad_(True)
    F.conv3d(xn, wr).backward(dz)
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(db.cpu(), dz.sum((0, 2, 3, 4)), rtol=1e-4, atol=1e-3)
    g = F.conv_transpose3d(dz.double(), w.double())
    ref = torch.stack([g.sum((0, 2, 3, 4)), (g * xq.double()).sum((0, 2, 3, 4))], 1)
    got = bs.sum(0).cpu()
    torch.testing.assert_close(got[:2], ref, rtol=1e-4, atol=1e-2)
    assert float(got[2:].abs().max()) == 0.0
    # fused variant: dz = (c0*g + c1*y + c2) * LeakyReLU'(y) formed inside the kernel == sp_bn_act_bwd followed by the above
    gq = rnd(L.SP_BF16, torch.randn(B, CO, *od))
    coef = torch.randn(3, CO) * 0.5
    g_cl = to_cl(gq, CO, L.SP_BF16)
    dz2 = torch.empty_like(dz_cl)
    dbs2 = O.reduce_rows(CO, 1, DEV)
    O.bn_act_bwd(g_cl, y, coef.to(DEV), L.SP_BF16, L.ACT_LEAKY, 0.01, dz2, dbs2)
    p_ref = torch.empty(nparts * 27 * CO * 2, device=DEV)
    L.call("sp_first_wgrad_n", O.ptr(xd), O.ptr(dz2), B, dims[0], dims[1], dims[2], O.ptr(p_ref), nparts, CO, O.stream())
    p_fus = torch.full_like(p_ref, float("nan"))
    dbs3 = O.reduce_rows(CO, 1, DEV)
    cd = coef.to(DEV)
    L.call("sp_first_wgrad_fused_n", O.ptr(xd), O.ptr(g_cl), O.ptr(y), O.ptr(cd), L.ACT_LEAKY, 0.01, B, dims[0], dims[1], dims[2],
           O.ptr(p_fus), nparts, O.ptr(dbs3), CO, O.stream())
    torch.testing.assert_close(p_fus.view(nparts, -1).sum(0), p_ref.view(nparts, -1).sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dbs3.sum(0), dbs2.sum(0), rtol=1e-5, atol=1e-4)
    # fp8 mode: the 16-bit output is not stored at all (y = NULL: same e4m3 copy, same statistics) and the fused weight gradient
    # takes y as that copy -- bit for bit what the 16-bit kernel makes of the de-quantised e4m3 values
    y8b = F8.alloc_f8(B, od, CO, DEV)
    st2 = torch.zeros(nrep, CO, 2, dtype=torch.float64, device=DEV)
    L.call("sp_first_conv_fwd_n", O.ptr(xd), B, dims[0], dims[1], dims[2], O.ptr(wfrag), O.ptr(bias_f), L.ACT_LEAKY, 0.01, None,
           O.ptr(st2), nrep, CO, O.ptr(y8b), y8b[0].numel(), O.stream())
    assert torch.equal(y8b, y8)
    torch.testing.assert_close(st2.sum(0), st.sum(0), rtol=1e-12, atol=1e-9)
    yq8 = y8.view(torch.float8_e4m3fn).float()                      # (P, B, D, H, W, 16) -> channels-last bf16 (exact: 3 mantissa bits)
    y_deq = yq8.permute(1, 2, 3, 4, 0, 5).reshape(B, *od, CO).to(torch.bfloat16).contiguous()
    p_a, p_b = torch.full_like(p_ref, float("nan")), torch.full_like(p_ref, float("nan"))
    dbs_a, dbs_b = O.reduce_rows(CO, 1, DEV), O.reduce_rows(CO, 1, DEV)
    L.call("sp_first_wgrad_fused_n", O.ptr(xd), O.ptr(g_cl), O.ptr(y_deq), O.ptr(cd), L.ACT_LEAKY, 0.01, B, dims[0], dims[1], dims[2],
           O.ptr(p_a), nparts, O.ptr(dbs_a), CO, O.stream())
    L.call("sp_first_wgrad_fused_y8", O.ptr(xd), O.ptr(g_cl), O.ptr(y8), y8[0].numel(), O.ptr(cd), L.ACT_LEAKY, 0.01, B, dims[0], dims[1],
           dims[2], O.ptr(p_b), nparts, O.ptr(dbs_b), CO, O.stream())
    assert torch.equal(p_a, p_b) and float(p_b.abs().sum()) > 0
    torch.testing.assert_close(dbs_b.sum(0), dbs_a.sum(0), rtol=1e-12, atol=1e-9)
