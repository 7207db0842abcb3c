"""Round-4 GPU tests of the CAE's batched passes (one launch per layer over all BatchNorm groups):

* csrc/sp_conv_zm.hip: per-group output statistics inside one launch (``group_batch``); the data gradient with the
  BatchNorm-backward sums in its epilogue (``stats_mode = 1``); the BatchNorm folded per group into a PADDED convolution
  (per-group fragments + a bias table over the border classes: ``sp_conv_prep_folded_groups``);
* csrc/sp_conv_par.hip: all parity classes of a transposed / strided-gradient convolution -- and strided convolutions -- in one pass;
* csrc/sp_wgrad_dma.hip: stride-2 / 2x2x2 weight gradients on the LDS-DMA kernel; the weight gradient of the padded layers on the
  RAW input (border-class sums of dz from ``sp_bn_act_bwd_groups_cls``, group-pure partial blocks, ``sp_wgrad_finish_folded_groups``),
  and the same for the pointwise layer (csrc/sp_conv_fc.hip with per-group scale / shift on the operand load, csrc/sp_wgrad_pw.hip);
* the reconstruction loss as three launches (``metrics.cae_reconstruction_loss``);
* the whole CAE step with the grouped z-march on and off.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from stroke_prediction_amd.runtime import lib as L
from stroke_prediction_amd.runtime import ops as O
from stroke_prediction_amd.runtime import plan as P

DEV = "cuda:0"


def bf(t):
    return t.bfloat16().float()


def _to_cl(x, cp):
    dst = O.alloc_cl(x.shape[0], x.shape[2:], cp, L.SP_BF16, DEV)
    O.ncdhw_to_cl(x.contiguous().to(DEV), dst, L.SP_BF16)
    return dst


def _from_cl(t, c):
    out = torch.empty((t.shape[0], c) + tuple(t.shape[1:4]), dtype=torch.float32, device=DEV)
    O.cl_to_ncdhw(t, out, L.SP_BF16)
    return out.cpu()


# cin, cout, input dims, padding, batch, group batch -- (P, NT) = (1,1) (2,2) (2,1) (1,2); volumes small enough that one workgroup's
# piece of the march crosses sample and group boundaries, and ragged in every direction
GROUP_CASES = [(16, 16, (9, 36, 40), (1, 0, 0), 6, 2), (24, 24, (6, 34, 36), (1, 2, 2), 4, 1),
               (24, 16, (5, 33, 20), (1, 2, 2), 6, 3), (16, 24, (4, 40, 17), (1, 1, 1), 4, 2)]


@pytest.mark.parametrize("cin,cout,dims,pad,B,gb", GROUP_CASES)
def test_z_marching_forward_statistics_per_batchnorm_group(cin, cout, dims, pad, B, gb, monkeypatch):
    monkeypatch.setattr(O, "ZM_MIN_PLANES", 0)
    g = torch.Generator().manual_seed(cin * 5 + cout + B)
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    x = bf(torch.randn(B, cin, *dims, generator=g) * (1.0 + torch.arange(B).view(B, 1, 1, 1, 1)))      # groups of different scale
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / math.sqrt(27 * cin)
    b = torch.randn(cout, generator=g) * 0.1
    op = P.conv_fwd_op(cin, cout, 3, 1, pad, dims, cpi, cpo, L.SP_BF16)
    run = O.ConvRunner(op, DEV, zm_batch=B)
    assert run.uses_zm()
    run.prep(w.to(DEV), b.to(DEV))
    xs = _to_cl(x, cpi)
    nrep, G = 4, B // gb
    y = O.alloc_cl(B, op.y_dims, cpo, L.SP_BF16, DEV)
    stats = torch.zeros(G * nrep * cpo * 2, dtype=torch.float64, device=DEV)
    run.run(xs, y, B, None, None, L.ACT_ELU, 1.0, stats, stats_nrep=nrep, group_batch=gb)
    got = _from_cl(y, cout)
    torch.testing.assert_close(got, F.elu(F.conv3d(x, bf(w), b, padding=pad), 1.0), rtol=3e-2, atol=3e-2)
    st = stats.view(G, nrep, cpo, 2).sum(1).cpu()
    nvox = got[:gb].numel() / cout
    for gi in range(G):
        part = got[gi * gb:(gi + 1) * gb].double()
        smax = float((gi + 1) * gb)      # (the input scale of the group's last sample)
        torch.testing.assert_close(st[gi, :cout, 0], part.sum(dim=(0, 2, 3, 4)), rtol=1e-3, atol=4e-3 * smax * math.sqrt(nvox))
        torch.testing.assert_close(st[gi, :cout, 1], (part ** 2).sum(dim=(0, 2, 3, 4)), rtol=1e-3, atol=4e-3 * smax ** 2 * math.sqrt(nvox))
    # one launch per group (the per-group rows of a runner planned for the group batch) gives the same tensor
    run1 = O.ConvRunner(op, DEV, zm_batch=gb)
    run1.prep(w.to(DEV), b.to(DEV))
    y1 = torch.empty_like(y)
    stats1 = torch.zeros_like(stats)
    run1.run(xs, y1, B, None, None, L.ACT_ELU, 1.0, stats1, stats_nrep=nrep, group_batch=gb)
    assert torch.equal(y1, y)
    torch.testing.assert_close(stats1.view(G, nrep, cpo, 2).sum(1).cpu(), st, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("cin,cout,dims,pad,B,gb", GROUP_CASES)
def test_z_marching_data_gradient_with_batchnorm_backward_sums(cin, cout, dims, pad, B, gb, monkeypatch):
    """g = conv^T(dz, W) on the z-marching kernel, (sum g, sum g*x) per BatchNorm group from its epilogue: the stored g against
    torch, the sums against float64 sums over the stored g and x (exact up to the fp32 accumulation inside a workgroup), and
    against the tiled kernel's stats_mode 1"""
    monkeypatch.setattr(O, "ZM_MIN_PLANES", 0)
    gen = torch.Generator().manual_seed(cin * 3 + cout + B)
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    out = tuple(dims[a] + 2 * pad[a] - 2 for a in range(3))
    x = bf(torch.randn(B, cin, *dims, generator=gen) + 0.5)
    dz = bf(torch.randn(B, cout, *out, generator=gen) * (1.0 + torch.arange(B).view(B, 1, 1, 1, 1)))
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) / math.sqrt(27 * cin)
    dop = P.conv_dgrad_op(cin, cout, 3, 1, pad, dims, cpo, cpi, L.SP_BF16)
    assert O.ConvRunner.zm_plan_bn_bwd_ok(P.zm_plan(dop))
    nrep, G = 4, B // gb
    xs, dzs = _to_cl(x, cpi), _to_cl(dz, cpo)
    res = {}
    for name, zmb in (("zm", B), ("tiled", None)):
        run = O.ConvRunner(dop, DEV, zm_batch=zmb)
        assert run.uses_zm() == (zmb is not None) and (zmb is None or run.zm_bn_bwd_ok())
        run.prep(w.to(DEV))
        gb_t = O.alloc_cl(B, dims, cpi, L.SP_BF16, DEV)
        gb_t.fill_(7.0)
        bs = torch.zeros(G * nrep * cpi * 2, dtype=torch.float64, device=DEV)
        run.run(dzs, gb_t, B, stats=bs, stats_nrep=nrep, stats_mode=1, aux=xs, group_batch=gb)
        res[name] = (_from_cl(gb_t, cin), bs.view(G, nrep, cpi, 2).sum(1).cpu())
    gz, sz = res["zm"]
    torch.testing.assert_close(gz, F.conv_transpose3d(dz, bf(w), padding=pad), rtol=3e-2, atol=3e-2 * B)
    for gi in range(G):
        sl = slice(gi * gb, (gi + 1) * gb)
        e1 = gz[sl].double().sum(dim=(0, 2, 3, 4))
        e2 = (gz[sl].double() * x[sl].double()).sum(dim=(0, 2, 3, 4))
        scale = float(gz[sl].abs().max()) * math.sqrt(gz[sl].numel() / cin)
        torch.testing.assert_close(sz[gi, :cin, 0], e1, rtol=1e-4, atol=2e-5 * scale)
        torch.testing.assert_close(sz[gi, :cin, 1], e2, rtol=1e-4, atol=6e-5 * scale)
        if cpi > cin:
            assert float(sz[gi, cin:].abs().max()) == 0.0
    gt, stl = res["tiled"]
    torch.testing.assert_close(gz, gt, rtol=2e-2, atol=2e-2 * B)      # (another summation order before the bf16 rounding)
    torch.testing.assert_close(sz, stl, rtol=5e-3, atol=5e-2 * B)


def test_cae_step_grouped_z_march_equals_per_group_launches(monkeypatch):
    """the CAE training step (batched passes) with one z-marching launch per layer over all BatchNorm groups -- forward with
    per-group statistics, data gradients with the BatchNorm-backward sums -- against one launch per group / the tiled data
    gradients (SP_ZM_GROUPS=0): two bf16 pipelines of the same function"""
    from test_gpu_round3 import _cae_step      # (pytest puts tests/ on sys.path)
    ch = [1, 16, 24, 32, 100, 200, 1]
    outs = {}
    for on in (False, True):
        monkeypatch.setattr(O, "ZM_GROUPS", on)
        outs[on] = _cae_step(ch, 29, 28, 64, "bf16", 0, batched=1)
    a, b = outs[False], outs[True]
    for k in a[0]:
        d = (a[0][k] - b[0][k]).abs()
        assert float(d.max()) <= 3e-2 and float(d.mean()) <= 2e-3, (k, float(d.max()), float(d.mean()))
    assert abs(a[1] - b[1]) <= 3e-3, (a[1], b[1])
    rel = float((a[2] - b[2]).double().norm() / a[2].double().norm())
    assert rel < 6e-2, rel


# ------------------------------------------------------------------------------------------------ parity classes in one pass
# kind, cin, cout, k, stride, pad, input dims of the forward op, batch, group batch
PAR_CASES = [("convT", 16, 16, 2, 2, 0, (6, 20, 22), 4, 2),          # Cae3D.py:196-204: one tap per class
             ("convT", 24, 24, 2, 2, 0, (7, 13, 19), 2, 1),
             ("convT", 104, 32, 3, 2, 0, (3, 12, 12), 2, 2),         # Cae3D.py:178-180: classes of unequal extent, 7 input planes
             ("dgrad", 16, 24, 3, 2, (1, 1, 1), (12, 30, 36), 4, 2),  # Cae3D.py:45: gradient of the stride-2 convolutions
             ("dgrad", 24, 32, 3, 2, (1, 1, 1), (6, 18, 26), 2, 1),
             ("dgrad", 32, 100, 3, 2, (0, 0, 0), (7, 25, 25), 2, 2),
             ("conv", 16, 24, 3, 2, (1, 1, 1), (12, 30, 36), 4, 2),   # the strided convolutions themselves: one class, input stride 2
             ("conv", 32, 100, 3, 2, (0, 0, 0), (7, 25, 25), 2, 1),
             ("convTd", 16, 16, 2, 2, 0, (6, 20, 22), 4, 2),         # ... and the data gradient of the transposed layers
             ("convTd", 104, 32, 3, 2, 0, (3, 12, 12), 2, 2)]


@pytest.mark.parametrize("kind,cin,cout,k,s,pad,dims,B,gb", PAR_CASES)
def test_parity_classes_in_one_pass(kind, cin, cout, k, s, pad, dims, B, gb, monkeypatch):
    """csrc/sp_conv_par.hip: all parity classes of a transposed convolution (forward, ELU, output statistics per BatchNorm
    group) or of a strided convolution's data gradient (with the (sum g, sum g*x) epilogue) in one launch -- against torch,
    against float64 sums of the stored output, and against one launch per class"""
    gen = torch.Generator().manual_seed(cin + 3 * cout + k)
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    nrep, G = 4, B // gb
    res = {}
    if kind == "convT":
        x = bf(torch.randn(B, cin, *dims, generator=gen))
        w = torch.randn(cin, cout, k, k, k, generator=gen) / math.sqrt(cin * k ** 3 / s ** 3)
        b = torch.randn(cout, generator=gen) * 0.1
        op = P.convT_fwd_op(cin, cout, k, s, pad, dims, cpi, cpo, L.SP_BF16)
        ref = F.elu(F.conv_transpose3d(x, bf(w), b, stride=s, padding=pad), 1.0)
        src, cs, cd, act = _to_cl(x, cpi), cout, cpo, L.ACT_ELU
    elif kind == "conv":
        x = bf(torch.randn(B, cin, *dims, generator=gen))
        w = torch.randn(cout, cin, k, k, k, generator=gen) / math.sqrt(cin * k ** 3)
        b = torch.randn(cout, generator=gen) * 0.1
        op = P.conv_fwd_op(cin, cout, k, s, pad, dims, cpi, cpo, L.SP_BF16)
        ref = F.elu(F.conv3d(x, bf(w), b, stride=s, padding=pad), 1.0)
        src, cs, cd, act = _to_cl(x, cpi), cout, cpo, L.ACT_ELU
    elif kind == "convTd":
        xin = bf(torch.randn(B, cin, *dims, generator=gen) + 0.5)
        w = torch.randn(cin, cout, k, k, k, generator=gen) / math.sqrt(cout * k ** 3 / s ** 3)
        b = None
        xr = xin.clone().requires_grad_(True)
        zr = F.conv_transpose3d(xr, bf(w), None, stride=s, padding=pad)
        dz = bf(torch.randn(zr.shape, generator=gen))
        ref = torch.autograd.grad(zr, xr, dz)[0]
        op = P.convT_dgrad_op(cin, cout, k, s, pad, dims, cpo, cpi, L.SP_BF16)
        src, cs, cd, act = _to_cl(dz, cpo), cin, cpi, L.ACT_NONE
        aux = _to_cl(xin, cpi)
    else:
        out = tuple((dims[a] + 2 * pad[a] - k) // s + 1 for a in range(3))
        xin = bf(torch.randn(B, cin, *dims, generator=gen) + 0.5)
        dz = bf(torch.randn(B, cout, *out, generator=gen))
        w = torch.randn(cout, cin, k, k, k, generator=gen) / math.sqrt(cin * k ** 3)
        b = None
        op = P.conv_dgrad_op(cin, cout, k, s, pad, dims, cpo, cpi, L.SP_BF16)
        xr = xin.clone().requires_grad_(True)
        ref = torch.autograd.grad(F.conv3d(xr, bf(w), None, stride=s, padding=pad), xr, dz)[0]
        src, cs, cd, act = _to_cl(dz, cpo), cin, cpi, L.ACT_NONE
        aux = _to_cl(xin, cpi)
    assert (len(op.subs) > 1) == (kind in ("convT", "dgrad"))
    for name, on in (("par", True), ("classes", False)):
        monkeypatch.setattr(O, "USE_PAR", on)
        run = O.ConvRunner(op, DEV)
        assert (run.par is not None) == on
        run.prep(w.to(DEV), None if b is None else b.to(DEV))
        y = O.alloc_cl(B, op.y_dims, cd, L.SP_BF16, DEV, zero=True)
        st = torch.zeros(G * nrep * cd * 2, dtype=torch.float64, device=DEV)
        if kind in ("convT", "conv"):
            run.run(src, y, B, None, None, act, 1.0, st, stats_nrep=nrep, group_batch=gb)
        elif kind == "convTd" and not on:      # (the register-staged tiled kernel has no stats_mode 1: production reduces separately)
            run.run(src, y, B)
        else:
            run.run(src, y, B, stats=st, stats_nrep=nrep, stats_mode=1, aux=aux, group_batch=gb)
        res[name] = (_from_cl(y, cs), st.view(G, nrep, cd, 2).sum(1).cpu(), y)
    got, sums, yraw = res["par"]
    torch.testing.assert_close(got, ref, rtol=3e-2, atol=3e-2)
    if cd > cs:
        assert float(yraw[..., cs:].float().abs().max()) == 0.0
    for gi in range(G):
        sl = slice(gi * gb, (gi + 1) * gb)
        e1 = got[sl].double().sum(dim=(0, 2, 3, 4))
        e2 = ((got[sl].double() ** 2) if kind in ("convT", "conv") else got[sl].double() * xin[sl].double()).sum(dim=(0, 2, 3, 4))
        scale = float(got[sl].abs().max()) * math.sqrt(got[sl].numel() / cs)
        torch.testing.assert_close(sums[gi, :cs, 0], e1, rtol=1e-4, atol=2e-5 * scale)
        torch.testing.assert_close(sums[gi, :cs, 1], e2, rtol=1e-4, atol=2e-4 * scale)
    # one launch per class (the tiled kernel): the same K order and the same MFMA sequence per output voxel -> the same stored tensor
    assert torch.equal(res["classes"][2], yraw)
    if kind != "convTd":
        torch.testing.assert_close(res["classes"][1], sums, rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------------------------------------ strided weight gradients
# kind, cin, cout, k, stride, pad, input dims of the forward op, batch, workgroups
WGS_CASES = [("conv", 16, 24, 3, 2, (1, 1, 1), (12, 30, 70), 2, None),     # Cae3D.py:45 (ragged: 70 -> 35 output voxels per row)
             ("conv", 24, 32, 3, 2, (1, 1, 1), (6, 18, 26), 2, 16),
             ("conv", 32, 100, 3, 2, (0, 0, 0), (7, 25, 25), 2, None),      # Cae3D.py:62: 7 output tiles of 16, an unused input remainder
             ("convT", 16, 16, 2, 2, 0, (6, 20, 22), 2, None),              # Cae3D.py:196-204: roles swapped, 8 taps
             ("convT", 24, 24, 2, 2, 0, (7, 13, 19), 1, 8),
             ("convT", 104, 32, 3, 2, 0, (3, 12, 12), 2, None)]             # Cae3D.py:178-180


@pytest.mark.parametrize("kind,cin,cout,k,s,pad,dims,B,nblocks", WGS_CASES)
def test_strided_weight_gradient_on_the_dma_kernel(kind, cin, cout, k, s, pad, dims, B, nblocks, monkeypatch):
    """csrc/sp_wgrad_dma.hip with stride 2 / 2x2x2 taps (the staged input tile is the dense box the strided taps reach, read with
    a voxel stride): the CAE's strided convolutions and -- operands swapped -- its transposed ones, against autograd on the same
    bf16-rounded operands and against the register-staged kernel"""
    gen = torch.Generator().manual_seed(cin * 13 + cout + k)
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    x = bf(torch.randn(B, cin, *dims, generator=gen))
    if kind == "conv":
        wr = torch.zeros(cout, cin, k, k, k, requires_grad=True)
        z = F.conv3d(x, wr, stride=s, padding=pad)
    else:
        wr = torch.zeros(cin, cout, k, k, k, requires_grad=True)
        z = F.conv_transpose3d(x, wr, stride=s, padding=pad)
    dz = bf(torch.randn(z.shape, generator=gen))
    z.backward(dz)
    od = tuple(z.shape[2:])
    xs, dzs = _to_cl(x, cpi), _to_cl(dz, cpo)
    if nblocks is not None:
        monkeypatch.setenv("SP_WGRAD_BLOCKS", str(nblocks))
    got = {}
    for on in (True, False):
        monkeypatch.setattr(O, "WGRAD_DMA_STRIDED", on)
        kk = k ** 3
        if kind == "conv":
            wg = O.WgradRunner(cin, cout, k, s, pad, dims, od, cpi, cpo, cin * kk, kk, L.SP_BF16, DEV)
        else:      # shifted operand = dz (on the transposed layer's output grid), fixed operand = x
            wg = O.WgradRunner(cout, cin, k, s, pad, od, dims, cpo, cpi, cout * kk, kk, L.SP_BF16, DEV)
        assert wg.dma == on
        dw = torch.zeros_like(wr, device=DEV)
        if kind == "conv":
            wg.run(xs, dzs, B, dw)
        else:
            wg.run(dzs, xs, B, dw)
        got[on] = dw.cpu()
    scale = float(wr.grad.abs().max())
    torch.testing.assert_close(got[True], wr.grad, rtol=2e-3, atol=2e-3 * scale)
    torch.testing.assert_close(got[True], got[False], rtol=1e-4, atol=2e-4 * scale)


# ------------------------------------------------------------------------------------------------ BatchNorm folded per group
@pytest.mark.parametrize("cin,cout,dims,pad,B,gb", GROUP_CASES + [(16, 16, (5, 37, 21), (2, 2, 1), 4, 1)])
def test_z_marching_forward_with_batchnorm_folded_per_group(cin, cout, dims, pad, B, gb, monkeypatch):
    """y = ELU(conv(zero-padded (s_g x + t_g))) from the RAW input: per-group weight fragments W s_g and a bias table over the border
    classes of the output (the taps that fall into the padding contribute no W t_g) -- sp_conv_prep_folded_groups + sp_conv3d_zm
    with bias_tab / wfrag_gstride, one launch over all groups -- against torch on the normalised, then padded input"""
    monkeypatch.setattr(O, "ZM_MIN_PLANES", 0)
    gen = torch.Generator().manual_seed(cin + 7 * cout + B)
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    G = B // gb
    x = bf(torch.randn(B, cin, *dims, generator=gen))
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) / math.sqrt(27 * cin)
    b = torch.randn(cout, generator=gen) * 0.1
    scale = torch.rand(G, cin, generator=gen) + 0.5
    shift = torch.randn(G, cin, generator=gen)
    coef = torch.zeros(G, 3, cpi)
    coef[:, 0, :cin], coef[:, 2, :cin] = scale, shift
    op = P.conv_fwd_op(cin, cout, 3, 1, pad, dims, cpi, cpo, L.SP_BF16)
    run = O.ConvRunner(op, DEV, zm_batch=B)
    assert run.zm is not None
    z = run.zm
    nfe = z["nsteps"] * z["NT"] * 64 * 8
    ncls = (2 * pad[0] + 1) * (2 * pad[1] + 1) * (2 * pad[2] + 1)
    gfrag = torch.empty(G * nfe, dtype=torch.bfloat16, device=DEV)
    gtab = torch.empty(G * ncls * cpo, dtype=torch.float32, device=DEV)
    wd, bd, cd = w.to(DEV), b.to(DEV), coef.to(DEV)
    L.call("sp_conv_prep_folded_groups", O.ptr(wd), op.w_sco, op.w_sci, cout, cin, O.ptr(z["kmap_d"]), z["nsteps"], z["NT"], O.ptr(gfrag), nfe * 2,
           O.ptr(cd), 3 * cpi, cpi, G, O.ptr(bd), pad[0], pad[1], pad[2], O.ptr(gtab), cpo, O.stream())
    xs = _to_cl(x, cpi)
    nrep = 4
    y = O.alloc_cl(B, op.y_dims, cpo, L.SP_BF16, DEV)
    stats = torch.zeros(G * nrep * cpo * 2, dtype=torch.float64, device=DEV)
    run.run(xs, y, B, None, None, L.ACT_ELU, 1.0, stats, stats_nrep=nrep, group_batch=gb, group_fold=(gfrag, nfe * 2, gtab, ncls * cpo))
    got = _from_cl(y, cout)
    ref = torch.empty_like(got)
    for gi in range(G):
        sl = slice(gi * gb, (gi + 1) * gb)
        xh = x[sl] * scale[gi].view(1, -1, 1, 1, 1) + shift[gi].view(1, -1, 1, 1, 1)
        ref[sl] = F.elu(F.conv3d(xh, w, b, padding=pad), 1.0)
    err = (got - ref).abs()
    assert float(err.max()) < 8e-2 and float(err.mean()) < 6e-3, (float(err.max()), float(err.mean()))
    # the border voxels are where a wrong class would show: held to the same bound on their own
    border = torch.ones_like(ref, dtype=torch.bool)
    border[:, :, 2:-2, 2:-2, 2:-2] = False
    assert float(err[border].max()) < 8e-2
    st = stats.view(G, nrep, cpo, 2).sum(1).cpu()
    for gi in range(G):
        part = got[gi * gb:(gi + 1) * gb].double()
        torch.testing.assert_close(st[gi, :cout, 0], part.sum(dim=(0, 2, 3, 4)), rtol=2e-3, atol=2e-2 * math.sqrt(part.numel() / cout))
    if cpo > cout:
        assert float(y[..., cout:].float().abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ weight gradient on the raw input
@pytest.mark.parametrize("cin,cout,dims,pad,B,gb", [(16, 16, (9, 36, 40), (1, 0, 0), 6, 2), (24, 24, (6, 34, 36), (1, 2, 2), 4, 1),
                                                    (24, 16, (5, 33, 20), (1, 2, 2), 8, 2), (16, 24, (4, 40, 17), (1, 1, 1), 4, 2),
                                                    # input tiles wider than 32 channels: the finish kernel's BatchNorm-backward sums walk the channels in
                                                    # strides of 32 (channels >= 32 were dropped)
                                                    (48, 16, (5, 33, 20), (1, 1, 1), 4, 2), (64, 16, (4, 21, 18), (1, 0, 0), 4, 2)])
def test_weight_gradient_on_the_raw_input_with_batchnorm_folded_per_group(cin, cout, dims, pad, B, gb, monkeypatch):
    """y = conv(zero-padded (s_g x + t_g)): dW, dbias and the BatchNorm-backward sums (sum g, sum g x) of the layer input, from
    (a) the border-class sums of dz that the pass forming dz leaves (sp_bn_act_bwd_groups_cls), (b) the weight gradient kernel on the
    RAW x with group-pure partial blocks, (c) sp_wgrad_finish_folded_groups -- against autograd through the normalised input"""
    gen = torch.Generator().manual_seed(cin * 17 + cout + B)
    cpi, cpo = O.cpad(cin, 16), O.cpad(cout, 16)
    G = B // gb
    out = tuple(dims[a] + 2 * pad[a] - 2 for a in range(3))
    x = bf(torch.randn(B, cin, *dims, generator=gen) + 0.3)
    gup = bf(torch.randn(B, cout, *out, generator=gen))                # what arrives from above; dz = gup here (identity coefficients)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=gen) / math.sqrt(27 * cin)).requires_grad_(True)
    bias = torch.zeros(cout, requires_grad=True)
    scale = torch.rand(G, cin, generator=gen) + 0.5
    shift = torch.randn(G, cin, generator=gen)
    xr = x.clone().requires_grad_(True)
    gidx = torch.arange(B) // gb
    xh = xr * scale[gidx].view(B, cin, 1, 1, 1) + shift[gidx].view(B, cin, 1, 1, 1)
    xh.retain_grad()
    F.conv3d(xh, w, bias, padding=pad).backward(gup)
    ghat = xh.grad                                                      # gradient at the BatchNorm's output
    # ---- (a) dz and its border-class sums
    xs, gs = _to_cl(x, cpi), _to_cl(gup, cpo)
    ident = torch.zeros(G, 3, cpo, device=DEV)
    ident[:, 0] = 1.0
    dz = torch.empty_like(gs)
    ncls = (2 * pad[0] + 1) * (2 * pad[1] + 1) * (2 * pad[2] + 1)
    cls = torch.zeros(G * ncls * cpo, dtype=torch.float64, device=DEV)
    dbs = torch.zeros(L.SP_REDUCE_ROWS * cpo, dtype=torch.float64, device=DEV)
    O.bn_act_bwd(gs, gs, ident, L.SP_BF16, L.ACT_NONE, 0.0, dz, dbs, cls=(gb, pad, cls))
    assert torch.equal(dz, gs)
    tot = cls.view(G, ncls, cpo).sum(1).cpu()
    for gi in range(G):
        torch.testing.assert_close(tot[gi, :cout], gup[gi * gb:(gi + 1) * gb].double().sum(dim=(0, 2, 3, 4)), rtol=1e-9, atol=1e-6)
    torch.testing.assert_close(dbs.view(L.SP_REDUCE_ROWS, cpo).sum(0).cpu()[:cout], gup.double().sum(dim=(0, 2, 3, 4)), rtol=1e-9, atol=1e-6)
    # ---- (b) + (c)
    wg = O.WgradRunner(cin, cout, 3, 1, pad, dims, out, cpi, cpo, cin * 27, 27, L.SP_BF16, DEV)
    wg.groups = G
    wg.run_raw(xs, dz, B)
    coef = torch.zeros(G, 3, cpi, device=DEV)
    coef[:, 0, :cin], coef[:, 2, :cin] = scale.to(DEV), shift.to(DEV)
    dw = torch.zeros(cout, cin, 3, 3, 3, device=DEV)
    db = torch.zeros(cout, device=DEV)
    nrep = 4
    bs = torch.zeros(G * nrep * cpi * 2, dtype=torch.float64, device=DEV)
    wd = w.detach().to(DEV)
    L.call("sp_wgrad_finish_folded_groups", O.ptr(wg.acc), wg.nparts, 27, G, wg.cot * 16, wg.cit * 16, cout, cin, wg.w_sco, wg.w_sci, O.ptr(coef), 3 * cpi, cpi,
           O.ptr(cls), pad[0], pad[1], pad[2], O.ptr(wd), O.ptr(dw), O.ptr(db), O.ptr(bs), nrep, cpi, O.stream())
    sw = float(w.grad.abs().max())
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=2e-3, atol=2e-3 * sw)
    torch.testing.assert_close(db.cpu(), bias.grad, rtol=1e-4, atol=1e-3)
    got = bs.view(G, nrep, cpi, 2).sum(1).cpu()
    for gi in range(G):
        sl = slice(gi * gb, (gi + 1) * gb)
        e1 = ghat[sl].double().sum(dim=(0, 2, 3, 4))
        e2 = (ghat[sl].double() * x[sl].double()).sum(dim=(0, 2, 3, 4))
        sc = float(ghat[sl].abs().max()) * math.sqrt(ghat[sl].numel() / cin)
        torch.testing.assert_close(got[gi, :cin, 0], e1, rtol=2e-3, atol=2e-3 * sc)
        torch.testing.assert_close(got[gi, :cin, 1], e2, rtol=2e-3, atol=2e-3 * sc)


# ------------------------------------------------------------------------------------------------ fused reconstruction loss
@pytest.mark.parametrize("factor", [0.0, 0.36])
def test_fused_cae_reconstruction_loss_equals_the_composed_one(factor, monkeypatch):
    """metrics.cae_reconstruction_loss (sp_cae_loss_fwd / _bwd: three launches) against the reference's own recipe composed of torch
    operators and BatchDiceLoss calls (CaeReconstructionLearner.py:52-70): value and the gradients of the four reconstructions
    (slices of one stacked tensor, as the batched decoder hands them out) and of the two latents"""
    from types import SimpleNamespace as NS
    from stroke_prediction_amd.common import metrics
    g = torch.Generator().manual_seed(3)
    B, dims = 2, (5, 12, 20)
    stacked = torch.rand(4 * B, 1, *dims, generator=g).to(DEV).requires_grad_(True)
    gts = [(torch.rand(B, 1, *dims, generator=g) > 0.6).float().to(DEV) for _ in range(3)]
    zi = torch.randn(B, 50, 1, 2, 2, generator=g).to(DEV).requires_grad_(True)
    zl = torch.randn(B, 50, 1, 2, 2, generator=g).to(DEV).requires_grad_(True)
    crit = metrics.BatchDiceLoss([1.0])
    res = {}
    for fused in (True, False):
        monkeypatch.setenv("SP_CAE_FUSED_LOSS", "1" if fused else "0")
        for t in (stacked, zi, zl):
            t.grad = None
        parts = [stacked[k * B:(k + 1) * B] for k in range(4)]          # decoder passes: core, penu, lesion, interpolation
        rec = NS(core=parts[0], penu=parts[1], lesion=parts[2], interpolation=parts[3])
        gt = NS(core=gts[0], penu=gts[1], lesion=gts[2])
        lat = NS(interpolation=zi, lesion=zl)
        loss = metrics.cae_reconstruction_loss(rec, gt, lat, factor, crit)
        assert (loss is not None) == fused
        if loss is None:
            d1, d2 = rec.penu - rec.interpolation, rec.penu - rec.core
            loss = (torch.mean(torch.abs(d1) - d1) + torch.mean(torch.abs(d2) - d2) + crit(rec.core, gt.core) + crit(rec.penu, gt.penu)
                    + crit(rec.lesion, gt.lesion) + factor * torch.mean(torch.abs(zi - zl))) / (5 + factor)
        (loss * 1.7).backward()
        res[fused] = (float(loss), stacked.grad.clone(), zi.grad.clone(), zl.grad.clone())
    a, b = res[True], res[False]
    assert abs(a[0] - b[0]) < 2e-6, (a[0], b[0])
    for k in (1, 2, 3):
        torch.testing.assert_close(a[k], b[k], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("cin", [16, 48, 64])
def test_pointwise_layer_on_the_raw_input_per_group(cin, monkeypatch):
    """the CAE's 1x1x1 16 -> 16 layer in the batched passes: forward with each group's BatchNorm applied on the operand load (one launch,
    statistics per group), weight gradient on the raw input with group-pure partial blocks + the group-aware folded finish"""
    gen = torch.Generator().manual_seed(77)
    cout = 16
    B, gb, dims = 4, 2, (4, 16, 32)
    G = B // gb
    x = bf(torch.randn(B, cin, *dims, generator=gen) + 0.2)
    w = (torch.randn(cout, cin, 1, 1, 1, generator=gen) / math.sqrt(cin)).requires_grad_(True)
    bias = (torch.randn(cout, generator=gen) * 0.1).requires_grad_(True)
    scale, shift = torch.rand(G, cin, generator=gen) + 0.5, torch.randn(G, cin, generator=gen)
    gidx = torch.arange(B) // gb
    xr = x.clone().requires_grad_(True)
    xh = xr * scale[gidx].view(B, cin, 1, 1, 1) + shift[gidx].view(B, cin, 1, 1, 1)
    xh.retain_grad()
    yref = F.elu(F.conv3d(xh, w, bias), 1.0)
    dy = bf(torch.randn(yref.shape, generator=gen))
    # ---- forward
    op = P.conv_fwd_op(cin, cout, 1, 1, 0, dims, cin, cout, L.SP_BF16)
    run = O.ConvRunner(op, DEV)
    assert run.fc is not None and run.fc["pointwise"]
    run.prep(w.detach().to(DEV), bias.detach().to(DEV))
    coef = torch.zeros(G, 3, cin, device=DEV)
    coef[:, 0], coef[:, 2] = scale.to(DEV), shift.to(DEV)
    xs = _to_cl(x, cin)
    y = O.alloc_cl(B, dims, cout, L.SP_BF16, DEV)
    nrep = 4
    st = torch.zeros(G * nrep * cout * 2, dtype=torch.float64, device=DEV)
    run.run(xs, y, B, coef[:, 0], coef[:, 2], L.ACT_ELU, 1.0, st, stats_nrep=nrep, group_batch=gb, coef_gstride=3 * cin)
    got = _from_cl(y, cout)
    torch.testing.assert_close(got, yref.detach(), rtol=3e-2, atol=3e-2)
    sums = st.view(G, nrep, cout, 2).sum(1).cpu()
    for gi in range(G):
        torch.testing.assert_close(sums[gi, :, 0], got[gi * gb:(gi + 1) * gb].double().sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-2)
    # ---- backward: dz = dy * ELU'(y) taken from torch; weight gradient on the raw input
    yref.backward(dy)
    dzt = bf(dy * torch.where(yref.detach() > 0, torch.ones_like(dy), yref.detach() + 1.0))
    dzs = _to_cl(dzt, cout)
    cls = torch.zeros(G * 1 * cout, dtype=torch.float64, device=DEV)
    ident = torch.zeros(G, 3, cout, device=DEV)
    ident[:, 0] = 1.0
    dz2 = torch.empty_like(dzs)
    O.bn_act_bwd(dzs, dzs, ident, L.SP_BF16, L.ACT_NONE, 0.0, dz2, None, cls=(gb, (0, 0, 0), cls))
    wg = O.WgradRunner(cin, cout, 1, 1, 0, dims, dims, cin, cout, cin, 1, L.SP_BF16, DEV)
    assert wg.pw
    wg.groups = G
    wg.run_raw(xs, dz2, B)
    dw = torch.zeros(cout, cin, 1, 1, 1, device=DEV)
    db = torch.zeros(cout, device=DEV)
    bs = torch.zeros(G * nrep * cin * 2, dtype=torch.float64, device=DEV)
    L.call("sp_wgrad_finish_folded_groups", O.ptr(wg.acc), wg.nparts, 1, G, wg.cot * 16, wg.cit * 16, cout, cin, wg.w_sco, wg.w_sci, O.ptr(coef), 3 * cin, cin,
           O.ptr(cls), 0, 0, 0, O.ptr(w.detach().to(DEV)), O.ptr(dw), O.ptr(db), O.ptr(bs), nrep, cin, O.stream())
    # reference gradients with the SAME (bf16-rounded) dz: dW = sum dz x^, dbias = sum dz
    dw_ref = torch.einsum("bovwx,bivwx->oi", dzt.double(), xh.detach().double()).float().view(cout, cin, 1, 1, 1)
    torch.testing.assert_close(dw.cpu(), dw_ref, rtol=3e-3, atol=3e-3 * float(dw_ref.abs().max()))
    torch.testing.assert_close(db.cpu(), dzt.double().sum(dim=(0, 2, 3, 4)).float(), rtol=1e-4, atol=1e-3)
    ghat = torch.einsum("bovwx,oi->bivwx", dzt.double(), w.detach().double().view(cout, cin))
    gotb = bs.view(G, nrep, cin, 2).sum(1).cpu()
    for gi in range(G):
        sl = slice(gi * gb, (gi + 1) * gb)
        torch.testing.assert_close(gotb[gi, :, 0], ghat[sl].sum(dim=(0, 2, 3, 4)), rtol=2e-3, atol=2e-2)
        torch.testing.assert_close(gotb[gi, :, 1], (ghat[sl] * x[sl].double()).sum(dim=(0, 2, 3, 4)), rtol=2e-3, atol=2e-2)


def test_cae_training_trajectory_with_and_without_the_raw_input_path(monkeypatch):
    """eight CaeReconstructionLearner.train_batch steps (bf16, batched passes, hipGraph replay) with the round-4 paths -- BatchNorm
    folded per group into the padded layers, weight gradients on the raw input, pointwise layer on the raw input, fused loss -- and
    with all of them off: the loss curves stay together (two bf16 pipelines of the same function) and fall"""
    from test_gpu_round3 import _cae_step
    ch = [1, 16, 24, 32, 100, 200, 1]
    curves = {}
    for on in (True, False):
        monkeypatch.setattr(O, "FOLD_GROUPS", on)
        monkeypatch.setattr(O, "RAW_WGRAD", on)
        monkeypatch.setenv("SP_CAE_FUSED_LOSS", "1" if on else "0")
        curves[on] = _cae_step(ch, 41, 28, 64, "bf16", 0, graph=True, steps=8, batched=1, warmup=2)[4]      # (two eager steps: the second sees the weights Adam changed, as every captured step will)
    a, b = curves[True], curves[False]
    assert len(a) == len(b) == 8
    assert max(abs(x - y) for x, y in zip(a, b)) < 4e-3, (a, b)
    assert a[-1] < a[0] - 0.01 and b[-1] < b[0] - 0.01, (a, b)
