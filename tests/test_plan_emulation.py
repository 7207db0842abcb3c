"""Replay the conv planner's K tables (kmap / ktab) on the CPU and compare with
torch.nn.functional: validates taps, padding, stride, parity classes of transposed
convolutions, channel grouping and the weight-stride conventions without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from stroke_prediction_amd.runtime import plan as P


def emulate(op, x_cl, w_flat, batch):
    """x_cl: (B, D, H, W, CPi) float64; w_flat: 1-D weight storage. Returns y (B, *y_dims, CPo)."""
    B = batch
    y = torch.zeros((B,) + tuple(op.y_dims) + (op.cpo,), dtype=torch.float64)
    Di, Hi, Wi = op.in_dims
    xp = x_cl
    for sub in op.subs:
        t = sub.tile
        steps, opg, opp = t["steps_per_group"], t["octs_per_group"], t["opp"]
        qd, qh, qw = sub.out_dims
        acc = torch.zeros((B, qd, qh, qw, op.cout), dtype=torch.float64)
        qz, qy, qx = torch.meshgrid(torch.arange(qd), torch.arange(qh), torch.arange(qw), indexing="ij")
        for g in range(t["ngroups"]):
            for i in range(steps * 4):
                km = int(sub.kmap[g * steps * 4 + i])
                if km < 0:
                    continue
                src, oct_g = km >> 16, km & 0xffff
                off = int(sub.ktab[i])
                plane, rem = divmod(off, t["plane_bytes"])
                vox, r2 = divmod(rem, t["vsb"])
                po = r2 // 16
                assert r2 % 16 == 0 and po < opp
                assert oct_g == g * opg + plane * opp + po, "kmap/ktab disagree on the channel octet"
                vz, r3 = divmod(vox, t["ITH"] * t["ITW"])
                vy, vx = divmod(r3, t["ITW"])
                iz = qz * op.stride[0] + sub.o0[0] + vz
                iy = qy * op.stride[1] + sub.o0[1] + vy
                ix = qx * op.stride[2] + sub.o0[2] + vx
                ok = (iz >= 0) & (iz < Di) & (iy >= 0) & (iy < Hi) & (ix >= 0) & (ix < Wi)
                xv = xp[:, iz.clamp(0, Di - 1), iy.clamp(0, Hi - 1), ix.clamp(0, Wi - 1), oct_g * 8:oct_g * 8 + 8]
                xv = xv * ok[None, ..., None]
                for j in range(8):
                    ci = oct_g * 8 + j
                    if ci >= op.cin:
                        continue
                    wv = torch.tensor([w_flat[co * op.w_sco + ci * op.w_sci + src] for co in range(op.cout)],
                                      dtype=torch.float64)
                    acc += xv[..., j:j + 1] * wv
        oz = qz * sub.out_stride[0] + sub.out_off[0]
        oy = qy * sub.out_stride[1] + sub.out_off[1]
        ox = qx * sub.out_stride[2] + sub.out_off[2]
        y[:, oz, oy, ox, :op.cout] = acc
    return y


def to_cl(x, cp):
    B, C = x.shape[:2]
    out = torch.zeros((B,) + tuple(x.shape[2:]) + (cp,), dtype=torch.float64)
    out[..., :C] = x.permute(0, 2, 3, 4, 1).double()
    return out


def from_cl(y, c):
    return y[..., :c].permute(0, 4, 1, 2, 3)


CONV_CASES = [
    (2, 16, 3, 1, (0, 0, 0), (6, 7, 19)),     # Unet3D.py:19 first layer
    (16, 16, 3, 1, (1, 0, 0), (5, 6, 20)),    # Cae3D.py:44
    (16, 24, 3, 2, (1, 1, 1), (6, 9, 21)),    # Cae3D.py:48
    (24, 32, 3, 2, (0, 0, 0), (7, 9, 19)),    # Cae3D.py:70 style
    (32, 24, 3, 1, (1, 2, 2), (4, 5, 17)),    # Cae3D.py:189
    (16, 5, 1, 1, (0, 0, 0), (3, 4, 18)),     # 1x1x1 head
    (40, 16, 3, 1, (0, 0, 0), (5, 5, 18)),    # multi-group-ish
]


@pytest.mark.parametrize("cin,cout,k,s,p,dims", CONV_CASES)
def test_conv_fwd_and_dgrad_tables(cin, cout, k, s, p, dims):
    torch.manual_seed(0)
    x = torch.randn(1, cin, *dims, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, k, k, k, dtype=torch.float64)
    ref = F.conv3d(x, w, None, stride=s, padding=p)
    cpi, cpo = -(-cin // 8) * 8, -(-cout // 8) * 8
    op = P.conv_fwd_op(cin, cout, k, s, p, dims, cpi, cpo)
    y = emulate(op, to_cl(x.detach(), cpi), w.reshape(-1).tolist(), 1)
    assert tuple(op.y_dims) == tuple(ref.shape[2:])
    torch.testing.assert_close(from_cl(y, cout), ref.detach(), rtol=1e-10, atol=1e-10)
    # data gradient
    dz = torch.randn_like(ref)
    (gref,) = torch.autograd.grad(ref, x, dz)
    dop = P.conv_dgrad_op(cin, cout, k, s, p, dims, cpo, cpi)
    g = emulate(dop, to_cl(dz, cpo), w.reshape(-1).tolist(), 1)
    torch.testing.assert_close(from_cl(g, cin), gref, rtol=1e-10, atol=1e-10)


CONVT_CASES = [
    (24, 16, 3, 1, 0, (1, 4, 17)),   # Cae3D.py:178 style
    (16, 8, 3, 2, 0, (3, 5, 17)),    # Cae3D.py:182
    (8, 8, 2, 2, 0, (4, 5, 16)),     # Cae3D.py:193,204
]


@pytest.mark.parametrize("cin,cout,k,s,p,dims", CONVT_CASES)
def test_convT_fwd_and_dgrad_tables(cin, cout, k, s, p, dims):
    torch.manual_seed(1)
    x = torch.randn(1, cin, *dims, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cin, cout, k, k, k, dtype=torch.float64)
    ref = F.conv_transpose3d(x, w, None, stride=s, padding=p)
    cpi, cpo = -(-cin // 8) * 8, -(-cout // 8) * 8
    op = P.convT_fwd_op(cin, cout, k, s, p, dims, cpi, cpo)
    assert tuple(op.y_dims) == tuple(ref.shape[2:])
    y = emulate(op, to_cl(x.detach(), cpi), w.reshape(-1).tolist(), 1)
    torch.testing.assert_close(from_cl(y, cout), ref.detach(), rtol=1e-10, atol=1e-10)
    dz = torch.randn_like(ref)
    (gref,) = torch.autograd.grad(ref, x, dz)
    dop = P.convT_dgrad_op(cin, cout, k, s, p, dims, cpo, cpi)
    g = emulate(dop, to_cl(dz, cpo), w.reshape(-1).tolist(), 1)
    torch.testing.assert_close(from_cl(g, cin), gref, rtol=1e-10, atol=1e-10)


def test_tiles_fit_lds_for_all_reference_layers():
    """Every conv of the two BASELINE networks gets a plan that fits 160 KiB of LDS."""
    for dt in (0, 1):
        for ci, co, d in [(2, 16, 128), (16, 16, 126), (16, 32, 62), (32, 32, 60), (32, 64, 29), (64, 64, 27),
                          (96, 32, 50), (32, 32, 48), (48, 16, 92), (16, 16, 90)]:
            cpi = -(-ci // 8) * 8
            op = P.conv_fwd_op(ci, co, 3, 1, 0, (d, d, d), cpi, co, dtype=dt)
            assert op.subs[0].tile["lds_bytes"] <= 160 * 1024
            assert op.subs[0].tile["read_cycles"] <= 4.5   # bank-conflict model: near conflict-free
            dop = P.conv_dgrad_op(ci, co, 3, 1, 0, (d, d, d), co, cpi, dtype=dt)
            assert dop.subs[0].tile["lds_bytes"] <= 160 * 1024


def emulate_zm(op, zm, x_cl, w_flat):
    """the z-marching kernel's view of its tables (csrc/sp_conv_zm.hip): input plane zi adds, for dz = 0..2, the in-plane
    octets of every K step into output plane zi - dz; ktab gives the octet's position inside a ring slot, kmap the weight"""
    sub = op.subs[0]
    qd, qh, qw = sub.out_dims
    Di, Hi, Wi = op.in_dims
    ith, ks = zm["ITH"], zm["KS"]
    y = torch.zeros((1, qd, qh, qw, op.cpo), dtype=torch.float64)
    qy, qx = torch.meshgrid(torch.arange(qh), torch.arange(qw), indexing="ij")
    for zi in range(sub.o0[0], sub.o0[0] + qd + 2):            # input planes the march visits
        for dz in range(3):
            zo = zi - sub.o0[0] - dz
            if not (0 <= zo < qd) or not (0 <= zi < Di):
                continue                                        # (out-of-volume planes arrive as zeros)
            for e in range(ks * 4):
                km = int(zm["kmap"][dz * ks * 4 + e])
                if km < 0:
                    continue
                src, octet = km >> 16, km & 0xffff
                off = int(zm["ktab"][e])
                p, rem = divmod(off, ith * P.ZM_ITW * 32)      # (a plane keeps the pitch of the instance's largest halo tile)
                vox, r2 = divmod(rem, 32)
                dy, dx = divmod(vox, zm["TW"] + 2)
                assert r2 % 16 == 0 and octet == p * 2 + r2 // 16, "kmap / ktab disagree on the channel octet"
                assert dy < 3 and dx < 3
                iy, ix = qy + sub.o0[1] + dy, qx + sub.o0[2] + dx
                ok = (iy >= 0) & (iy < Hi) & (ix >= 0) & (ix < Wi)
                xv = x_cl[0, zi][iy.clamp(0, Hi - 1), ix.clamp(0, Wi - 1)][..., octet * 8:octet * 8 + 8] * ok[..., None]
                for j in range(8):
                    ci = octet * 8 + j
                    wv = torch.tensor([w_flat[co * op.w_sco + ci * op.w_sci + src] for co in range(op.cout)], dtype=torch.float64)
                    y[0, zo, :, :, :op.cout] += xv[..., j:j + 1] * wv
    return y


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (16, 48), (32, 16), (32, 32), (48, 16)])
def test_z_marching_tables(cin, cout):
    """forward and data gradient of a valid 3x3x3 convolution through the z-marching plan's tables"""
    torch.manual_seed(cin + cout)
    dims = (5, 6, 19)
    x = torch.randn(1, cin, *dims, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, 3, dtype=torch.float64)
    ref = F.conv3d(x, w)
    op = P.conv_fwd_op(cin, cout, 3, 1, 0, dims, cin, cout)
    zm = P.zm_plan(op)
    assert zm is not None and (zm["P"], zm["NT"]) == (cin // 16, cout // 16) and zm["nsteps"] == 3 * zm["KS"]
    y = emulate_zm(op, zm, to_cl(x.detach(), cin), w.reshape(-1).tolist())
    torch.testing.assert_close(from_cl(y, cout), ref.detach(), rtol=1e-10, atol=1e-10)
    dz = torch.randn_like(ref)
    (gref,) = torch.autograd.grad(ref, x, dz)
    dop = P.conv_dgrad_op(cin, cout, 3, 1, 0, dims, cout, cin)
    zmd = P.zm_plan(dop)
    assert zmd is not None and (zmd["P"], zmd["NT"]) == (cout // 16, cin // 16)
    g = emulate_zm(dop, zmd, to_cl(dz, cout), w.reshape(-1).tolist())
    torch.testing.assert_close(from_cl(g, cin), gref, rtol=1e-10, atol=1e-10)
    # ring-slot geometry: every table offset lies inside the staged plane set
    for z in (zm, zmd):
        assert int(z["ktab"].max()) + 16 <= z["P"] * z["ITH"] * P.ZM_ITW * 32
    assert P.zm_plan(P.conv_fwd_op(96, 32, 3, 1, 0, dims, 96, 32)) is None         # weight set too large: tiled kernel
    assert P.zm_plan(P.conv_fwd_op(16, 24, 3, 2, 1, dims, 16, 24)) is None         # strided


def test_zm_tile_covers_the_plane_with_the_fewest_tiles():
    """plan.zm_tile: the flattened (row, column) tile of a z-marching workgroup -- never more tiles than the classic NW MT x 16
    shape, within the instance's voxel and halo budgets, and the shapes the headline network's planes get"""
    for rows in (8, 16, 32):
        for ho, wo in ((50, 50), (60, 60), (58, 58), (90, 90), (88, 88), (124, 124), (46, 46), (25, 25), (126, 128), (7, 300), (3, 5)):
            tw, th = P.zm_tile(ho, wo, rows // 4, 4)
            assert tw * th <= 16 * rows and (tw + 2) * (th + 2) <= (rows + 2) * 18
            assert -(-wo // tw) * -(-ho // th) <= -(-wo // 16) * -(-ho // rows)
    assert P.zm_tile(50, 50, 4, 4) == (25, 10)           # 10 tiles per plane instead of 16
    assert P.zm_tile(48, 48, 4, 4) == (16, 16)           # the classic shape where it divides the plane
    op = P.conv_dgrad_op(32, 32, 3, 1, 0, (50, 50, 50), 32, 32)
    z = P.zm_plan(op)
    assert (z["TW"], z["TH"]) == P.zm_tile(50, 50, z["NW"], z["MT"])
    assert int(z["ktab"].max()) + 16 <= z["P"] * z["ITH"] * P.ZM_ITW * 32


@pytest.mark.parametrize("cin,cout", [(48, 16), (96, 32), (32, 16)])
def test_plane_serial_tables(cin, cout, monkeypatch):
    """plan.zm_pser_plan (round 5): the z-march that takes one 16-channel plane per sub-step -- the one-plane K table and the
    per-plane weight maps reproduce the convolution (every (tap, input octet) exactly once)"""
    monkeypatch.setattr(P, "ZM_PSER_ON", True)
    torch.manual_seed(cin + cout)
    dims = (5, 6, 19)
    x = torch.randn(1, cin, *dims, dtype=torch.float64)
    w = torch.randn(cout, cin, 3, 3, 3, dtype=torch.float64)
    ref = F.conv3d(x, w)
    op = P.conv_fwd_op(cin, cout, 3, 1, 0, dims, cin, cout)
    z = P.zm_pser_plan(op)
    assert z is not None and z["pser"] and z["P"] == 1 and z["PT"] == cin // 16 and z["NT"] == cout // 16 and z["nsteps"] == z["PT"] * 3 * z["KS"]
    assert (z["MT"], z["nslot"], z["NW"]) == P.ZM_CONFIGS_PS[(cout // 16, False)]
    sub = op.subs[0]
    qd, qh, qw = sub.out_dims
    xc = to_cl(x, cin)
    wf = w.reshape(-1).tolist()
    y = torch.zeros((1, qd, qh, qw, cout), dtype=torch.float64)
    qy, qx = torch.meshgrid(torch.arange(qh), torch.arange(qw), indexing="ij")
    ks = z["KS"]
    seen = set()
    for zi in range(qd + 2):
        for p in range(z["PT"]):                      # the sub-steps of input plane zi
            for dz in range(3):
                zo = zi - dz
                if not (0 <= zo < qd):
                    continue
                for e in range(ks * 4):
                    km = int(z["kmap"][((p * 3 + dz) * ks) * 4 + e])
                    if km < 0:
                        continue
                    src, octet = km >> 16, km & 0xffff
                    assert octet // 2 == p
                    vox, r2 = divmod(int(z["ktab"][e]), 32)
                    dy, dx = divmod(vox, z["TW"] + 2)
                    assert r2 // 16 == octet % 2 and dy < 3 and dx < 3
                    if zi == 2:
                        seen.add((src, octet))
                    xv = xc[0, zi][qy + dy, qx + dx][..., octet * 8:octet * 8 + 8]
                    for j in range(8):
                        ci = octet * 8 + j
                        wv = torch.tensor([wf[co * op.w_sco + ci * op.w_sci + src] for co in range(cout)], dtype=torch.float64)
                        y[0, zo, :, :, :cout] += xv[..., j:j + 1] * wv
    torch.testing.assert_close(from_cl(y, cout), ref, rtol=1e-10, atol=1e-10)
    assert len(seen) == 27 * cin // 8
    assert P.zm_pser_plan(P.conv_fwd_op(16, 16, 3, 1, 0, dims, 16, 16)) is None          # one input plane: the plain march
    assert P.zm_pser_plan(P.conv_fwd_op(64, 64, 3, 1, 0, dims, 64, 64)) is None          # four output tiles: no instance


def test_fc_plan_tables():
    """runtime/plan.py:fc_plan (split-K kernel for the FC-like layers): tap-major K order, steps per tap padded to the kernel's
    prefetch depth, every (tap, octet) exactly once, padding entries -1"""
    from stroke_prediction_amd.runtime import plan as P
    op = P.convT_fwd_op(800, 100, 3, 1, 0, (1, 10, 10), 800, 112, 0)
    f = P.fc_plan(op)
    assert f is not None and f["ntap"] == 27 and f["NT"] == 7
    assert f["spt"] % 4 == 0 and f["spt"] * 4 >= 100 and f["nsteps"] == 27 * f["spt"]
    km = f["kmap"].reshape(27, f["spt"] * 4)
    for t in range(27):
        real = km[t][km[t] >= 0]
        assert len(real) == 100 and set(int(v) & 0xffff for v in real) == set(range(100))
        assert len({int(v) >> 16 for v in real}) == 1
    assert len({int(km[t][0]) >> 16 for t in range(27)}) == 27          # every source tap once
    assert f["taps"].shape == (27, 3)
    # shallow K or big volumes stay on the tiled kernels
    assert P.fc_plan(P.conv_fwd_op(100, 800, 3, 1, 0, (3, 12, 12), 112, 800, 0)) is None
    assert P.fc_plan(P.conv_fwd_op(256, 16, 3, 1, 0, (40, 40, 40), 256, 16, 0)) is None


def test_zm_plan_accepts_padded_channel_counts_and_padding():
    """the CAE's 24-channel layers (pitch 32) and padded convolutions get z-marching plans; the tables only depend on (P, NT) and
    the workgroup's tile"""
    from stroke_prediction_amd.runtime import plan as P
    a = P.zm_plan(P.conv_fwd_op(24, 24, 3, 1, (1, 2, 2), (8, 40, 40), 32, 32, 0), tile=(16, 16))
    b = P.zm_plan(P.conv_fwd_op(32, 32, 3, 1, 0, (8, 40, 40), 32, 32, 0), tile=(16, 16))
    assert a is not None and (a["P"], a["NT"]) == (2, 2)
    np.testing.assert_array_equal(a["ktab"], b["ktab"])
    np.testing.assert_array_equal(a["kmap"], b["kmap"])
    assert P.zm_plan(P.conv_fwd_op(24, 24, 3, 1, 0, (8, 40, 40), 24, 24, 0)) is None       # pitch not a multiple of 16


def test_parity_classes_share_one_row_tile():
    """sp_conv3d_igemm_multi needs one register blocking for all classes of an op"""
    from stroke_prediction_amd.runtime import plan as P
    for op in (P.convT_fwd_op(100, 32, 3, 2, 0, (3, 12, 12), 112, 32, 0), P.convT_fwd_op(16, 16, 2, 2, 0, (14, 62, 62), 16, 16, 0),
               P.conv_dgrad_op(16, 24, 3, 2, 1, (28, 124, 124), 32, 16, 0)):
        assert len(op.subs) == 8
        assert len({(sb.tile["MT"], sb.tile["TD"], sb.tile["TH"]) for sb in op.subs}) == 1
