"""Host-side planning for the table-driven implicit-GEMM convolution (``sp_conv3d_igemm``).

A convolution-like op (``nn.Conv3d`` forward, its data gradient, ``nn.ConvTranspose3d``
forward, ...) is decomposed into one or more *sub-convolutions*: dense stride-``s``
correlations over the input with a list of taps, whose outputs are written with an
output stride/offset (the ``s^3`` parity classes of a transposed convolution).  For each
sub-convolution the planner fixes the LDS tiling and emits two tables:

``kmap[step*4+g]``  (src_tap << 16) | cin_octet   -> weight re-packing (``sp_conv_prep_weights``)
``ktab[step*4+g]``  LDS byte offset of that octet  -> kernel K loop

Pure Python/numpy: testable without a GPU (``tests/test_plan_emulation.py`` replays the
tables against ``torch.nn.functional`` on the CPU).
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import os

import numpy as np

LDS_BUDGET = {0: 72 * 1024, 1: 144 * 1024, 2: 144 * 1024}   # per workgroup, by dtype (bf16 / f32 split / bf16 pairs: hi + lo planes in LDS)

# ds_read_b128 services a wave in 4 groups of 16 lanes (MI355X_MICROARCH.md, LDS)
_B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
                list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
                list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
                list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def _b128_cycles(addr):
    tot = 0
    for grp in _B128_GROUPS:
        slots = {}
        for l in grp:
            slots.setdefault((addr[l] // 16) % 16, set()).add(addr[l])
        tot += max(len(v) for v in slots.values())
    return tot


@dataclass
class SubConv:
    """One dense correlation: out[q] = sum_taps w[src] * in[q*s + o0 + off]."""
    taps: List[Tuple[int, int, int, int]]           # (offD, offH, offW, src_tap_index), offsets >= 0
    o0: Tuple[int, int, int]                        # input coordinate of (q=0, off=0)
    out_dims: Tuple[int, int, int]                  # logical output grid (qD, qH, qW)
    out_stride: Tuple[int, int, int] = (1, 1, 1)    # y coordinate = q*out_stride + out_off
    out_off: Tuple[int, int, int] = (0, 0, 0)
    # filled by plan_tiles
    tile: dict = field(default_factory=dict)
    kmap: np.ndarray = None
    ktab: np.ndarray = None

    @property
    def ext(self):
        return tuple(max(t[a] for t in self.taps) + 1 for a in range(3))


@dataclass
class ConvOp:
    """A planned convolution-like op between channels-last tensors."""
    cin: int
    cout: int
    cpi: int                     # input channel pitch
    cpo: int                     # output channel pitch
    in_dims: Tuple[int, int, int]
    y_dims: Tuple[int, int, int]
    stride: Tuple[int, int, int]  # input step per output step (all sub-convs)
    w_sco: int                   # weight element (co, ci, tap) = w[co*w_sco + ci*w_sci + tap]
    w_sci: int
    subs: List[SubConv]
    dtype: int = 0
    nt: int = 1
    nttot: int = 1
    algo_macs: int = 0           # ALGORITHMIC multiply-accumulates per sample and (cin, cout) pair: voxels x taps of the
                                 # convolution this op belongs to (a data gradient counts the forward's output voxels, not
                                 # the zero-padded taps its dense sub-convolutions run over)

    def flops(self, batch):
        """algorithmic FLOPs of one launch set (2 x 27 x Cin x Cout per output voxel of the convolution)"""
        return 2 * batch * self.algo_macs * self.cin * self.cout

    def issued_macs(self):
        return sum(int(np.prod(s.out_dims)) * len(s.taps) for s in self.subs)


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


def conv_fwd_op(cin, cout, k, stride, pad, in_dims, cpi, cpo, dtype=0):
    """nn.Conv3d forward; weight (cout, cin, k, k, k)."""
    k, s, p = _triple(k), _triple(stride), _triple(pad)
    out = tuple((in_dims[a] + 2 * p[a] - k[a]) // s[a] + 1 for a in range(3))
    assert min(out) >= 1, "convolution output is empty: input %s kernel %s" % (in_dims, k)
    taps = [(a, b, c, (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])]
    kk = k[0] * k[1] * k[2]
    sub = SubConv(taps, tuple(-x for x in p), out)
    return _finish(ConvOp(cin, cout, cpi, cpo, tuple(in_dims), out, s, cin * kk, kk, [sub], dtype, algo_macs=int(np.prod(out)) * kk))


def _transposed_subs(k, s, p, in_dims, y_dims):
    """Sub-convolutions of y[o] = sum_{i,tap: o = i*s - p + tap} w[tap] x[i]  (x has ``in_dims``)."""
    per_axis = []
    for a in range(3):
        classes = []
        for r in range(s[a]):
            t0 = (r + p[a]) % s[a]
            taps_a = list(range(t0, k[a], s[a]))          # tap = t0 + s*m
            nq = (y_dims[a] - r + s[a] - 1) // s[a] if y_dims[a] > r else 0
            if not taps_a or nq <= 0:
                classes.append(None)
                continue
            M = len(taps_a)
            base = (r + p[a]) // s[a]
            # i = q + base - m ; offset index off = M-1-m  -> i = q + (base-(M-1)) + off
            offs = [(M - 1 - m, taps_a[m]) for m in range(M)]
            classes.append((r, nq, base - (M - 1), offs))
        per_axis.append(classes)
    subs = []
    for cz in per_axis[0]:
        for cy in per_axis[1]:
            for cx in per_axis[2]:
                if cz is None or cy is None or cx is None:
                    continue
                taps = [(oz, oy, ox, (tz * k[1] + ty) * k[2] + tx)
                        for oz, tz in cz[3] for oy, ty in cy[3] for ox, tx in cx[3]]
                subs.append(SubConv(taps, (cz[2], cy[2], cx[2]), (cz[1], cy[1], cx[1]),
                                    tuple(s), (cz[0], cy[0], cx[0])))
    return subs


def conv_dgrad_op(cin, cout, k, stride, pad, in_dims, cp_dz, cp_g, dtype=0, cin_total=None):
    """Data gradient of nn.Conv3d(cin->cout): dz (on the conv's output grid) -> g (on ``in_dims``).
    Output positions no tap reaches (strided convs) are NOT written: caller zero-fills g.
    cin_total: the op covers ``cin`` consecutive input channels of a weight with ``cin_total`` of them (the caller
    offsets the weight pointer): gradient of ONE part of a channel-concatenated input into its own dense tensor."""
    k, s, p = _triple(k), _triple(stride), _triple(pad)
    out = tuple((in_dims[a] + 2 * p[a] - k[a]) // s[a] + 1 for a in range(3))
    kk = k[0] * k[1] * k[2]
    subs = _transposed_subs(k, s, p, out, tuple(in_dims))
    # roles swap: "cout" of this op is the conv's cin.  element (co'=ci, ci'=co, tap) = w[co, ci, tap]
    return _finish(ConvOp(cout, cin, cp_dz, cp_g, out, tuple(in_dims), (1, 1, 1), kk, (cin_total or cin) * kk, subs, dtype,
                          algo_macs=int(np.prod(out)) * kk))


def convT_fwd_op(cin, cout, k, stride, pad, in_dims, cpi, cpo, dtype=0):
    """nn.ConvTranspose3d forward (output_padding 0); weight (cin, cout, k, k, k)."""
    k, s, p = _triple(k), _triple(stride), _triple(pad)
    y = tuple((in_dims[a] - 1) * s[a] - 2 * p[a] + k[a] for a in range(3))
    kk = k[0] * k[1] * k[2]
    subs = _transposed_subs(k, s, p, tuple(in_dims), y)
    return _finish(ConvOp(cin, cout, cpi, cpo, tuple(in_dims), y, (1, 1, 1), kk, cout * kk, subs, dtype,
                          algo_macs=min(int(np.prod(in_dims)) * kk, sum(int(np.prod(sb.out_dims)) * len(sb.taps) for sb in subs))))


def convT_dgrad_op(cin, cout, k, stride, pad, in_dims, cp_dz, cp_g, dtype=0):
    """Data gradient of nn.ConvTranspose3d(cin->cout): an ordinary strided correlation of dz
    (on the convT output grid) with the same weights: g[i,ci] = sum w[ci,co,tap] dz[i*s - p + tap, co]."""
    k, s, p = _triple(k), _triple(stride), _triple(pad)
    y = tuple((in_dims[a] - 1) * s[a] - 2 * p[a] + k[a] for a in range(3))
    kk = k[0] * k[1] * k[2]
    taps = [(a, b, c, (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])]
    sub = SubConv(taps, tuple(-x for x in p), tuple(in_dims))
    # op "cout" = convT cin ; element (co'=ci, ci'=co, tap) = w[ci, co, tap]
    return _finish(ConvOp(cout, cin, cp_dz, cp_g, y, tuple(in_dims), s, cout * kk, kk, [sub], dtype,
                          algo_macs=min(int(np.prod(in_dims)) * kk, int(np.prod(y)) * kk)))


# ------------------------------------------------------------------------------------------------ tiling

def _pick_nt(nttot):
    best = None
    for nt in (4, 3, 2, 1):      # fewest passes over the input tile first, then least padding
        padded = -(-nttot // nt) * nt
        key = (padded // nt, padded)
        if best is None or key < best[0]:
            best = (key, nt, padded)
    return best[1], best[2]


ROW_CONFIGS = ((8, 4, 8), (4, 4, 4), (4, 2, 8), (2, 2, 4), (2, 1, 8))   # (MT, TD, TH): 4*MT rows of 16 voxels


def _pick_rows(out_dims, fits):
    """Largest row blocking that suits the volume and whose staged plane fits LDS."""
    qd, qh, _ = out_dims
    for limit in ("budget", "hard"):
        ok = [c for c in ROW_CONFIGS if fits(c, limit)]
        for c in ok:
            if qd >= c[1] and qh >= c[2]:
                return c
        if ok:
            return min(ok, key=lambda c: c[0] * 100 + c[1])
    raise AssertionError("no tile configuration fits LDS")


def _plan_sub(op: ConvOp, sub: SubConv, force_rows=None):
    s = op.stride
    ext = sub.ext
    octs = op.cpi // 8
    opp = 2 if octs % 2 == 0 else 1
    np_planes = 2 if op.dtype in (1, 2) else 1      # f32 (split on load) and bf16 pairs (SP_HL): hi and lo planes
    budget = LDS_BUDGET[op.dtype]

    def tile_dims(c):
        return (c[1] - 1) * s[0] + ext[0], (c[2] - 1) * s[1] + ext[1], 15 * s[2] + ext[2]

    def fits(c, limit):   # one plane at the widest candidate voxel stride must fit
        d = tile_dims(c)
        return d[0] * d[1] * d[2] * (opp + 1) * 16 * np_planes + 4096 <= (budget if limit == "budget" else 156 * 1024)

    mt, td, th = _pick_rows(sub.out_dims, fits) if force_rows is None else force_rows
    # Layers with two or more output tiles and at most four 16-channel input planes: 4x4x16-voxel tiles (MT = 4) instead of
    # 4x8x16.  The staged halo tile shrinks from 70 KB to ~41 KB for two planes, three or four workgroups fit a CU, and
    # their stage / MFMA / store phases overlap (tools/stamp_conv.py: each phase leaves the matrix pipe idle for its own
    # workgroup): 32->32 @58^3 85 -> 79 us, 32->96 @48^3 158 -> 138 us, 64->64 @25^3 56 -> 41 us; with six input planes
    # (96->32) the larger halo re-read costs more than the overlap gains (100 -> 114 us; 1848 -> 2312 us at 168^3), while from
    # eight planes on the input is staged in several channel groups anyway and the small tile wins again (4-scale net:
    # 192->64 @88^3 1167 -> 932 us, 384->128 @48^3 739 -> 557 us, 128->128 @59^3 552 -> 417 us).  SP_PLAN_ROWS_WIDE=MT,TD,TH
    # overrides, SP_PLAN_ROWS_WIDE=0 restores the large tile.
    _force = os.environ.get("SP_PLAN_ROWS_WIDE", "4,4,4")
    if force_rows is None and _force not in ("", "0") and -(-op.cout // 16) >= 2 and op.dtype == 0 and (op.cpi <= 64 or op.cpi >= 128) and s == (1, 1, 1):
        c = tuple(int(v) for v in _force.split(","))
        if fits(c, "hard") and sub.out_dims[0] >= c[1] and sub.out_dims[1] >= c[2]:
            mt, td, th = c
    itd, ith, itw = tile_dims((mt, td, th))
    nvox = itd * ith * itw

    def tap_vox(t):
        return (t[0] * ith + t[1]) * itw + t[2]

    def cost(vs):
        """average ds_read_b128 cycles over the K steps of one plane-group (bank-conflict model)."""
        seq = [(t, oc) for t in sub.taps for oc in range(opp)]
        while len(seq) % 4:
            seq.append(seq[-1])
        tot = 0
        for i in range(0, len(seq), 4):
            addr = []
            for l in range(64):
                t, oc = seq[i + (l >> 4)]
                addr.append((tap_vox(t) + (l & 15) * s[2]) * vs * 16 + oc * 16)
            tot += _b128_cycles(addr)
        return tot / (len(seq) // 4)

    cands = sorted(range(opp, opp + 2), key=lambda vs: (round(cost(vs), 2), vs))
    vs = cands[0]
    vsb = vs * 16
    plane_bytes = (nvox * vsb + 15) // 16 * 16
    # channel groups: as many planes as fit the LDS budget, dividing the octets evenly
    nplanes_total = octs // opp
    ppg = nplanes_total
    while ppg > 1 and (ppg * plane_bytes * np_planes > budget or nplanes_total % ppg):
        ppg -= 1
    opg = ppg * opp
    ngroups = octs // opg
    dma = int((op.dtype == 0 or (op.dtype == 2 and HL_DMA)) and vsb == opp * 16)      # (dtype 2: bf16 pairs, hi and lo tiles side by side)
    if dma:   # the DMA kernel's job table holds 128 (plane, row, segment) jobs per group
        jobs_per_plane = itd * ith * (-(-(itw * opp) // 64))
        while ppg > 1 and (ppg * jobs_per_plane > 128 or nplanes_total % ppg):
            ppg -= 1
        if ppg * jobs_per_plane > 128:
            dma = 0
        opg = ppg * opp
        ngroups = octs // opg
    seq = [(ti, oc) for ti in range(len(sub.taps)) for oc in range(opg)]
    while len(seq) % 4:
        seq.append(None)
    steps = len(seq) // 4
    nt_guess = _pick_nt(-(-op.cout // 16))[0]
    resident = steps in (1, 2, 4, 7, 14) and steps * nt_guess <= 16 and op.dtype != 2      # (pairs: the run-time K loop only)
    zs_steps = steps in (7, 14)       # the z-marching variant keeps up to 14 x 3 weight fragments itself
    if dma and not resident and steps % 2:      # run-time K loop of the DMA kernel works on step pairs
        seq += [None] * 4
        steps += 1
    ktab = np.zeros(steps * 4, dtype=np.int32)
    kmap = np.full(ngroups * steps * 4, -1, dtype=np.int32)
    for i, e in enumerate(seq):
        if e is None:
            continue
        ti, oc = e
        t = sub.taps[ti]
        ktab[i] = tap_vox(t) * vsb + (oc // opp) * plane_bytes + (oc % opp) * 16
        for g in range(ngroups):
            kmap[g * steps * 4 + i] = (t[3] << 16) | (g * opg + oc)
    ktab_bytes = (steps * 16 + 15) // 16 * 16
    tile_bytes = ppg * plane_bytes
    # LDS-DMA staging (bf16 fast path) needs lane-linear planes; zero-fill when taps can leave the volume
    zfill = int(any(sub.o0[a] < 0 or (sub.out_dims[a] - 1) * s[a] + sub.o0[a] + ext[a] > op.in_dims[a]
                    for a in range(3)))
    lds = ktab_bytes + tile_bytes * np_planes + (1024 if dma else 0)
    lds = max(lds, 4 * 16 * 2 * 4)
    assert lds <= 160 * 1024, "LDS plan does not fit: %d bytes" % lds
    sub.tile = dict(MT=mt, TD=td, TH=th, ITD=itd, ITH=ith, ITW=itw, opp=opp, vsb=vsb, plane_bytes=plane_bytes,
                    octs_per_group=opg, ngroups=ngroups, steps_per_group=steps,
                    lo_offset=tile_bytes if np_planes == 2 else 0, lds_bytes=lds, read_cycles=cost(vs),
                    dma=dma, zfill=zfill)
    sub.kmap, sub.ktab = kmap, ktab
    # z-marching ring variant of the DMA kernel (sp_conv_dma.hip, conv_igemm_zs_kernel): one 16-channel plane in, one
    # 16-channel tile out, stride 1, resident weights.  Same K order (same weight fragments); its table holds the
    # in-plane offset inside a (32 + ext_y - 1) x ITW plane slot, with the tap's z index in the low two bits.
    sub.ktab_zs = None
    sub.kmap_zr = sub.ktab_zr = None
    if (dma and op.dtype == 0 and s == (1, 1, 1) and ngroups == 1 and opg == 2 and op.cpi == 16 and zs_steps
            and -(-op.cout // 16) <= 2 and ext[0] <= 3 and mt == 8      # (three output tiles: 359 VGPRs, measured 1.7x slower)
            and sub.out_dims[1] >= 32):
        kz = np.zeros(steps * 4, dtype=np.int32)
        for i, e in enumerate(seq):
            if e is None:
                kz[i] = kz[i - 1] if i else 0          # padding entries carry zero weights: any valid address
                continue
            ti, oc = e
            t = sub.taps[ti]
            kz[i] = ((t[1] * itw + t[2]) * vsb + (oc % opp) * 16) | t[0]
        sub.ktab_zs = kz
        sub.tile["ITH_zs"] = 31 * s[1] + ext[1]
        # row-reuse variant (conv_igemm_zr_kernel): one output tile, 3x3x3.  K steps grouped by dy: for every dy the nine
        # (dz, dx) taps in ONE fixed order, two octets each -> 18 entries = 4.5 steps, padded to 5; step dy*5 + t.  The
        # fragment of step type t read at input row r then serves dy = 0, 1, 2 (output rows r, r-1, r-2).
        if -(-op.cout // 16) == 1 and tuple(ext) == (3, 3, 3) and len(sub.taps) == 27:
            by_dy = {dy: sorted([t for t in sub.taps if t[1] == dy], key=lambda t: (t[0], t[2])) for dy in range(3)}
            if all(len(v) == 9 for v in by_dy.values()) and \
                    all([(t[0], t[2]) for t in by_dy[dy]] == [(t[0], t[2]) for t in by_dy[0]] for dy in range(3)):
                kmap_zr = np.full(15 * 4, -1, dtype=np.int32)
                ktab_zr = np.zeros(5 * 4, dtype=np.int32)
                for dy in range(3):
                    for i, (t, oc) in enumerate([(t, oc) for t in by_dy[dy] for oc in range(2)]):
                        kmap_zr[dy * 20 + i] = (t[3] << 16) | oc
                        if dy == 0:
                            ktab_zr[i] = ((t[2]) * vsb + oc * 16) | t[0]
                ktab_zr[18] = ktab_zr[16]; ktab_zr[19] = ktab_zr[17]      # zero-weight half step: any valid address
                sub.kmap_zr, sub.ktab_zr = kmap_zr, ktab_zr


def _finish(op: ConvOp):
    assert op.cpi % 8 == 0 and op.cpo % 8 == 0 and op.cin <= op.cpi and op.cout <= op.cpo
    op.nt, op.nttot = _pick_nt(-(-op.cout // 16))
    # Tiny output volumes (the FC-like 100 <-> 800 layers of the CAE: 1 x 10 x 10 / 3 x 12 x 12 voxels per sample) give one
    # M tile per sample: with four output tiles per workgroup only a handful of workgroups exist (8 for 800 -> 100 at
    # batch 4).  One output tile per workgroup multiplies the workgroup count by up to four; the input they re-stage is
    # small.  SP_PLAN_NT_SMALL=0 restores the wide tiles.
    nvox_out = max(int(np.prod(sub.out_dims)) for sub in op.subs)      # threshold swept: 512 .. 32768, 8192 best for the CAE
    if os.environ.get("SP_PLAN_NT_SMALL", "1") != "0" and nvox_out <= int(os.environ.get("SP_PLAN_NT_SMALL_VOX", "8192")) and op.dtype == 0:
        op.nt, op.nttot = 1, -(-op.cout // 16)
    for sub in op.subs:
        _plan_sub(op, sub)
    # The parity classes of a transposed / strided-gradient op go out in ONE launch when they share the register blocking
    # (sp_conv3d_igemm_multi): classes whose extents differ by a voxel may pick different row tiles -- settle on the smallest
    # that every class can stage.
    if len(op.subs) > 1 and len({(sb.tile["MT"], sb.tile["TD"], sb.tile["TH"]) for sb in op.subs}) > 1:
        for cand in sorted({(sb.tile["MT"], sb.tile["TD"], sb.tile["TH"]) for sb in op.subs}):
            try:
                for sub in op.subs:
                    _plan_sub(op, sub, force_rows=cand)
                break
            except AssertionError:
                continue
    return op


def wgrad_taps(k, transposed_roles=False):
    k = _triple(k)
    return [(a, b, c, (a * k[1] + b) * k[2] + c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])]


# ------------------------------------------------------------------------------------------------ z-marching plan
# Output-stationary z-marching kernel (csrc/sp_conv_zm.hip): (P input planes of 16 channels, NT output tiles of 16) ->
# (MT rows per wave, ring slots, NW waves per workgroup).  Mirrors sp_conv3d_zm_config (tests/test_cabi.py checks that the
# two agree).  Default: eight waves (two per SIMD -- one wave's epilogue / DMA / LDS instructions issue under its partner's
# MFMAs), four for three input planes; SP_ZM_NW=4: four waves with twice the rows each everywhere (read on both sides; A/B runs).
ZM_CONFIGS_NW4 = {(1, 1): (8, 3, 4), (1, 2): (4, 3, 4), (1, 3): (4, 3, 4), (2, 1): (8, 3, 4), (2, 2): (4, 3, 4), (3, 1): (4, 3, 4)}
ZM_CONFIGS_DEFAULT = {(1, 1): (4, 3, 8), (1, 2): (2, 3, 8), (1, 3): (2, 3, 8), (2, 1): (4, 3, 8), (2, 2): (2, 3, 8), (3, 1): (3, 2, 8)}
ZM_CONFIGS = ZM_CONFIGS_NW4 if os.environ.get("SP_ZM_NW") == "4" else ZM_CONFIGS_DEFAULT
if os.environ.get("SP_ZM_31") == "w4":      # A/B: three input planes on four waves, 16 x 16 tiles, three ring slots (the form up to round 5)
    ZM_CONFIGS = {**ZM_CONFIGS, (3, 1): (4, 3, 4)}
# bf16-pair instances (dtype 2 = SP_HL, the forward convolutions of the "bf16x3" mode): twice the planes per ring slot and hi + lo
# weight fragments in LDS -> smaller tiles / two ring slots where Cin x Cout grows.  Mirrors sp_conv3d_zm_config_hl.
ZM_CONFIGS_HL = {(1, 1): (4, 3, 8), (1, 2): (2, 3, 8), (2, 1): (4, 2, 4), (2, 2): (2, 2, 4), (3, 1): (2, 2, 4)}
ZM_ITW = 18
HL_DMA = bool(int(os.environ.get("SP_HL_DMA", "1")))      # bf16-pair layers without a z-marching instance on the LDS-DMA tiled kernel (0: register-staged)


# (P, NT) instances that exist but lose against their own slices run as teams of one launch (SP_ZM_SPLIT="1,3;..."): measurement knob
ZM_SPLIT = {tuple(int(v) for v in it.split(",")) for it in os.environ.get("SP_ZM_SPLIT", "").split(";") if it}


ZM_TILE = os.environ.get("SP_ZM_TILE", "auto")      # "16": the classic NW MT rows x 16 voxels everywhere (A/B runs)


def zm_tile(ho, wo, nw, mt):
    """(tw, th): the output tile of a z-marching workgroup -- th rows of tw voxels, flattened row-major onto the NW MT column
    groups of 16 (csrc/sp_conv_zm.hip: ConvZmDev.tw / th).  Fewest tiles per plane first (every tile costs the same MFMAs
    whatever its fill: 50-voxel rows run as 25 x 10 instead of 16 x 16 tiles, 25 instead of 16 tiles per plane), then the
    least staged halo, then the classic shape.  Limits: tw th <= 16 NW MT voxels, (tw + 2)(th + 2) <= (NW MT + 2) 18 staged."""
    rows = nw * mt
    if ZM_TILE == "16":
        return 16, rows
    if "x" in ZM_TILE:      # "32x16": that shape wherever it fits (A/B runs), else the automatic choice
        tw, th = (int(v) for v in ZM_TILE.split("x"))
        if tw * th <= 16 * rows and (tw + 2) * (th + 2) <= (rows + 2) * 18 and tw <= 253 and th <= 253:
            return tw, th
    best = None
    for tw in range(4, min(wo, 253) + 1):
        for th in range(2, min(ho, 253) + 1):
            if tw * th > 16 * rows or (tw + 2) * (th + 2) > (rows + 2) * 18:
                continue
            tiles = -(-wo // tw) * -(-ho // th)
            key = (tiles, tiles * (tw + 2) * (th + 2), 0 if (tw, th) == (16, rows) else 1, -tw)
            if best is None or key < best[0]:
                best = (key, (tw, th))
    dflt = -(-wo // 16) * -(-ho // rows)
    # under 10 % fewer tiles: keep the classic shape -- a column group that straddles two rows reads LDS with a 2-voxel skew
    # (bank conflicts): 60 x 60 planes as 20 x 12 tiles (15 instead of 16) measured 8-13 % SLOWER, 50 x 50 as 25 x 10 (10
    # instead of 16) 32 % faster, 88 x 88 as 22 x 22 (16 instead of 18) 14 % faster (profiles/r05_zm_tiles.txt)
    if best is None or best[0][0] > 0.9 * dflt:
        return 16, rows
    return best[1]


def zm_plan(op: ConvOp, tile=None):
    """K tables of the z-marching kernel for a stride-1 3x3x3 op between whole 16-channel tiles, or None.

    One input plane feeds the three output planes above it (taps dz = 0, 1, 2); the K loop of a step runs over the 18 P
    in-plane octets (dy, dx, plane p, octet o) in that order, four per step:
      ktab[s*4 + g]            byte offset of the octet inside a ring slot: (p*ITH*18 + dy*(TW + 2) + dx)*32 + o*16
      kmap[(dz*KS + s)*4 + g]  (source tap << 16) | input octet for sp_conv_prep_weights, -1 for the padding octets
    tile: (TW, TH) output tile of a workgroup (zm_tile; None: chosen from the op's output plane; "classic": NW MT rows of 16).
    """
    if op.dtype not in (0, 2) or tuple(op.stride) != (1, 1, 1) or len(op.subs) != 1:
        return None
    sub = op.subs[0]
    if len(sub.taps) != 27 or tuple(sub.ext) != (3, 3, 3) or tuple(sub.out_stride) != (1, 1, 1) or tuple(sub.out_off) != (0, 0, 0):
        return None
    # whole 16-channel planes / tiles in memory; channels past cin / cout (24 -> 32 in the CAE) carry zero weights and bias
    P_, NT = op.cpi // 16, -(-op.cout // 16)
    if op.cpi % 16 or op.cpo % 16 or op.cin > op.cpi or op.cpo < NT * 16:
        return None
    configs = ZM_CONFIGS_HL if op.dtype == 2 else ZM_CONFIGS      # (pairs: the table addresses the hi planes; the lo planes follow)
    if (P_, NT) not in configs or ((P_, NT) in ZM_SPLIT and op.dtype == 0):
        return None
    MT, nslot, nw = configs[(P_, NT)]
    ith = nw * MT + 2                        # compile-time plane pitch of a ring slot: ITH x 18 voxels
    if tile == "classic":                    # (the pooling epilogue pairs rows inside a wave and lanes inside a row)
        tile = (16, nw * MT)
    tw, th = tile if tile is not None else zm_tile(sub.out_dims[1], sub.out_dims[2], nw, MT)
    assert tw * th <= 16 * nw * MT and (tw + 2) * (th + 2) <= ith * ZM_ITW
    ks = (18 * P_ + 3) // 4
    src = {(t[0], t[1], t[2]): t[3] for t in sub.taps}
    ktab = np.zeros(ks * 4, dtype=np.int32)
    kmap = np.full(3 * ks * 4, -1, dtype=np.int32)
    for e in range(18 * P_):
        t2d, rest = divmod(e, 2 * P_)
        p, o = divmod(rest, 2)
        dy, dx = divmod(t2d, 3)
        ktab[e] = (p * ith * ZM_ITW + dy * (tw + 2) + dx) * 32 + o * 16
        for dz in range(3):
            kmap[dz * ks * 4 + e] = (src[(dz, dy, dx)] << 16) | (p * 2 + o)
    for e in range(18 * P_, ks * 4):
        ktab[e] = ktab[e - 2]                # zero-weight padding octets: any valid, conflict-free address
    return dict(P=P_, NT=NT, MT=MT, NW=nw, TH=th, TW=tw, nslot=nslot, KS=ks, ITH=ith, ktab=ktab, kmap=kmap, nsteps=3 * ks)


# ------------------------------------------------------------------------------------------------ plane-serial z-marching plan
# csrc/sp_conv_zm.hip, PS instances (round 5): (output tiles NT, bf16 pairs?) -> (MT rows per wave, ring slots, waves).  Mirrors
# sp_conv3d_zm_config_ps (tests/test_cabi.py).  The march takes ONE 16-channel plane per sub-step and streams that plane's weight
# fragments through two LDS buffers, so the number of input planes is not limited by LDS.
ZM_CONFIGS_PS = {(1, False): (4, 3, 8), (2, False): (2, 3, 8), (1, True): (4, 2, 8)}
# (P, NT, dtype) with a plain z-marching instance that nevertheless run plane-serial (measured faster); SP_ZM_PSER="" turns all off,
# "all" routes every op with a PS instance there
_ps_env = os.environ.get("SP_ZM_PSER", "3,1,2")      # (bf16 48 -> 16 measured SLOWER plane-serial: 60-MFMA sub-steps are too short for their barrier)
ZM_PSER_PREFER = {tuple(int(v) for v in it.split(",")) for it in _ps_env.split(";") if it and it != "all"}
ZM_PSER_ALL = _ps_env == "all"
ZM_PSER_ON = _ps_env != ""


def zm_pser_plan(op: ConvOp, tile=None):
    """K tables of the plane-serial z-march for a stride-1 3x3x3 op of two or more 16-channel input planes, or None:
      ktab[s*4 + g]                          byte offset of the octet inside a ONE-plane ring slot: (dy*(TW + 2) + dx)*32 + o*16
      kmap[((p*3 + dz)*KS + s)*4 + g]        (source tap << 16) | input octet of channel plane p, -1 for the padding octets
    (KS = 5 steps of four octets cover the 18 in-plane octets of one plane; fragments in memory: [p][(dz KS + s) NT + n])"""
    if not ZM_PSER_ON or op.dtype not in (0, 2) or tuple(op.stride) != (1, 1, 1) or len(op.subs) != 1:
        return None
    sub = op.subs[0]
    if len(sub.taps) != 27 or tuple(sub.ext) != (3, 3, 3) or tuple(sub.out_stride) != (1, 1, 1) or tuple(sub.out_off) != (0, 0, 0):
        return None
    PT, NT = op.cpi // 16, -(-op.cout // 16)
    hl = op.dtype == 2
    if op.cpi % 16 or op.cpo % 16 or op.cin > op.cpi or op.cpo < NT * 16 or PT < 2 or (NT, hl) not in ZM_CONFIGS_PS:
        return None
    MT, nslot, nw = ZM_CONFIGS_PS[(NT, hl)]
    ith = nw * MT + 2
    if tile == "classic":
        tile = (16, nw * MT)
    tw, th = tile if tile is not None else zm_tile(sub.out_dims[1], sub.out_dims[2], nw, MT)
    assert tw * th <= 16 * nw * MT and (tw + 2) * (th + 2) <= ith * ZM_ITW
    ks = 5
    src = {(t[0], t[1], t[2]): t[3] for t in sub.taps}
    ktab = np.zeros(ks * 4, dtype=np.int32)
    kmap = np.full(PT * 3 * ks * 4, -1, dtype=np.int32)
    for e in range(18):
        t2d, o = divmod(e, 2)
        dy, dx = divmod(t2d, 3)
        ktab[e] = (dy * (tw + 2) + dx) * 32 + o * 16
        for p in range(PT):
            for dz in range(3):
                kmap[((p * 3 + dz) * ks) * 4 + e] = (src[(dz, dy, dx)] << 16) | (p * 2 + o)
    for e in range(18, ks * 4):
        ktab[e] = ktab[e - 2]                # zero-weight padding octets: any valid, conflict-free address
    return dict(P=1, PT=PT, NT=NT, MT=MT, NW=nw, TH=th, TW=tw, nslot=nslot, KS=ks, ITH=ith, ktab=ktab, kmap=kmap, nsteps=PT * 3 * ks, pser=True)


# ------------------------------------------------------------------------------------------------ fp8 z-marching plan
# csrc/sp_conv_zm8.hip: (P planes of 16 fp8 input channels, NT output tiles) -> (MT rows per wave, ring slots, waves).
# Mirrors sp_conv3d_zm8_config (tests/test_cabi.py).
ZM8_CONFIGS = {(2, 2): (2, 3, 8), (2, 1): (4, 3, 8), (4, 2): (2, 3, 8), (4, 1): (4, 3, 8), (6, 2): (2, 2, 8), (8, 1): (4, 2, 4)}


def zm8_plan(op: ConvOp):
    """K tables of the fp8 z-marching kernel for a stride-1 3x3x3 op between whole 16-channel planes / tiles, or None.

    One MFMA step takes K = 128 = 8 chunks of 16 fp8 channels; chunk e = (in-plane tap (dy, dx), plane p), tap-major, so that
    the two chunks of a lane group are two planes of one tap (P even):
      ktab[(s*4 + g)*2 + h]             byte offset of the chunk inside a ring slot: ((p*ITH + dy)*18 + dx)*16
      kmap[((dz*KS + s)*4 + g)*2 + h]   (source tap << 16) | input plane for sp_conv_prep_f8, -1 for padding chunks"""
    if op.dtype != 0 or tuple(op.stride) != (1, 1, 1) or len(op.subs) != 1:
        return None
    sub = op.subs[0]
    if len(sub.taps) != 27 or tuple(sub.ext) != (3, 3, 3) or tuple(sub.out_stride) != (1, 1, 1) or tuple(sub.out_off) != (0, 0, 0):
        return None
    P_, NT = op.cpi // 16, -(-op.cout // 16)
    if op.cpi % 16 or op.cpo % 16 or op.cin > op.cpi or op.cpo < NT * 16 or (P_, NT) not in ZM8_CONFIGS:
        return None
    MT, nslot, nw = ZM8_CONFIGS[(P_, NT)]
    ith = nw * MT + 2
    ks = (9 * P_ + 7) // 8
    src = {(t[0], t[1], t[2]): t[3] for t in sub.taps}
    ktab = np.zeros(ks * 8, dtype=np.int32)
    kmap = np.full(3 * ks * 8, -1, dtype=np.int32)
    for e in range(9 * P_):
        t2d, p = divmod(e, P_)
        dy, dx = divmod(t2d, 3)
        ktab[e] = ((p * ith + dy) * ZM_ITW + dx) * 16
        for dz in range(3):
            kmap[dz * ks * 8 + e] = (src[(dz, dy, dx)] << 16) | p
    for e in range(9 * P_, ks * 8):
        ktab[e] = ktab[e - 2]                # zero-weight padding chunks: any valid address
    return dict(P=P_, NT=NT, MT=MT, NW=nw, TH=nw * MT, nslot=nslot, KS=ks, ITH=ith, ktab=ktab, kmap=kmap, nsteps=3 * ks)


def zm8_slices(op: ConvOp):
    """[(first output channel, channels, sub-op)]: the op as one fp8 launch (a single slice) or one launch per 32 output
    channels when it has more output tiles than a kernel instance holds (each launch reads the whole narrow input); None when
    no instance applies."""
    import dataclasses
    if zm8_plan(op) is not None:
        return [(0, op.cout, op)]
    P_ = op.cpi // 16
    width = 32 if (P_, 2) in ZM8_CONFIGS else (16 if (P_, 1) in ZM8_CONFIGS else 0)      # output channels per launch
    if op.dtype != 0 or op.cout % 16 or op.cpi % 16 or not width:
        return None
    out, c0 = [], 0
    while c0 < op.cout:
        cn = min(width, op.cout - c0)
        sub_op = dataclasses.replace(op, cout=cn)
        if zm8_plan(sub_op) is None:
            return None
        out.append((c0, cn, sub_op))
        c0 += cn
    return out


# ------------------------------------------------------------------------------------------------ FC-like layers
FC_MIN_CPI = 256          # input channel pitch from which the split-K kernel is considered
FC_MAX_VOX = 4096         # output voxels per sample up to which it is


def fc_plan(op: ConvOp):
    """Tables of the split-K kernel (csrc/sp_conv_fc.hip) for a single dense bf16 correlation with a deep K and a tiny
    output volume -- the 800 -> 100 transposed convolution and the data gradient of the 100 -> 800 convolution around
    the CAE's latent (Cae3D.py:72-76, 178-180) -- or None.  K order: tap-major, then the input octets four per step."""
    if op.dtype != 0 or len(op.subs) != 1:
        return None
    sub = op.subs[0]
    if tuple(sub.out_stride) != (1, 1, 1) or tuple(sub.out_off) != (0, 0, 0) or tuple(sub.out_dims) != tuple(op.y_dims):
        return None
    octs = op.cpi // 8
    if len(sub.taps) == 1 and op.cpi <= 64:
        # pointwise mode of the same kernel file (conv_pw_kernel): one tap, one or two K steps, no K split, any volume --
        # the 1x1x1 layers at the CAE's tail (Cae3D.py:214-218) and generic classify heads
        spt = -(-octs // 4)
        kmap = np.full(spt * 4, -1, dtype=np.int32)
        t = sub.taps[0]
        kmap[:octs] = (t[3] << 16) | np.arange(octs)
        return dict(ntap=1, spt=spt, nsteps=spt, NT=-(-op.cout // 16), kmap=kmap, taps=np.array([[t[0], t[1], t[2]]], dtype=np.int32),
                    pointwise=True)
    if op.cpi < FC_MIN_CPI or int(np.prod(sub.out_dims)) > FC_MAX_VOX:
        return None
    spt = -(-(-(-octs // 4)) // 4) * 4       # K steps per tap, padded to the kernel's prefetch depth (zero-weight octets)
    ntap = len(sub.taps)
    kmap = np.full(ntap * spt * 4, -1, dtype=np.int32)
    for ti, t in enumerate(sub.taps):
        for o in range(octs):
            kmap[ti * spt * 4 + o] = (t[3] << 16) | o
    taps = np.array([[t[0], t[1], t[2]] for t in sub.taps], dtype=np.int32)
    return dict(ntap=ntap, spt=spt, nsteps=ntap * spt, NT=-(-op.cout // 16), kmap=kmap, taps=taps, pointwise=False)


def zm_pser_slices(op: ConvOp):
    """[(c0, cn, sub-op)]: a bf16-PAIR op with two or more input planes and more output tiles than the plane-serial pair instance
    holds (96 -> 32 of the pair mode: the tiled kernel takes 2.7 x its bf16 time there) as one plane-serial launch per 16 output
    channels, every launch reading the input again; None when the op is not a candidate"""
    import dataclasses
    if not ZM_PSER_ON or op.dtype != 2 or op.cout % 16 or op.cpi % 16 or op.cpi // 16 < 2 or op.cout // 16 < 2 or zm_pser_plan(op) is not None:
        return None
    out = []
    for c0 in range(0, op.cout, 16):
        sub_op = dataclasses.replace(op, cout=16)
        if zm_pser_plan(sub_op) is None:
            return None
        out.append((c0, 16, sub_op))
    return out


def zm_slices(op: ConvOp):
    """Output-channel slices [(c0, cn)] that put an op with MORE output tiles than any z-marching kernel holds onto that
    kernel anyway: one launch per slice of 32 (or 16) output channels, every launch reading the (narrow) input again --
    32 -> 64 forward, 32 -> 96 data gradient of the 4-scale net (tiled kernel 610-690 TFLOP/s, z-marching 1100+).  None when
    the op is not a candidate."""
    if op.dtype != 0 or op.cout % 16 or op.cpi % 16 or op.cpi // 16 > 2 or zm_plan(op) is not None:
        return None
    import dataclasses
    P_, nt = op.cpi // 16, op.cout // 16
    step = 2 if ((P_, 2) in ZM_CONFIGS and nt % 2 == 0) else 1      # (equal slices: they then run as teams of one launch)
    if (P_, step) not in ZM_CONFIGS or nt <= step:
        return None
    out = []
    c0 = 0
    while c0 < op.cout:
        cn = min(step * 16, op.cout - c0)
        sub_op = dataclasses.replace(op, cout=cn)
        if zm_plan(sub_op) is None:
            return None
        out.append((c0, cn, sub_op))
        c0 += cn
    return out
