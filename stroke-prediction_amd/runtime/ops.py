"""Thin Python drivers over the C ABI: torch owns device memory and the stream; every op below
enqueues hand-written gfx950 kernels through ``libstroke_amd.so`` (no torch compute kernels)."""
import math
import os
import sys
import ctypes as C

import numpy as np
import torch

from . import lib as L
from . import plan as P

class _TorchDt:
    """storage code -> torch dtype of the tensors this thread's current library build works on (lib.use): the 16-bit code is
    bfloat16 in libstroke_amd.so and float16 in the "f16" build"""

    def __getitem__(self, code):
        if code == L.SP_F32:
            return torch.float32
        return torch.float16 if L.current_variant() == "f16" else torch.bfloat16      # (SP_HL: each half of a pair is a 16-bit tensor)


TORCH_DT = _TorchDt()


# bumped whenever parameters may have changed behind torch's back (FusedAdam's kernel writes raw pointers)
PARAM_EPOCH = [0]


def bump_param_epoch():
    PARAM_EPOCH[0] += 1


FUSE_BN_FINALIZE = bool(int(os.environ.get("SP_FUSE_BN_FINALIZE", "1")))   # sp_bn_finalize inside the weight re-pack kernel of the folded layers (one launch less per layer)
SPLIT_G = bool(int(os.environ.get("SP_SPLIT_G1", "1")))   # one-launch split of a concatenating layer's data gradient into two dense tensors (ConvRunner.zm_split_ok)
PSER_FALLBACK = bool(int(os.environ.get("SP_ZM_PSER_FALLBACK", "0")))   # plane-serial march for ops with no whole-set instance (bf16 96 -> 32): 124 us against 95 us on the tiled kernel (off)
HL_PSER_SLICES = bool(int(os.environ.get("SP_HL_PSER_SLICES", "0")))   # pair mode: 96 -> 32 as two plane-serial launches of 16 output channels -- measured: 2 x 134 us against 275 us on the tiled kernel, no gain (off)
FUSE_POOL = bool(int(os.environ.get("SP_FUSE_POOL", "1")))   # MaxPool3d(2) in the epilogue of the down blocks' second convolution (training steps)
FUSE_DZ = bool(int(os.environ.get("SP_FUSE_DZ", "1")))   # the second convolution's data gradient writes the first one's dz (BatchNorm / activation backward in its epilogue)
BN_SUMS_FROM_WGRAD = not os.environ.get("SP_BN_SUMS_DGRAD")   # BatchNorm-backward sums from the weight-gradient accumulator (layers.py)
MATERIALIZE_BN = not os.environ.get("SP_NO_MATERIALIZE_BN")   # padded convs behind a BatchNorm: write the normalised input once, then DMA kernels (layers.py)
WGRAD_PARTS = not os.environ.get("SP_WGRAD_ATOMICS")      # weight-gradient partial blocks + summing finish instead of fp32 atomics
# Row-reuse z-marching kernel for single-tile 3x3x3 layers (conv_igemm_zr_kernel: 2.4 MFMAs per LDS fragment read
# instead of 1).  Opt-in: it removes the LDS-read bound of the 16 -> 16 layers (forward 178 -> 162 us, data gradient
# 157 -> 157 us at 126^3) but those layers sit at the HBM ridge (216 FLOP per byte of activations in + out), and the
# second set of weight fragments costs what the forward gains: 4.00 vs 4.00 ms per step.
USE_ZR = os.environ.get("SP_CONV_ZR", "0") != "0"
USE_ZM_SLICES = bool(int(os.environ.get("SP_ZM_SLICES", "1")))      # ops with too many output tiles for one z-marching launch: a launch per 32-channel slice
ZM_SLICE_MIN_PLANES = int(os.environ.get("SP_ZM_SLICE_MIN_PLANES", "2000"))  # ... when the volume is large enough (one launch per slice: 32->96 @48^3 gains nothing, @166^3 40 %; as teams of one launch: @48^3 147 -> 119 us)
USE_PW_WGRAD = bool(int(os.environ.get("SP_WGRAD_PW", "1")))      # streaming weight-gradient kernel for pointwise layers
PAR_STRIDED = bool(int(os.environ.get("SP_CONV_PAR_STRIDED", "1")))
WGRAD_DMA_STRIDED = bool(int(os.environ.get("SP_WGRAD_DMA_STRIDED", "1")))      # stride-2 / 2x2x2 weight gradients on the LDS-DMA kernel (0: register-staged)
# batched passes: BatchNorm folded per group into the z-marching forward (per-group fragments + a bias table over the border classes,
# sp_conv_prep_folded_groups).  Alone (the normalised copy written later, beside the weight gradient) it measured 7.05 -> 7.18 ms/step:
# the copy's bytes only move to the side stream of the bandwidth-bound backward; with RAW_WGRAD below 6.91 -> 6.51
FOLD_GROUPS = bool(int(os.environ.get("SP_FOLD_GROUPS", "1")))
# ... and the weight gradient of those layers on the RAW input (border-class sums of dz, group-aware folded finish): no normalised copy at all
RAW_WGRAD = bool(int(os.environ.get("SP_RAW_WGRAD", "1")))
USE_PAR = bool(int(os.environ.get("SP_CONV_PAR", "1")))      # parity classes of transposed / strided-gradient ops: one pass over the output (csrc/sp_conv_par.hip)
ZM_GROUPS = bool(int(os.environ.get("SP_ZM_GROUPS", "1")))      # batched passes: one z-marching launch over all BatchNorm groups (0: one per group, tiled data gradients)
ZM_CAE = bool(int(os.environ.get("SP_ZM_CAE", "1")))      # z-marching kernel (ELU epilogue, padding) for the CAE's materialised 3x3x3 layers
USE_MULTI = bool(int(os.environ.get("SP_CONV_MULTI", "1")))      # parity classes of an op in one launch where the kernel allows
USE_FC = bool(int(os.environ.get("SP_CONV_FC", "1")))      # split-K kernel for FC-like layers (deep K, tiny output volume)
USE_PERSIST = bool(int(os.environ.get("SP_CONV_PERSIST", "0")))     # persistent double-buffered conv variant: measured slower than 3 workgroups/CU (272 vs 238 us on 16->16 @126^3), opt-in
WGRAD_ZS = int(os.environ.get("SP_WGRAD_ZS", "1"))   # z-marching ring variant of the DMA weight gradient (0 off, 1 where it pays, 2 wherever it applies)
CAT_PLANAR = bool(int(os.environ.get("SP_CAT_PLANAR", "1")))   # plane-major concat buffers (dense 16-channel planes for the DMA consumers)
USE_ZS = bool(int(os.environ.get("SP_CONV_ZS", "1")))     # z-marching ring conv variant for the 16->16-channel stride-1 layers (-20 %)
USE_DMA = True     # bf16 LDS-DMA conv path (tests flip it to compare both kernels)
# output-stationary z-marching kernel (csrc/sp_conv_zm.hip) for the stride-1 3x3x3 layers between whole 16-channel tiles whose
# volume gives every CU a few planes to march through; SP_CONV_ZM=0 falls back to the tiled / ring kernels
USE_ZM = os.environ.get("SP_CONV_ZM", "1") != "0"
ZM_FUSE_SLICES = bool(int(os.environ.get("SP_ZM_FUSE_SLICES", "1")))      # output-channel slices as workgroup teams of one launch
ZM_MIN_PLANES = int(os.environ.get("SP_CONV_ZM_MIN_PLANES", "1024"))     # (column, plane) pairs per launch below which the march is all prologue
_ZEROS = {}


def zero_page(device):
    """a few readable zero bytes on the device: source of every out-of-volume DMA chunk of the z-marching kernel"""
    key = str(device)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(256, dtype=torch.uint8, device=device)
    return _ZEROS[key]

# optional live kernel timing (bench.py): list of (tag, algorithmic_flops, start_event, end_event)
PROFILE = None


class _Timed:
    def __init__(self, tag, flops, detail=""):
        self.tag, self.flops, self.detail = tag, flops, detail

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.tag, self.flops, self.e0, self.e1, self.detail))
        return False


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("stroke_prediction_amd needs an AMD GPU (gfx950): the hot path has no CPU fallback")


def stream():
    return torch.cuda.current_stream().cuda_stream


# ---- a second stream for the backward pass.  fork(): the side stream waits for everything enqueued so far; join():
# the main stream waits for the side work.  Under hipGraph capture the two become parallel branches of the graph.
# SP_OVERLAP=2 (default): every folded layer's WEIGHT-GRADIENT kernel (+ its finish and the BatchNorm-backward finalize)
# runs on the side stream beside the layer's data-gradient convolution -- both read dz, neither feeds the other, and on
# the small layers (blocks 2-4: 200-500 workgroups on 256 CUs) each alone leaves the chip half empty: 3.98 -> 3.86 ms per
# step.  SP_OVERLAP=1: only the ~5-10 us bookkeeping kernels (finish, finalize, re-packs) go to the side stream --
# measured neutral to slightly slower (they take workgroup slots from the convolution they run beside).  SP_OVERLAP=0:
# one stream.
# Only while a hipGraph is being captured (the fork / join become graph edges): launched eagerly from Python, the stream
# switches and event record / wait pairs cost more CPU time than the overlap saves (4.15 vs 3.98 ms); SP_OVERLAP_EAGER=1
# forces it there too.
_OV = os.environ.get("SP_OVERLAP", "2")
_OV_EAGER = bool(os.environ.get("SP_OVERLAP_EAGER"))


import threading
_tls = threading.local()


class no_fork:
    """with no_fork(): the second stream is not used by what runs inside (this thread).  The concurrent passes of the CAE
    (common/model/Cae3D.py) are parallel branches already; forking a side stream per branch inside a stream capture ends in a
    segmentation fault of hipStreamEndCapture (ROCm 7.2; tools/dbg_cae_capture.py)."""

    def __enter__(self):
        _tls.suppress = getattr(_tls, "suppress", 0) + 1

    def __exit__(self, *exc):
        _tls.suppress -= 1
        return False


def overlap_level():
    if _OV in ("0", "") or getattr(_tls, "suppress", 0):
        return 0
    if not _OV_EAGER and not torch.cuda.is_current_stream_capturing():
        return 0
    return 2 if _OV == "2" else 1


_SIDE = {}


class fork:
    def __init__(self):
        self.main = torch.cuda.current_stream()
        self.on = overlap_level() > 0
        if self.on:
            key = (self.main.device.index, self.main.cuda_stream)      # one side stream per launching stream (concurrent CAE passes)
            if key not in _SIDE:
                _SIDE[key] = torch.cuda.Stream(device=self.main.device)
            self.side = _SIDE[key]
            self.side.wait_stream(self.main)
        self.ctx = None

    def __enter__(self):
        if self.on:
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)


def ptr(t):
    return None if t is None else t.data_ptr()


def cpad(c, m=8):
    return -(-c // m) * m


def alloc_cl(batch, dims, cp, dtype, device, zero=False):
    fn = torch.zeros if zero else torch.empty
    return fn((batch,) + tuple(dims) + (cp,), dtype=TORCH_DT[dtype], device=device)


def _dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


class ConvRunner:
    """One planned convolution-like op (see ``plan.ConvOp``) bound to device tables and weight fragments."""

    def zm_y8_ok(self):
        """the z-marching instance of this runner can also write the e4m3 plane-major copy of its output (run(y8=...))"""
        return bool(self.uses_zm() and self.zms is None and self.zm["P"] == 1 and self.zm["NT"] == 2 and not self.zm.get("pser"))

    def _run_zm(self, a, x_planar, batch, with_stats, st):
        return _run_zm_impl(self, a, x_planar, batch, with_stats, st)

    def __init__(self, op: P.ConvOp, device, share=None, zm_batch=None, zm_tile=None):
        """share: a dict owned by the caller; runners built for the SAME op geometry that pass the same dict use
        one set of packed weights / tables (the 3 encoder and 4 decoder passes of a CAE step).
        zm_batch: the caller promises to run this op with that batch size, without affine-on-load, with plain (or no)
        statistics and a LeakyReLU / identity epilogue: the z-marching kernel is used where a plan exists and the volume
        gives it enough planes, and the weights are packed in ITS K order only.
        zm_tile: plan.zm_plan's tile argument ("classic" for the layers whose epilogue pools)."""
        self.op = op
        self.device = device
        st = share if share is not None else {}
        if "subs" not in st:
            subs = []
            for sub in op.subs:
                t = sub.tile
                nsteps = t["ngroups"] * t["steps_per_group"]
                frag_elems = nsteps * op.nttot * 64 * 8
                hi = torch.empty(frag_elems, dtype=torch.bfloat16, device=device)
                lo = torch.empty(frag_elems, dtype=torch.bfloat16, device=device) if op.dtype in (L.SP_F32, L.SP_HL) else None
                d = dict(sub=sub, kmap=_dev_i32(sub.kmap, device), ktab=_dev_i32(sub.ktab, device),
                         ktab_zs=None if getattr(sub, "ktab_zs", None) is None else _dev_i32(sub.ktab_zs, device),
                         hi=hi, lo=lo, nsteps=nsteps, ktab_zr=None)
                if USE_ZR and getattr(sub, "ktab_zr", None) is not None and op.dtype == L.SP_BF16:
                    # second set of weight fragments in the row-reuse order (15 steps: dy-major)
                    d.update(ktab_zr=_dev_i32(sub.ktab_zr, device), kmap_zr=_dev_i32(sub.kmap_zr, device),
                             hi_zr=torch.empty(15 * op.nttot * 64 * 8, dtype=torch.bfloat16, device=device))
                subs.append(d)
            st["subs"] = subs
            zm = P.zm_plan(op, tile=zm_tile) if (USE_ZM and USE_DMA and zm_batch) else None
            # plane-serial march (round 5): the (P, NT, dtype) triples measured faster there (plan.ZM_PSER_PREFER: the pair mode's 48 -> 16).
            # Ops with more input planes than a whole-set ring holds (bf16 96 -> 32) stay on the tiled kernel: 94 us against 124 us
            # plane-serial in one call (SP_ZM_PSER_FALLBACK=1 routes them here)
            if USE_ZM and USE_DMA and zm_batch and ((zm is None and PSER_FALLBACK) or P.ZM_PSER_ALL or (op.cpi // 16, -(-op.cout // 16), op.dtype) in P.ZM_PSER_PREFER):
                zp = P.zm_pser_plan(op, tile=zm_tile)
                if zp is not None:
                    zm = zp
            if zm is not None:
                cols = -(-op.subs[0].out_dims[1] // zm["TH"]) * -(-op.subs[0].out_dims[2] // zm["TW"])
                if zm_batch * cols * op.subs[0].out_dims[0] < ZM_MIN_PLANES:
                    zm = None       # too few (column, plane) pairs: the march would be all prologue
            if zm is not None:      # its own K order -> its own weight fragments (they replace the tiled kernel's: one re-pack per step)
                zm = dict(zm, ktab_d=_dev_i32(zm["ktab"], device), kmap_d=_dev_i32(zm["kmap"], device),
                          hi=torch.empty(zm["nsteps"] * zm["NT"] * 64 * 8, dtype=torch.bfloat16, device=device),
                          lo=torch.empty(zm["nsteps"] * zm["NT"] * 64 * 8, dtype=torch.bfloat16, device=device) if op.dtype == L.SP_HL else None)
            st["zm"] = zm
            # more output tiles than a z-marching kernel holds: one launch per slice of output channels (plan.zm_slices)
            zms = None
            if zm is None and USE_ZM and USE_DMA and zm_batch and USE_ZM_SLICES and op.dtype == L.SP_BF16:
                sl = P.zm_slices(op)
                cols = -(-op.subs[0].out_dims[1] // 16) * -(-op.subs[0].out_dims[2] // 16) if sl else 0
                pairs = zm_batch * cols * op.subs[0].out_dims[0] if sl else 0
                if sl and pairs >= ZM_SLICE_MIN_PLANES:
                    plans = [(c0, cn, P.zm_plan(sub_op)) for c0, cn, sub_op in sl]
                    # equal slices: one fragment pool and teams of m slices per launch (sp_conv_args.nslices, as the fp8 runner)
                    nt0, n0 = plans[0][2]["NT"], plans[0][2]["nsteps"]
                    fuse_m = 0
                    if ZM_FUSE_SLICES and all(z["NT"] == nt0 and z["nsteps"] == n0 and c0 == i * nt0 * 16 for i, (c0, _, z) in enumerate(plans)):
                        cands = [m for m in range(2, min(16, len(plans)) + 1) if len(plans) % m == 0]
                        good = [m for m in cands if (32 // m) * m >= 29]
                        fuse_m = (max(good) if good else max(cands, key=lambda m: ((32 // m) * m, m))) if cands else 0
                        if fuse_m and pairs < 4 * 8 * (32 // fuse_m):
                            fuse_m = 0
                    frag = n0 * nt0 * 64 * 8
                    pool = torch.empty(len(plans) * frag, dtype=torch.bfloat16, device=device) if fuse_m else None
                    zms = []
                    for i, (c0, cn, z) in enumerate(plans):
                        hi = pool[i * frag:(i + 1) * frag] if fuse_m else torch.empty(z["nsteps"] * z["NT"] * 64 * 8, dtype=torch.bfloat16, device=device)
                        zms.append(dict(z, c0=c0, cn=cn, ktab_d=_dev_i32(z["ktab"], device), kmap_d=_dev_i32(z["kmap"], device), hi=hi,
                                        fuse_m=fuse_m, wstride=frag * 2))
            # ... and bf16-PAIR ops past the plane-serial pair instance's one output tile: a plane-serial launch per 16 output channels
            if zm is None and zms is None and USE_ZM and USE_DMA and zm_batch and USE_ZM_SLICES and op.dtype == L.SP_HL and HL_PSER_SLICES:
                sl = P.zm_pser_slices(op)
                if sl:
                    zms = []
                    for c0, cn, sub_op in sl:
                        z = P.zm_pser_plan(sub_op, tile=zm_tile)
                        n = z["nsteps"] * z["NT"] * 64 * 8
                        zms.append(dict(z, c0=c0, cn=cn, ktab_d=_dev_i32(z["ktab"], device), kmap_d=_dev_i32(z["kmap"], device),
                                        hi=torch.empty(n, dtype=torch.bfloat16, device=device), lo=torch.empty(n, dtype=torch.bfloat16, device=device),
                                        fuse_m=0, wstride=0))
            st["zms"] = zms
            fc = P.fc_plan(op) if (USE_FC and zm is None and zms is None and op.dtype != L.SP_HL) else None
            if fc is not None:      # split-K kernel for FC-like layers: its own (tap-major) K order and fragments
                fc = dict(fc, kmap_d=_dev_i32(fc["kmap"], device), taps_d=_dev_i32(fc["taps"], device),
                          hi=torch.empty(fc["nsteps"] * fc["NT"] * 64 * 8, dtype=torch.bfloat16, device=device), partial=None)
            st["fc"] = fc
            st["bias"] = torch.zeros(max(op.nttot, 0 if zm is None else zm["NT"], 0 if fc is None else fc["NT"]) * 16,
                                     dtype=torch.float32, device=device)
            st["has_bias"] = False
            st["prep_key"] = None
        self._st = st
        self.subs = st["subs"]
        self.bias = st["bias"]
        if "par" not in st:
            st["par"] = self._par_tables(op, device)
        self.par = st["par"]
        self.zm = st.get("zm")
        self.zms = st.get("zms")
        self.zm_batch = zm_batch if (self.zm is not None or self.zms is not None) else None
        self.fc = st.get("fc")

    @staticmethod
    def _par_tables(op, device):
        """gather tables of csrc/sp_conv_par.hip for an op of several parity classes (transposed convolutions, data gradients of
        strided ones): per K slot of every class -- in the order of the class's kmap, i.e. of its packed weight fragments --
        (byte offset of (tap, octet) from the lane's base voxel, the tap's per-axis offsets and the octet); None when
        the op does not go there"""
        # ... and strided convolutions (one class, input stride 2: Cae3D.py:45-64 forward, the data gradient of the transposed
        # layers): the tiled kernel stages a (2 T + 1)^3 halo tile per T^3 outputs there
        if not (USE_PAR and USE_DMA and op.dtype == L.SP_BF16 and (2 <= len(op.subs) <= 8 or (PAR_STRIDED and len(op.subs) == 1 and max(op.stride) > 1))):
            return None
        Hi, Wi = op.in_dims[1], op.in_dims[2]
        rows, gofs = [], [0]
        for sub in op.subs:
            off = {t[3]: t[:3] for t in sub.taps}
            if any(max(o) > 2 or min(o) < 0 for o in off.values()):
                return None
            tab = np.zeros((len(sub.kmap), 2), dtype=np.int32)
            for k, e in enumerate(sub.kmap):
                oz, oy, ox, oc = (0, 0, 0, 0) if e < 0 else off[int(e) >> 16] + (int(e) & 0xffff,)      # padding slots: zero weights
                tab[k, 0] = ((oz * Hi + oy) * Wi + ox) * op.cpi * 2 + oc * 16
                tab[k, 1] = oz | (4 + oy) << 8 | (8 + ox) << 16 | oc << 24
            rows.append(tab)
            gofs.append(gofs[-1] + len(sub.kmap))
        if gofs[-1] * 8 > 60 * 1024:
            return None
        return dict(gtab=torch.from_numpy(np.concatenate(rows)).to(device), gofs=(C.c_int32 * len(gofs))(*gofs))

    def uses_zm(self):
        """the z-marching kernel runs this op (and its weight fragments are the only ones packed)"""
        return self.zm is not None or self.zms is not None

    @staticmethod
    def zm_plan_bn_bwd_ok(z):
        """the z-marching instance of plan z exists with the BatchNorm-backward-sums epilogue (stats_mode 1: sum g, sum g x)"""
        return bool(z is not None and not z.get("pser") and z["NW"] == 8 and z["P"] <= 2 and z["MT"] * z["NT"] == 4 and z["nslot"] == 3)

    def par_ok(self, dtype_out):
        """csrc/sp_conv_par.hip runs this op (all parity classes / the strided convolution in one pass; its epilogue takes
        statistics of either kind per BatchNorm group)"""
        return self.par is not None and not self.uses_zm() and self.fc is None and dtype_out == L.SP_BF16

    def zm_pool_ok(self):
        """run(pool=...) applies: a z-marching instance with the MaxPool3d(2) epilogue ((P, NT) = (1, 1) / (2, 2), rows in pairs per
        wave, the classic 16-voxel-wide tile) -- the second convolutions of the down blocks"""
        z = self.zm
        return bool(FUSE_POOL and z is not None and self.zms is None and not z.get("pser") and (z["P"], z["NT"]) in ((1, 1), (2, 2)) and z["MT"] % 2 == 0
                    and (z["NW"] == 8 or self.op.dtype == L.SP_HL) and z["TW"] == 16 and z["TH"] == z["NW"] * z["MT"]
                    and min(self.op.y_dims) >= 2 and tuple(self.op.subs[0].out_dims) == tuple(self.op.y_dims))

    def zm_split_ok(self):
        """run(y2=..., split_nt=...) applies: the z-marching instance with two output tensors ((P, NT) = (1, 3): 16 -> 48, the data
        gradient of the 3-scale network's last concatenating layer)"""
        z = self.zm
        return bool(SPLIT_G and z is not None and self.zms is None and not z.get("pser") and self.op.dtype == L.SP_BF16 and (z["P"], z["NT"]) == (1, 3) and z["NW"] == 8
                    and tuple(self.op.subs[0].out_dims) == tuple(self.op.y_dims))

    def zm_bn_bwd_ok(self):
        return self.zm is not None and self.op.dtype == L.SP_BF16 and ConvRunner.zm_plan_bn_bwd_ok(self.zm)

    def _pack(self):
        """(kmap, nsteps, hi, lo, NTtot, first output channel, output channels) of every set of fragments the kernel(s) that
        will run this op read; all but the output-channel slices of the z-marching kernel cover the whole op"""
        cout = self.op.cout
        if self.zms is not None:
            return [(z["kmap_d"], z["nsteps"], z["hi"], z.get("lo"), z["NT"], z["c0"], z["cn"]) for z in self.zms]
        if self.uses_zm():
            z = self.zm
            return [(z["kmap_d"], z["nsteps"], z["hi"], z.get("lo"), z["NT"], 0, cout)]
        if self.fc is not None:
            f = self.fc
            return [(f["kmap_d"], f["nsteps"], f["hi"], None, f["NT"], 0, cout)]
        return [(s["kmap"], s["nsteps"], s["hi"], s["lo"], self.op.nttot, 0, cout) for s in self.subs]

    @property
    def has_bias(self):
        return self._st["has_bias"]

    @has_bias.setter
    def has_bias(self, v):
        self._st["has_bias"] = v

    def can_fuse_bn(self):
        """prep(bn=...) exists for this runner: one set of fragments, re-packed by sp_conv_prep_folded"""
        return FUSE_BN_FINALIZE and len(self._pack()) == 1

    def prep(self, w, b=None, fold_scale=None, fold_shift=None, bn=None):
        """Re-pack the current fp32 weights (any layout described by the plan's strides) and bias.
        fold_scale / fold_shift: fold a BatchNorm (x*scale+shift on the input channels) into weights and bias
        -- exact only for un-padded convolutions.
        bn (lib.BnFinArgs, ``can_fuse_bn()``): the BatchNorm finalize runs inside the re-pack kernel (sp_conv_prep_folded_bn), which
        writes fold_scale / fold_shift (= bn.scale / bn.shift) itself."""
        op = self.op
        assert w.dtype == torch.float32 and w.is_contiguous()
        if fold_scale is None:
            # un-folded fragments depend on the weights only: the 3 encoder / 4 decoder passes of a CAE step (and
            # repeated inference calls) reuse them until the parameter changes (torch version counter, or the epoch
            # FusedAdam bumps because its kernel writes through raw pointers)
            key = (w.data_ptr(), w._version, PARAM_EPOCH[0], None if b is None else (b.data_ptr(), b._version))
            if self._st["prep_key"] == key:
                return
            self._st["prep_key"] = key
        else:
            self._st["prep_key"] = None
        packs = self._pack()
        if bn is not None:
            assert len(packs) == 1 and fold_scale is not None and fold_shift is not None
            kmap, nsteps, hi, lo, nttot, _, _ = packs[0]
            ntaps = w.numel() // (op.cin * op.cout)
            L.call("sp_conv_prep_folded_bn", ptr(w), op.w_sco, op.w_sci, op.cout, op.cin, ptr(kmap), nsteps, nttot, ptr(hi), ptr(lo), ntaps,
                   ptr(b), ptr(self.bias), nttot * 16, C.byref(bn), stream())
            self.has_bias = True
            if not self.uses_zm() and self.fc is None:
                self._prep_zr(w, fold_scale)
            return
        if fold_scale is not None and fold_shift is not None and len(packs) == 1:
            kmap, nsteps, hi, lo, nttot, _, _ = packs[0]
            ntaps = w.numel() // (op.cin * op.cout)
            L.call("sp_conv_prep_folded", ptr(w), op.w_sco, op.w_sci, op.cout, op.cin, ptr(kmap), nsteps, nttot,
                   ptr(hi), ptr(lo), ptr(fold_scale), ntaps, ptr(b), ptr(fold_shift), ptr(self.bias), nttot * 16,
                   stream())
            self.has_bias = True
            if not self.uses_zm() and self.fc is None:
                self._prep_zr(w, fold_scale)
            return
        for kmap, nsteps, hi, lo, nttot, c0, cn in packs:      # (a slice: element (co', ci', tap) of it is w[(c0 + co') sCo + ci' sCi + tap])
            L.call("sp_conv_prep_weights", w.data_ptr() + 4 * c0 * op.w_sco, op.w_sco, op.w_sci, cn, op.cin, ptr(kmap), nsteps,
                   nttot, ptr(hi), ptr(lo), ptr(fold_scale), stream())
        if not self.uses_zm() and self.fc is None:
            self._prep_zr(w, fold_scale)
        if fold_shift is not None:
            ntaps = w.numel() // (op.cin * op.cout)
            L.call("sp_conv_fold_bias", ptr(w), op.w_sco, op.w_sci, op.cout, op.cin, ntaps, ptr(b), ptr(fold_shift),
                   ptr(self.bias), op.nttot * 16, stream())
            self.has_bias = True
        elif b is not None:
            self.bias[:op.cout].copy_(b.detach())
            self.has_bias = True

    def _prep_zr(self, w, fold_scale):
        op = self.op
        for s in self.subs:
            if s.get("ktab_zr") is not None:
                L.call("sp_conv_prep_weights", ptr(w), op.w_sco, op.w_sci, op.cout, op.cin, ptr(s["kmap_zr"]), 15, op.nttot,
                       ptr(s["hi_zr"]), None, ptr(fold_scale), stream())

    def run(self, x, y, batch, in_scale=None, in_shift=None, act=L.ACT_NONE, act_param=0.0, stats=None,
            dtype_out=None, use_bias=True, stats_nrep=1, stats_mode=0, aux=None, x_planar=False, group_batch=0, y8=None,
            x_lo=None, y_lo=None, group_fold=None, coef_gstride=0, bnb=None, dz_sums=None, pool=None, y2=None, split_nt=0):
        """x_planar: x (shaped (B, D, H, W, CPi) like any input) is stored plane-major [CPi/16][B][D][H][W][16] -- the concat
        buffers written by upsample2_crop_cat_fwd(planar=True); DMA kernel only.
        y8: plane-major uint8 tensor (runtime/f8.alloc_f8) that receives the e4m3 copy of the output (``zm_y8_ok()`` runners).
        y2 / split_nt (``zm_split_ok()``; data gradients of a concatenating layer): output tiles [0, split_nt) go to y (a tensor of
        split_nt * 16 channels or more), the others to the second dense tensor y2.
        pool = (pooled, pooled_lo | None) (``zm_pool_ok()``): MaxPool3d(2) of the output is written to `pooled` by the same kernel and
        `stats` receives the statistics of the POOLED tensor.
        stats_mode 2 (z-marching data gradients, ``zm_bn_bwd_ok()``): y receives dz = (c0 g + c1 aux + c2) act'(aux) instead of the data
        gradient g -- bnb = dict(sums, nrep, count, gamma, mean, invstd, C, CP, dgamma, dbeta, pscale[, coef]) are sp_bn_bwd_finalize's
        arguments (the kernel finalizes the coefficients itself), dz_sums the (SP_REDUCE_ROWS, CPo) accumulator of sum dz; act /
        act_param describe the activation whose derivative is taken."""
        op = self.op
        assert y8 is None or (self.zm_y8_ok() and not group_batch and use_bias and act in (L.ACT_NONE, L.ACT_LEAKY))
        dtype_out = op.dtype if dtype_out is None else dtype_out
        zm_groups = self.zm is not None and batch == self.zm_batch     # the z-marching kernel flushes its statistics per group itself
        if group_batch and group_batch < batch and stats is not None and (self.uses_zm() and not zm_groups):
            # BatchNorm groups (statistics rows per group) on a kernel that is not group-aware: one launch per group on the
            # contiguous slices of the batch (output-channel slices of the z-marching kernel, the split-K kernel)
            assert in_scale is None and batch % group_batch == 0
            per = stats.numel() // (batch // group_batch)
            for gi in range(batch // group_batch):
                sl = slice(gi * group_batch, (gi + 1) * group_batch)
                self.run(x[sl], y[sl], group_batch, None, None, act, act_param, stats[gi * per:(gi + 1) * per], dtype_out, use_bias,
                         stats_nrep, stats_mode, None if aux is None else aux[sl], x_planar, 0)
            return
        if group_batch and (self.uses_zm()) and batch != self.zm_batch:
            for gi in range(batch // group_batch):       # (no statistics: still one launch per group -- the runner is planned for one)
                sl = slice(gi * group_batch, (gi + 1) * group_batch)
                self.run(x[sl], y[sl], group_batch, None, None, act, act_param, None, dtype_out, use_bias, stats_nrep, stats_mode,
                         None, x_planar, 0)
            return
        assert x.dtype == TORCH_DT[op.dtype] and y.dtype == TORCH_DT[dtype_out]
        if op.dtype == L.SP_HL:      # bf16 pairs: x / y are the hi halves, x_lo / y_lo tensors of the same shape hold the lo halves
            assert dtype_out == L.SP_HL and x_lo is not None and y_lo is not None and x_lo.shape == x.shape and y_lo.shape == y.shape
            assert x_lo.dtype == x.dtype and y_lo.dtype == y.dtype and x_lo.is_contiguous() and y_lo.is_contiguous()
            assert not group_batch and stats_mode == 0 and y8 is None and in_scale is None
        assert tuple(x.shape) == (batch,) + tuple(op.in_dims) + (op.cpi,), (tuple(x.shape), op.in_dims, op.cpi)
        assert tuple(y.shape[:4]) == (batch,) + tuple(op.y_dims) and y.shape[4] >= (op.cpo if y2 is None else split_nt * 16)
        a = L.ConvArgs()
        a.x, a.y = ptr(x), ptr(y)
        a.in_scale, a.in_shift = ptr(in_scale), ptr(in_shift)
        a.bias = ptr(self.bias) if (self.has_bias and use_bias) else None
        a.stats = ptr(stats)
        a.stats_nrep = stats_nrep
        a.stats_mode, a.aux = stats_mode, ptr(aux)
        a.dtype_in, a.dtype_out = op.dtype, dtype_out
        a.B = batch
        a.Di, a.Hi, a.Wi = op.in_dims
        a.CPi = op.cpi
        a.YD, a.YH, a.YW = op.y_dims
        a.CPo = y.shape[4]
        a.Cout = op.cout
        a.sD, a.sH, a.sW = op.stride
        a.NT, a.NTtot = op.nt, op.nttot
        a.act, a.act_param = act, act_param
        a.group_batch = group_batch if (group_batch and group_batch < batch and stats is not None) else 0
        if y2 is not None:
            assert self.zm_split_ok() and 1 <= split_nt < self.zm["NT"] and stats is None and stats_mode == 0 and act == L.ACT_NONE and y8 is None and pool is None
            assert tuple(y2.shape[:4]) == tuple(y.shape[:4]) and y2.shape[4] >= (self.zm["NT"] - split_nt) * 16 and y2.dtype == y.dtype and y2.is_contiguous()
            a.y2, a.split_nt, a.CPo2 = ptr(y2), int(split_nt), y2.shape[4]
        if pool is not None:
            pooled, pooled_lo = pool
            assert self.zm_pool_ok() and stats is not None and stats_mode == 0 and not group_batch and y8 is None and group_fold is None
            assert tuple(pooled.shape) == (batch,) + tuple(d // 2 for d in op.y_dims) + (y.shape[4],) and pooled.dtype == y.dtype and pooled.is_contiguous()
            a.pool_y = ptr(pooled)
            if op.dtype == L.SP_HL:
                assert pooled_lo is not None and pooled_lo.shape == pooled.shape and pooled_lo.is_contiguous()
                a.pool_lo_delta = pooled_lo.data_ptr() - pooled.data_ptr()
        if stats_mode == 2:
            assert self.zm_bn_bwd_ok() and bnb is not None and dz_sums is not None and aux is not None and stats is None and not group_batch
            assert dz_sums.dtype == torch.float64 and dz_sums.numel() == L.SP_REDUCE_ROWS * y.shape[4]
            b = a.bnb
            b.sums, b.gamma, b.mean, b.invstd = ptr(bnb["sums"]), ptr(bnb["gamma"]), ptr(bnb["mean"]), ptr(bnb["invstd"])
            b.dgamma, b.dbeta, b.coef = ptr(bnb.get("dgamma")), ptr(bnb.get("dbeta")), ptr(bnb.get("coef"))
            b.count, b.pscale, b.nrep, b.C, b.CP = float(bnb["count"]), float(bnb.get("pscale", 1.0)), int(bnb["nrep"]), int(bnb["C"]), int(bnb["CP"])
            a.dz_sums = ptr(dz_sums)
        if op.dtype == L.SP_HL:
            a.x_lo_delta, a.y_lo_delta = x_lo.data_ptr() - x.data_ptr(), y_lo.data_ptr() - y.data_ptr()
        if y8 is not None:
            assert (self.has_bias or act != L.ACT_NONE), "y8: the bias / activation epilogue instance only"
            assert y8.dtype == torch.uint8 and tuple(y8.shape) == (y.shape[4] // 16, batch) + tuple(op.y_dims) + (16,)
            a.y8, a.y8_plane, a.y8_scale = ptr(y8), batch * int(np.prod(op.y_dims)) * 16, 1.0
        st = stream()
        if group_fold is not None:      # (fragments, bytes per group, bias tables, floats per group): sp_conv_prep_folded_groups
            assert self.zm is not None and batch == self.zm_batch and group_batch and act == L.ACT_ELU and stats_mode == 0
            gf, gfs, gt, gts = group_fold
            a.group_batch = group_batch
            a.bias_tab, a.bias_tab_gstride, a.wfrag_gstride = ptr(gt), gts, gfs
            return _run_zm_impl(self, a, x_planar, batch, stats is not None, st, wfrag=gf)
        if self.uses_zm():
            assert batch == self.zm_batch and in_scale is None and (stats_mode == 0 or (stats_mode in (1, 2) and self.zm_bn_bwd_ok())) \
                and act in (L.ACT_NONE, L.ACT_LEAKY, L.ACT_ELU), \
                "this runner packed its weights for the z-marching kernel (ConvRunner(zm_batch=...)): batch size, " \
                "affine-on-load, statistics mode and activation must be what was promised"
            if self.zms is not None:
                m = self.zms[0]["fuse_m"]
                for z in (self.zms[::m] if m else self.zms):
                    _run_zm_impl(self, a, x_planar, batch, stats is not None, st, z=z, y=y, stats=stats, use_bias=use_bias)
                return
            return self._run_zm(a, x_planar, batch, stats is not None, st)
        if self.fc is not None:     # (its fragments are the only ones packed: every call of this runner goes there)
            return _run_fc(self, x, y, batch, in_scale, in_shift, act, act_param, stats, dtype_out, use_bias, stats_nrep,
                           stats_mode, aux, st, x_planar, group_batch=group_batch, coef_gstride=coef_gstride)
        par = self.par is not None and in_scale is None and not x_planar and dtype_out == L.SP_BF16 and y8 is None
        multi = (L.ConvArgs * len(self.subs))() if (par or (USE_MULTI and 2 <= len(self.subs) <= 8 and not x_planar)) else None
        for si, s in enumerate(self.subs):
            sub = s["sub"]
            t = sub.tile
            a.wfrag_hi, a.wfrag_lo, a.ktab = ptr(s["hi"]), ptr(s["lo"]), ptr(s["ktab"])
            a.Do, a.Ho, a.Wo = sub.out_dims
            a.osD, a.osH, a.osW = sub.out_stride
            a.ooD, a.ooH, a.ooW = sub.out_off
            a.o0D, a.o0H, a.o0W = sub.o0
            for k in ("TD", "TH", "ITD", "ITH", "ITW", "MT", "ngroups", "octs_per_group", "opp", "vsb",
                      "plane_bytes", "lo_offset", "steps_per_group", "lds_bytes", "zfill"):
                setattr(a, k, t[k])
            a.dma = int(t["dma"] and in_scale is None and USE_DMA)
            a.persist = 0 if (a.group_batch or op.dtype == L.SP_HL) else int(USE_PERSIST)      # (BatchNorm groups, bf16 pairs: the tiled kernel only)
            a.x_plane = 0
            if x_planar:
                assert (a.dma or op.dtype == L.SP_HL) and t["opp"] == 2, "plane-major input: DMA kernel (or the bf16-pair register-staged one) with 16-channel planes only"
                a.x_plane = batch * int(np.prod(op.in_dims)) * 16
                a.persist = 0               # (the persistent variants address channels-last rows)
            if USE_ZS and a.dma and s.get("ktab_zs") is not None and stats_mode == 0 and a.CPo >= 16 and not a.group_batch:
                a.persist, a.ktab, a.ITH_zs = 3, ptr(s["ktab_zs"]), t["ITH_zs"]     # z-marching ring variant
                if s.get("ktab_zr") is not None and not x_planar:
                    a.persist, a.ktab, a.wfrag_hi = 4, ptr(s["ktab_zr"]), ptr(s["hi_zr"])   # ... with row reuse
            if multi is not None:       # the parity classes of one op go out in ONE launch (sp_conv3d_igemm_multi)
                C.memmove(C.byref(multi, si * C.sizeof(L.ConvArgs)), C.byref(a), C.sizeof(L.ConvArgs))
                continue
            # algorithmic FLOPs (plan.ConvOp.algo_macs), shared among the sub-convolutions by the work each issues
            share = int(np.prod(sub.out_dims)) * len(sub.taps) / max(1, op.issued_macs())
            with _Timed("conv_igemm", op.flops(batch) * share,
                        "%d->%d @%s%s" % (op.cin, op.cout, "x".join(map(str, op.in_dims)), " +stats" if stats is not None else "")):
                L.call("sp_conv3d_igemm", C.byref(a), st)
        if multi is not None:
            with _Timed("conv_igemm", op.flops(batch),
                        "%d->%d @%s x%d classes%s" % (op.cin, op.cout, "x".join(map(str, op.in_dims)), len(self.subs),
                                                      " +stats" if stats is not None else "")):
                if par:      # all classes in one pass over the output
                    L.call("sp_conv3d_par", multi, len(self.subs), ptr(self.par["gtab"]), self.par["gofs"], ptr(zero_page(self.device)), st)
                else:
                    L.call("sp_conv3d_igemm_multi", multi, len(self.subs), st)


def wgrad_dma_ok(cpi, cpo, dtype):
    """the DMA weight-gradient kernels apply (whole 16-channel tiles; the tile-count limits are knobs only)"""
    return bool(USE_DMA and dtype == L.SP_BF16 and cpi % 16 == 0 and cpo % 16 == 0
                and -(-cpo // 16) <= int(os.environ.get("SP_WGRAD_DMA_MAXCOT", "64"))
                and -(-cpi // 16) <= int(os.environ.get("SP_WGRAD_DMA_MAXCIT", "64")))


def _run_zm_impl(runner, a, x_planar, batch, with_stats, st, z=None, y=None, stats=None, use_bias=True, wfrag=None):
    op = runner.op
    sliced = z is not None
    z = runner.zm if z is None else z
    sub = op.subs[0]
    if sliced:      # output channels [c0, c0 + cn) of y (full pitch), of the bias and of the statistics rows
        esz = y.element_size()
        a.y = y.data_ptr() + z["c0"] * esz
        a.bias = (runner.bias.data_ptr() + 4 * z["c0"]) if (runner.has_bias and use_bias) else None
        a.stats = None if stats is None else stats.data_ptr() + 16 * z["c0"]
    a.wfrag_hi, a.wfrag_lo, a.ktab = ptr(z["hi"] if wfrag is None else wfrag), ptr(z.get("lo")), ptr(z["ktab_d"])
    if sliced and op.dtype == L.SP_HL:      # (pairs: the lo halves sit at the same delta behind the slice's channels)
        pass
    if wfrag is not None:
        a.bias = None
    a.Do, a.Ho, a.Wo = sub.out_dims
    a.osD, a.osH, a.osW = 1, 1, 1
    a.ooD, a.ooH, a.ooW = 0, 0, 0
    a.o0D, a.o0H, a.o0W = sub.o0
    a.MT, a.NT, a.NTtot = z["MT"], z["NT"], z["NT"]
    a.TD, a.TH, a.ITD, a.ITH, a.ITW = 1, z["TH"], 1, z["TH"] + 2, z["TW"] + 2      # the workgroup's tile (plan.zm_tile)
    a.Cout = z["NT"] * 16            # whole tiles: channels past op.cout have zero weights and bias
    a.dma, a.persist, a.zfill = 1, 5, 0
    a.octs_per_group, a.ngroups, a.opp, a.vsb = 2 * z["P"], 1, 2, 32
    a.pser_planes = z.get("PT", 0) if z.get("pser") else 0
    a.x_plane = (batch * int(np.prod(op.in_dims)) * 16) if x_planar else 0
    m = z.get("fuse_m", 0) if sliced else 0
    a.nslices, a.slice_wfrag_stride = (m, z["wstride"]) if m else (0, 0)
    with _Timed("conv_igemm", op.flops(batch) * (z["cn"] * max(m, 1) / op.cout if sliced else 1.0),
                "%d->%d @%s zm%s%s" % (op.cin, op.cout, "x".join(map(str, op.in_dims)), ((" %d slices in one" % m) if m else " slices") if sliced else "",
                                       " +stats" if with_stats else "")):
        L.call("sp_conv3d_zm", C.byref(a), ptr(zero_page(runner.device)), st)


def _run_fc(runner, x, y, batch, in_scale, in_shift, act, act_param, stats, dtype_out, use_bias, stats_nrep, stats_mode, aux, st,
            x_planar=False, group_batch=0, coef_gstride=0):
    """split-K kernel for FC-like layers (csrc/sp_conv_fc.hip)"""
    op, f = runner.op, runner.fc
    sub = op.subs[0]
    M = batch * int(np.prod(sub.out_dims))
    need = 0 if f["pointwise"] else f["ntap"] * M * f["NT"] * 16
    part = getattr(runner, "_fc_partial", None)       # per runner (not in the shared bank): concurrent passes each need their own
    if need and (part is None or part.numel() < need):
        part = runner._fc_partial = torch.empty(need, dtype=torch.float32, device=runner.device)
    a = L.ConvFcArgs()
    a.x, a.y, a.wfrag = ptr(x), ptr(y), ptr(f["hi"])
    a.in_scale, a.in_shift = ptr(in_scale), ptr(in_shift)
    a.bias = ptr(runner.bias) if (runner.has_bias and use_bias) else None
    a.stats, a.aux, a.partial, a.taps = ptr(stats), ptr(aux), (ptr(part) if need else None), ptr(f["taps_d"])
    a.B = batch
    a.Di, a.Hi, a.Wi = op.in_dims
    a.CPi = op.cpi
    a.Do, a.Ho, a.Wo = sub.out_dims
    a.CPo, a.Cout = y.shape[4], op.cout
    a.sD, a.sH, a.sW = op.stride
    a.o0D, a.o0H, a.o0W = sub.o0
    a.ntap = f["ntap"]
    a.act, a.act_param = act, act_param
    a.stats_mode, a.stats_nrep = stats_mode, stats_nrep
    a.dtype_out = op.dtype if dtype_out is None else dtype_out
    a.x_plane = (batch * int(np.prod(op.in_dims)) * 16) if x_planar else 0
    a.group_batch, a.coef_gstride = (group_batch, coef_gstride) if (group_batch and group_batch < batch) else (0, 0)
    with _Timed("conv_igemm", op.flops(batch), "%d->%d @%s %s%s" % (op.cin, op.cout, "x".join(map(str, op.in_dims)), "pw" if f["pointwise"] else "fc",
                                                                    " +stats" if stats is not None else "")):
        L.call("sp_conv_fc", C.byref(a), st)


class WgradRunner:
    """Weight gradient of a convolution (or, with swapped roles, a transposed convolution)."""

    def __init__(self, cin, cout, k, stride, pad, in_dims, out_dims, cpi, cpo, w_sco, w_sci, dtype, device,
                 nblocks=512):
        k, s, p = P._triple(k), P._triple(stride), P._triple(pad)
        self.cin, self.cout, self.k = cin, cout, k
        self.unpadded = max(p) == 0
        self.w_sco, self.w_sci = w_sco, w_sci
        taps = [(a, b, c) for a in range(k[0]) for b in range(k[1]) for c in range(k[2])]
        self.ntap = len(taps)
        self.taps = _dev_i32(np.array(taps), device)
        self.tapsrc = _dev_i32(np.arange(self.ntap), device)
        self.cot, self.cit = -(-cpo // 16), -(-cpi // 16)
        self.acc = None                      # allocated on first run (the partial-block count depends on the batch)
        self.device = device
        a = L.WgradArgs()
        a.dtype = dtype
        a.Di, a.Hi, a.Wi = in_dims
        a.Do, a.Ho, a.Wo = out_dims
        a.CPi, a.CPo = cpi, cpo
        a.sD, a.sH, a.sW = s
        a.o0D, a.o0H, a.o0W = (-p[0], -p[1], -p[2])
        a.ntap = self.ntap
        a.kD, a.kH, a.kW = k
        a.CoT, a.CiT = self.cot, self.cit
        a.nblocks = nblocks
        # bf16 fast path: un-padded stride-1 3x3x3 convolution -> DMA double-buffered kernel, BatchNorm folded into finish
        # (round 4: also stride 2 and 2x2x2 -- the CAE's strided layers and, with swapped roles, its transposed ones)
        unit = s == (1, 1, 1) and k == (3, 3, 3) and tuple(in_dims) == tuple(d + 2 - 2 * q for d, q in zip(out_dims, p))
        strided = (WGRAD_DMA_STRIDED and s == (2, 2, 2) and k in ((3, 3, 3), (2, 2, 2))
                   and all(i + 2 * q >= (o - 1) * 2 + kk for i, o, q, kk in zip(in_dims, out_dims, p, k)))
        self.dma = bool(USE_DMA and dtype == L.SP_BF16 and (unit or strided) and max(p) <= 2 and cpi % 16 == 0 and cpo % 16 == 0
                        and self.cot <= int(os.environ.get("SP_WGRAD_DMA_MAXCOT", "64")) and self.cit <= int(os.environ.get("SP_WGRAD_DMA_MAXCIT", "64")))   # (limits are knobs: the row-sliding kernel takes any tile counts -- 192->64 @88^3 2155 -> 706 us against the register-staged kernel)
        # pointwise layers (1x1x1, stride 1): streaming kernel of csrc/sp_wgrad_pw.hip, BatchNorm folded into the finish as well
        self.pw = bool(USE_PW_WGRAD and WGRAD_PARTS and dtype == L.SP_BF16 and k == (1, 1, 1) and s == (1, 1, 1) and max(p) == 0
                       and cpi % 8 == 0 and cpo % 8 == 0 and tuple(in_dims) == tuple(out_dims))
        a.dma = int(self.dma)
        if self.dma:
            a.nblocks = int(os.environ.get("SP_WGRAD_BLOCKS", "512"))
            a.tile_rows = int(os.environ.get("SP_WGRAD_ROWS", "0"))
        # DMA kernel: ONE 16-channel input plane per workgroup (grid.z = cin tiles).  With 2-3 planes the staged tile
        # shrinks to 64 voxels and its halo is re-read 6x (48->16 @92^3: 505 -> 372 us, 96->32 @50^3: 363 -> 203 us)
        a.cib = int(os.environ.get("SP_WGRAD_CIB", "1" if self.dma else "0"))
        self.args = a
        self.dtype = dtype

    def folds(self, in_scale):
        """BatchNorm folded into the finish step (raw x in the kernel): exact only without padding"""
        return bool((self.dma or self.pw) and in_scale is not None and self.unpadded)

    def _alloc_acc(self, batch):
        """Accumulator block(s).  WGRAD_PARTS: one block per persistent workgroup, written with plain stores and summed
        by the finish kernel (device-scope atomics from 8 XCDs cost 40-90 us per layer); the workgroup count is sized
        so that every workgroup has >= ~512 output voxels and the (cout, cin) tile grid times it is ~2 per CU."""
        import os
        a = self.args
        total = self.ntap * self.cot * 16 * self.cit * 16
        if WGRAD_PARTS:
            cob = 4 if self.cot >= 4 else (2 if self.cot >= 2 else 1)
            cib = 1 if cob == 4 else min(self.cit, 3)
            if a.cib > 0:
                cib = min(cib, a.cib)
            yz = -(-self.cot // cob) * -(-self.cit // cib)
            vox = batch * a.Do * a.Ho * a.Wo
            # multiple of 8: XCD-aware tile walk (floor 512 voxels: tools/wg_sweep.sh; strided layers stage 64-voxel tiles and are
            # bound by the DMA latency per tile -- 32->100 @7x25x25: 8 workgroups x 36 tiles = 119 us -- : floor 128)
            floor = 512 if (a.sD, a.sH, a.sW) == (1, 1, 1) else int(os.environ.get("SP_WGRAD_STRIDED_FLOOR", "128"))
            nb = max(8, min(512 // yz, vox // floor)) // 8 * 8
            # few output voxels under a LARGE weight tensor (the CAE's 100 -> 800 layer: 1200 voxels, 9.7 MB of dw): every partial
            # block is a whole dw written and read again -- 8 blocks = 77 MB for 0.2 MB of operands.  The tile grid alone fills the chip.
            big = int(os.environ.get("SP_WGRAD_BIG_BLOCKS", "2"))
            if big > 0 and total * 4 >= (4 << 20) and yz >= 64:
                nb = min(nb, big)
            if getattr(self, "groups", 1) > 1:      # group-aware finish: the partial blocks of a group are consecutive
                q = 8 * self.groups // math.gcd(8, self.groups)
                nb = max(q, nb // q * q)
            a.nblocks = int(os.environ.get("SP_WGRAD_BLOCKS", nb))
            a.parts, self.nparts = 1, a.nblocks
            if os.environ.get("SP_WGRAD_DEBUG"):
                print("wgrad %d->%d @%dx%dx%d batch %d: %d partial blocks of %.2f MB (tile grid %d, groups %d)" % (
                    self.cin, self.cout, a.Di, a.Hi, a.Wi, batch, a.nblocks, total * 4 / 1e6, yz, getattr(self, "groups", 1)), file=sys.stderr)
            self.acc = torch.empty(self.nparts * total, dtype=torch.float32, device=self.device)
        else:
            a.parts, self.nparts = 0, 1
            self.acc = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.acc_batch = batch

    def run(self, x, dz, batch, dw, in_scale=None, in_shift=None, dz_scale=None, dz_shift=None, dbias_sums=None,
            dbias_grad=None, nbias=0, bn_w=None, bn_sums=None, bn_nrep=1, defer_finish=False, x_planar=False):
        """dw (fp32, the parameter's own layout) += gradient.  On the DMA path the BatchNorm (in_scale/in_shift) is
        folded into the finish step and needs dbias_sums = sum over voxels of dz per output channel (fp64); there
        bn_sums (with bn_w = the conv weight) also receives the BatchNorm-backward sums of the input, which
        replaces the data-gradient convolution of a layer whose input gradient is not needed."""
        a = self.args
        assert x.dtype == TORCH_DT[self.dtype] and dz.dtype == TORCH_DT[self.dtype]
        assert tuple(x.shape) == (batch, a.Di, a.Hi, a.Wi, a.CPi), (tuple(x.shape), (batch, a.Di, a.Hi, a.Wi, a.CPi))
        assert tuple(dz.shape) == (batch, a.Do, a.Ho, a.Wo, a.CPo), (tuple(dz.shape), (batch, a.Do, a.Ho, a.Wo, a.CPo))
        if self.acc is None or self.acc_batch != batch:
            self._alloc_acc(batch)
        a.x, a.dz, a.dw_acc, a.taps = ptr(x), ptr(dz), ptr(self.acc), ptr(self.taps)
        fold = self.folds(in_scale)
        if fold:
            assert dbias_sums is not None and dz_scale is None
        a.dma = int(self.dma and dz_scale is None and (fold or in_scale is None))   # a padded conv behind a BatchNorm cannot fold: register-staged kernel
        a.in_scale, a.in_shift = (None, None) if fold else (ptr(in_scale), ptr(in_shift))
        a.dz_scale, a.dz_shift = ptr(dz_scale), ptr(dz_shift)
        a.B = batch
        a.zs = int(WGRAD_ZS)
        a.groups = 0
        a.x_plane = 0
        if x_planar:
            assert a.dma and a.cib == 1, "plane-major input: DMA weight-gradient kernel, one plane per workgroup"
            a.x_plane = batch * a.Di * a.Hi * a.Wi * 16
        with _Timed("conv_wgrad", 2 * batch * a.Do * a.Ho * a.Wo * self.ntap * self.cin * self.cout,
                    "%d->%d @%dx%dx%d %s" % (self.cin, self.cout, a.Di, a.Hi, a.Wi, "dma" if a.dma else ("pw" if (self.pw and not a.in_scale) else "reg"))):
            L.call("sp_conv3d_wgrad", C.byref(a), stream())
        finish = lambda: self._finish(fold, dw, in_scale, in_shift, dbias_sums, dbias_grad, nbias, bn_w, bn_sums, bn_nrep)
        if defer_finish:
            return finish          # the caller runs it (e.g. on the side stream, next to the data-gradient conv)
        finish()

    def run_raw(self, x, dz, batch):
        """the kernel only, on the RAW input, into this runner's partial blocks (self.acc, self.nparts): the caller finishes
        (layers.ConvLayer._backward_grouped_raw: sp_wgrad_finish_folded_groups)"""
        a = self.args
        assert (self.dma or self.pw) and x.dtype == TORCH_DT[self.dtype] and dz.dtype == TORCH_DT[self.dtype]
        assert tuple(x.shape) == (batch, a.Di, a.Hi, a.Wi, a.CPi) and tuple(dz.shape) == (batch, a.Do, a.Ho, a.Wo, a.CPo)
        if self.acc is None or self.acc_batch != batch:
            self._alloc_acc(batch)
        assert a.parts == 1 and self.nparts % getattr(self, "groups", 1) == 0
        a.x, a.dz, a.dw_acc, a.taps = ptr(x), ptr(dz), ptr(self.acc), ptr(self.taps)
        a.dma, a.in_scale, a.in_shift, a.dz_scale, a.dz_shift = int(self.dma), None, None, None, None
        a.B, a.zs, a.x_plane = batch, int(WGRAD_ZS), 0
        a.groups = getattr(self, "groups", 1)
        with _Timed("conv_wgrad", 2 * batch * a.Do * a.Ho * a.Wo * self.ntap * self.cin * self.cout,
                    "%d->%d @%dx%dx%d dma raw" % (self.cin, self.cout, a.Di, a.Hi, a.Wi)):
            L.call("sp_conv3d_wgrad", C.byref(a), stream())

    def _finish(self, fold, dw, in_scale, in_shift, dbias_sums, dbias_grad, nbias, bn_w, bn_sums, bn_nrep):
        st = stream()
        if fold:
            L.call("sp_wgrad_finish_folded", ptr(self.acc), self.nparts, ptr(self.tapsrc), self.ntap, self.cot * 16, self.cit * 16,
                   self.cout, self.cin, self.w_sco, self.w_sci, ptr(in_scale), ptr(in_shift), ptr(dbias_sums), ptr(dw),
                   ptr(dbias_grad), ptr(bn_w), ptr(bn_sums), bn_nrep, 0, _dbias_stride(dbias_sums), st)
        else:
            assert bn_sums is None, "BatchNorm sums from the weight gradient need the folded (DMA) path"
            L.call("sp_wgrad_finish", ptr(self.acc), self.nparts, ptr(self.tapsrc), self.ntap, self.cot * 16, self.cit * 16,
                   self.cout, self.cin, self.w_sco, self.w_sci, ptr(dw), ptr(dbias_sums) if dbias_grad is not None else None,
                   ptr(dbias_grad), nbias, _dbias_stride(dbias_sums), st)


_PREP_ITEM = np.dtype([("w", "<u8"), ("sCo", "<i8"), ("sCi", "<i8"), ("Cout", "<i4"), ("Cin", "<i4"), ("kmap", "<u8"),
                       ("nsteps", "<i4"), ("NTtot", "<i4"), ("hi", "<u8"), ("lo", "<u8"), ("fold", "<u8"),
                       ("bias", "<u8"), ("bias_out", "<u8"), ("bias_n", "<i4"), ("bias_pad", "<i4")])   # sp_prep_item
_prep_tables = {}


def prep_batch(pairs):
    """Re-pack the (un-folded) weights of many ConvRunners in ONE launch (sp_conv_prep_weights_batch).
    pairs: [(runner, w)] or [(runner, w, b)] -- with b the layer's bias is copied (zero-padded) in the same launch; runners
    whose fragments are current (same key as ConvRunner.prep(w, b)) are skipped, and after the call every runner's key is
    current, so a later runner.prep(w, b) is a no-op."""
    todo = []
    for item in pairs:
        r, w = item[0], item[1]
        b = item[2] if len(item) > 2 else None
        key = (w.data_ptr(), w._version, PARAM_EPOCH[0], None if b is None else (b.data_ptr(), b._version))
        if r._st["prep_key"] != key:
            todo.append((r, w, b, key))
    if not todo:
        return
    # the cached device table holds raw addresses only: key it on every address it contains, so an entry can only be
    # replayed for runners that own exactly those buffers (ids / addresses recycled after an engine was freed)
    tkey = tuple((w.data_ptr(), 0 if b is None else b.data_ptr(), r.bias.data_ptr(), r.op.w_sco, r.op.w_sci, r.op.cout, r.op.cin, r.op.nttot) +
                 tuple((k.data_ptr(), n, h.data_ptr(), 0 if lo_ is None else lo_.data_ptr(), nt_, c0_, cn_) for k, n, h, lo_, nt_, c0_, cn_ in r._pack()) +
                 tuple((0 if sub.get("ktab_zr") is None else sub["hi_zr"].data_ptr()) for sub in r.subs)
                 for r, w, b, _ in todo)
    tab = _prep_tables.get(tkey)
    if tab is None and torch.cuda.is_current_stream_capturing():
        # a set of runners no eager step has re-packed together (a runner whose packing was settled by the previous backward):
        # building the table is a host-to-device copy, which a capture refuses -- the per-runner launches take every address
        # as a kernel argument and are captured as they are
        for r, w, b, _ in todo:
            r.prep(w, b)
        return
    if tab is None:
        if len(_prep_tables) > 64:
            _prep_tables.clear()
        items = []
        for r, w, b, _ in todo:
            assert w.dtype == torch.float32 and w.is_contiguous()
            first = True
            for kmap, nsteps, hi, lo, nttot, c0, cn in r._pack():
                bias_fields = (0, 0, 0, 0)
                if first and b is not None:        # once per runner: bias[0..cout) + zero padding into the runner's padded copy
                    bias_fields = (b.data_ptr(), r.bias.data_ptr(), r.op.cout, r.bias.numel())
                first = False
                items.append((w.data_ptr() + 4 * c0 * r.op.w_sco, r.op.w_sco, r.op.w_sci, cn, r.op.cin, kmap.data_ptr(), nsteps,
                              nttot, hi.data_ptr(), 0 if lo is None else lo.data_ptr(), 0) + bias_fields)
            for sub in ([] if r.uses_zm() else r.subs):
                if sub.get("ktab_zr") is not None:
                    items.append((w.data_ptr(), r.op.w_sco, r.op.w_sci, r.op.cout, r.op.cin, sub["kmap_zr"].data_ptr(), 15,
                                  r.op.nttot, sub["hi_zr"].data_ptr(), 0, 0, 0, 0, 0, 0))
        arr = np.array(items, dtype=_PREP_ITEM)
        dev = torch.from_numpy(arr.view(np.uint8).copy()).to(todo[0][1].device)
        maxb = max((int(it[6]) * int(it[7]) * 64 + 255) // 256 for it in items)
        tab = _prep_tables[tkey] = (dev, len(items), maxb)
    dev, n, maxb = tab
    L.call("sp_conv_prep_weights_batch", dev.data_ptr(), n, maxb, stream())
    for r, _, b, key in todo:
        r._st["prep_key"] = key
        if b is not None:
            r.has_bias = True


def _dbias_stride(dbias_sums):
    """dbias_sums: [SP_REDUCE_ROWS][CP] replica rows as the elementwise kernels fill them (2-D), or one row (1-D)."""
    if dbias_sums is None or dbias_sums.dim() == 1:
        return 0
    assert dbias_sums.shape[0] == L.SP_REDUCE_ROWS and dbias_sums.is_contiguous(), tuple(dbias_sums.shape)
    return dbias_sums.shape[1]


def reduce_rows(c, n=1, device="cuda"):
    """Zeroed fp64 accumulator for the elementwise kernels: SP_REDUCE_ROWS replica rows of c channels x n sums."""
    return torch.zeros((L.SP_REDUCE_ROWS, c) if n == 1 else (L.SP_REDUCE_ROWS, c, n), dtype=torch.float64, device=device)


# ------------------------------------------------------------------------------------------------ elementwise drivers

def ncdhw_to_cl(src, dst, dtype):
    B, Cc = src.shape[:2]
    dhw = int(np.prod(src.shape[2:]))
    L.call("sp_ncdhw_to_cl", ptr(src), ptr(dst), dtype, B, Cc, dhw, dst.shape[-1], stream())


def cl_to_ncdhw(src, dst, dtype):
    B, Cc = dst.shape[:2]
    dhw = int(np.prod(dst.shape[2:]))
    L.call("sp_cl_to_ncdhw", ptr(src), ptr(dst), dtype, B, Cc, dhw, src.shape[-1], stream())


def bn_stats(x, dtype, sums):
    nvox = x.numel() // x.shape[-1]
    L.call("sp_bn_stats", ptr(x), dtype, nvox, x.shape[-1], ptr(sums), stream())


def bn_finalize(sums, count, gamma, beta, rmean, rvar, momentum, eps, training, c, cp, scale, shift, mean, invstd,
                nrep=1):
    L.call("sp_bn_finalize", ptr(sums), nrep, float(count), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), momentum, eps,
           int(training), c, cp, ptr(scale), ptr(shift), ptr(mean), ptr(invstd), stream())


def bn_bwd_reduce(g, x, dtype, sums):
    nvox = x.numel() // x.shape[-1]
    L.call("sp_bn_bwd_reduce", ptr(g), ptr(x), dtype, nvox, x.shape[-1], ptr(sums), stream())


def bn_bwd_finalize(sums, count, gamma, mean, invstd, c, cp, dgamma, dbeta, coef, nrep=1, pscale=1.0):
    L.call("sp_bn_bwd_finalize", ptr(sums), nrep, float(count), ptr(gamma), ptr(mean), ptr(invstd), c, cp, ptr(dgamma),
           ptr(dbeta), ptr(coef), pscale, stream())


def _q8_args(q8, nvox):
    """q8 = (plane-major fp8 tensor, format, scale): the fp8 shadow output of an elementwise kernel (runtime/f8.py)"""
    t, fmt, scale = q8
    assert t.dtype == torch.uint8 and t.numel() == t.shape[0] * nvox * 16, (tuple(t.shape), nvox)
    return ptr(t), nvox * 16, int(fmt), float(scale)


def bn_act_bwd(g, y, coef, dtype, act, act_param, dz, dbias, q8=None, group_vox=0, cls=None, y8=None):
    """cls = (group batch, (padD, padH, padW), class sums): the layer whose dz this forms reads the RAW input in its weight gradient
    (ConvLayer.raw_wgrad): the border-class sums of dz come out of the same pass (sp_bn_act_bwd_groups_cls).
    y8: the e4m3 plane-major copy of y (runtime/f8.py: alloc_f8) -- read INSTEAD of y (fp8 mode: y was not stored; y gives the shape)"""
    nvox = y.numel() // y.shape[-1]
    if y8 is not None:
        assert cls is None and not group_vox and dtype == L.SP_BF16 and y8.dtype == torch.uint8 and y8.numel() == nvox * y.shape[-1]
        q = _q8_args(q8, nvox) if q8 is not None else (None, 0, 0, 1.0)
        L.call("sp_bn_act_bwd_y8", ptr(g), ptr(y8), nvox * 16, ptr(coef), nvox, y.shape[-1], act, act_param, ptr(dz), ptr(dbias), *q, stream())
        return
    if cls is not None:
        gb, pads, sums = cls
        assert q8 is None and coef is not None and y.dim() == 5
        B, D, H, W, CP = y.shape
        L.call("sp_bn_act_bwd_groups_cls", ptr(g), ptr(y), ptr(coef), dtype, B, D, H, W, CP, act, act_param, ptr(dz), ptr(dbias), gb,
               pads[0], pads[1], pads[2], ptr(sums), stream())
        return
    if group_vox:       # coef is a [G][3][CP] table, one per group of group_vox consecutive voxels (batched CAE passes)
        assert q8 is None and coef is not None
        L.call("sp_bn_act_bwd_groups", ptr(g), ptr(y), ptr(coef), dtype, nvox, y.shape[-1], act, act_param, ptr(dz), ptr(dbias),
               group_vox, stream())
        return
    if q8 is not None:
        L.call("sp_bn_act_bwd_q8", ptr(g), ptr(y), ptr(coef), dtype, nvox, y.shape[-1], act, act_param, ptr(dz),
               ptr(dbias), *_q8_args(q8, nvox), stream())
        return
    L.call("sp_bn_act_bwd", ptr(g), ptr(y), ptr(coef), dtype, nvox, y.shape[-1], act, act_param, ptr(dz),
           ptr(dbias), stream())


def maxpool2_fwd(x, y, dtype, stats=None, q8=None, x8=None):
    """x8: the e4m3 plane-major copy of x, read INSTEAD of x (fp8 mode: x was not stored; x gives the shape)"""
    B, D, H, W, CP = x.shape
    if x8 is not None:
        assert dtype == L.SP_BF16 and x8.dtype == torch.uint8 and x8.numel() == x.numel()
        q = _q8_args(q8, y.numel() // CP) if q8 is not None else (None, 0, 0, 1.0)
        L.call("sp_maxpool2_fwd_x8", ptr(x8), B * D * H * W * 16, ptr(y), B, D, H, W, CP, ptr(stats), *q, stream())
        return
    if q8 is not None:
        L.call("sp_maxpool2_fwd_q8", ptr(x), ptr(y), dtype, B, D, H, W, CP, ptr(stats), *_q8_args(q8, y.numel() // CP), stream())
        return
    L.call("sp_maxpool2_fwd", ptr(x), ptr(y), dtype, B, D, H, W, CP, ptr(stats), stream())


def upsample2_fwd(x, y, dtype, stats=None):
    B, D, H, W, CP = x.shape
    L.call("sp_upsample2_fwd", ptr(x), ptr(y), dtype, B, D, H, W, CP, y.shape[-1], ptr(stats), stream())


def upsample2_crop_cat_fwd(low, skip, cat, dtype, stats=None, planar=False, q8=None, store=True, skip8=None):
    """cat = concat(upsample2(low), centre_crop(skip)) in one pass (+ per-channel (sum, sum^2) of cat into stats).
    planar: cat (same shape) is written plane-major [C/16][B][D][H][W][16] for the DMA consumers (x_planar=True).
    store=False (with q8): the 16-bit tensor is not written -- every reader takes the fp8 copy.
    skip8 (with q8): the e4m3 plane-major copy of skip, read INSTEAD of skip (fp8 mode: skip was not stored; skip gives the shape)"""
    B, D, H, W, CPu = low.shape
    _, Ds, Hs, Ws, CPs = skip.shape
    assert tuple(cat.shape) == (B, 2 * D, 2 * H, 2 * W, CPu + CPs), (tuple(cat.shape), tuple(low.shape), tuple(skip.shape))
    if skip8 is not None:
        assert q8 is not None and planar and CPu % 16 == 0 and CPs % 16 == 0 and dtype == L.SP_BF16 and skip8.numel() == skip.numel()
        L.call("sp_upsample2_crop_cat_fwd_q8s8", ptr(low), CPu, ptr(skip8), B * Ds * Hs * Ws * 16, CPs, ptr(cat) if store else None, CPu + CPs,
               B, D, H, W, Ds, Hs, Ws, B * 8 * D * H * W * 16, ptr(stats), *_q8_args(q8, B * 8 * D * H * W), stream())
        return
    if q8 is not None:
        assert planar and CPu % 16 == 0 and CPs % 16 == 0
        L.call("sp_upsample2_crop_cat_fwd_q8", ptr(low), CPu, ptr(skip), CPs, ptr(cat) if store else None, CPu + CPs, dtype, B, D, H, W, Ds, Hs, Ws,
               B * 8 * D * H * W * 16, ptr(stats), *_q8_args(q8, B * 8 * D * H * W), stream())
        return
    assert store
    L.call("sp_upsample2_crop_cat_fwd", ptr(low), CPu, ptr(skip), CPs, ptr(cat), CPu + CPs, dtype, B, D, H, W, Ds, Hs, Ws,
           (B * 8 * D * H * W * 16) if planar else 0, ptr(stats), stream())


def crop_copy(src, dst, c0, dtype, stats=None):
    B, Ds, Hs, Ws, CPs = src.shape
    _, Dd, Hd, Wd, CPd = dst.shape
    L.call("sp_crop_copy", ptr(src), ptr(dst), dtype, B, Ds, Hs, Ws, CPs, Dd, Hd, Wd, CPd, c0, ptr(stats), stream())


def pool_skip_act_bwd(y, gp, coefp, cat, gs, coefs, cs0, dtype, act, act_param, dz, dbias, coef_c0=0, coef_stride=0, q8=None, y8=None):
    """gs: gradient tensor holding the skip part in channels [cs0, cs0+CP) (pitch gs.shape[-1]); coefs indexed with
    (coef_c0, coef_stride) when the gradient is a dense tensor of the skip part only (else like the gradient).
    y8: the e4m3 plane-major copy of y, read INSTEAD of y (fp8 mode: y was not stored; y gives the shape)"""
    B, D, H, W, CP = y.shape
    if gs is not None:
        _, Dc, Hc, Wc, CPcat = gs.shape
    else:
        Dc = Hc = Wc = CPcat = 0
    if y8 is not None:
        assert dtype == L.SP_BF16 and y8.dtype == torch.uint8 and y8.numel() == y.numel()
        q = _q8_args(q8, B * D * H * W) if q8 is not None else (None, 0, 0, 1.0)
        L.call("sp_pool_skip_act_bwd_y8", ptr(y8), B * D * H * W * 16, ptr(gp), ptr(coefp), ptr(gs), ptr(coefs), cs0, CPcat, coef_c0, coef_stride,
               B, D, H, W, CP, Dc, Hc, Wc, act, act_param, ptr(dz), ptr(dbias), *q, stream())
        return
    if q8 is not None:
        L.call("sp_pool_skip_act_bwd_q8", ptr(y), ptr(gp), ptr(coefp), ptr(cat), ptr(gs), ptr(coefs), cs0, CPcat, coef_c0, coef_stride,
               dtype, B, D, H, W, CP, Dc, Hc, Wc, act, act_param, ptr(dz), ptr(dbias), *_q8_args(q8, B * D * H * W), stream())
        return
    L.call("sp_pool_skip_act_bwd", ptr(y), ptr(gp), ptr(coefp), ptr(cat), ptr(gs), ptr(coefs), cs0, CPcat, coef_c0, coef_stride,
           dtype, B, D, H, W, CP, Dc, Hc, Wc, act, act_param, ptr(dz), ptr(dbias), stream())


def upsample2_act_bwd(y, cat, g, coef, dtype, act, act_param, dz, dbias, coef_stride=0, q8=None):
    B, D, H, W, CP = y.shape
    if q8 is not None:
        L.call("sp_upsample2_act_bwd_q8", ptr(y), ptr(cat), ptr(g), ptr(coef), g.shape[-1], coef_stride, dtype, B, D, H, W, CP, act,
               act_param, ptr(dz), ptr(dbias), *_q8_args(q8, B * D * H * W), stream())
        return
    L.call("sp_upsample2_act_bwd", ptr(y), ptr(cat), ptr(g), ptr(coef), g.shape[-1], coef_stride, dtype, B, D, H, W, CP, act,
           act_param, ptr(dz), ptr(dbias), stream())


def out_grad_to_cl(dout, out, dtype, act, act_param, dz, dbias):
    B, Cc = out.shape[:2]
    dhw = int(np.prod(out.shape[2:]))
    L.call("sp_out_grad_to_cl", ptr(dout), ptr(out), B, Cc, dhw, dz.shape[-1], dtype, act, act_param, ptr(dz),
           ptr(dbias), stream())


def add_f64_to_f32(src, dst, n, scale=1.0):
    L.call("sp_add_f64_to_f32", ptr(src), ptr(dst), n, scale, stream())


def lerp_batch(c, p, step, out, dtype):
    B = c.shape[0]
    L.call("sp_lerp_batch", ptr(c), ptr(p), ptr(step), ptr(out), dtype, B, c.numel() // B, stream())


def axpby(x, y, out, dtype, a, b):
    L.call("sp_axpby", ptr(x), ptr(y), ptr(out), dtype, x.numel(), a, b, stream())


def adam_step_flat(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    L.call("sp_adam_step_flat", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, wd, step,
           grad_scale, stream())
