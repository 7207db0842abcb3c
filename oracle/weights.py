"""Deterministic weights / inputs for parity tests (test infrastructure).

Every tensor is drawn from its own numpy ``PCG64`` stream keyed by
``(seed, state_dict name)`` so that the reference modules (golden generation),
this oracle and the HIP path can all be loaded with bit-identical fp32 values
without shipping weight files.  BatchNorm affine parameters and running
statistics are deliberately non-trivial (the reference initialises them to
1/0, which would hide scale/shift bugs).

State-dict key layout follows the reference modules:
``common/model/Unet3D.py:14-54`` (``block{1..5}.bn_conv_relu_2x.{0,1,3,4}``,
``classify.{0,2}``) and ``common/model/Cae3D.py:39-76,176-220``
(``encoder.{0,1,3,4,...}``, ``decoder.{...}``).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def _bn_entries(prefix, c):
    return [(prefix + ".weight", (c,), "bn_w"), (prefix + ".bias", (c,), "bn_b"),
            (prefix + ".running_mean", (c,), "bn_rm"), (prefix + ".running_var", (c,), "bn_rv"),
            (prefix + ".num_batches_tracked", (), "bn_nbt")]


def _conv_entries(prefix, cout, cin, k, transposed=False):
    shape = (cin, cout) + k if transposed else (cout, cin) + k
    # torch's fan_in convention uses dim 1 of the weight for both kinds
    fan_in = shape[1] * int(np.prod(k))
    return [(prefix + ".weight", shape, ("conv_w", fan_in)), (prefix + ".bias", (cout,), ("conv_b", fan_in))]


def unet_spec(channels):
    """(name, shape, kind) list in state_dict order: ``Unet3D.py:31-54`` for 8 channel counts (three scales),
    ``LargeUnet3D`` ``Unet3D.py:88-116`` for 10 (four scales)."""
    S = (len(channels) - 2) // 2
    n_in, w, bc, ncls = channels[0], list(channels[1:2 * S]), channels[-2], channels[-1]
    pairs = [(n_in if i == 1 else w[i - 2], w[i - 1]) for i in range(1, S + 1)]
    pairs += [(w[u - 2] + w[2 * S - u - 1], w[u - 1]) for u in range(S + 1, 2 * S)]
    spec = []
    for i, (ci, co) in enumerate(pairs, start=1):
        p = "block%d.bn_conv_relu_2x." % i
        spec += _bn_entries(p + "0", ci) + _conv_entries(p + "1", co, ci, (3, 3, 3))
        spec += _bn_entries(p + "3", co) + _conv_entries(p + "4", co, co, (3, 3, 3))
    spec += _conv_entries("classify.0", bc, w[-1], (1, 1, 1))
    spec += _conv_entries("classify.2", ncls, bc, (1, 1, 1))
    return spec


# (kind, cin_key, cout_key, kernel, stride, padding) -- Cae3D.py:39-76
ENC_LAYERS = [
    ("conv", "in", "o", 3, 1, (1, 0, 0)), ("conv", "o", "o", 3, 1, (1, 0, 0)),
    ("conv", "o", "d2", 3, 2, (1, 1, 1)),
    ("conv", "d2", "d2", 3, 1, (1, 0, 0)), ("conv", "d2", "d2", 3, 1, (1, 0, 0)),
    ("conv", "d2", "d4", 3, 2, (1, 1, 1)),
    ("conv", "d4", "d4", 3, 1, (1, 0, 0)), ("conv", "d4", "d4", 3, 1, (1, 0, 0)),
    ("conv", "d4", "d8", 3, 2, (0, 0, 0)),
    ("conv", "d8", "fc", 3, 1, (0, 0, 0)),
]
# Cae3D.py:176-220; last entry ends in Sigmoid instead of ELU
DEC_LAYERS = [
    ("convT", "fc", "d8", 3, 1, (0, 0, 0)), ("convT", "d8", "d4", 3, 2, (0, 0, 0)),
    ("conv", "d4", "d4", 3, 1, (1, 2, 2)), ("conv", "d4", "d2", 3, 1, (1, 2, 2)),
    ("convT", "d2", "d2", 2, 2, (0, 0, 0)),
    ("conv", "d2", "d2", 3, 1, (1, 2, 2)), ("conv", "d2", "o", 3, 1, (1, 2, 2)),
    ("convT", "o", "o", 2, 2, (0, 0, 0)),
    ("conv", "o", "o", 3, 1, (1, 2, 2)), ("conv", "o", "o", 3, 1, (1, 2, 2)),
    ("conv", "o", "o", 1, 1, (0, 0, 0)), ("conv", "o", "cls", 1, 1, (0, 0, 0)),
]


def cae_channel_map(channels):
    """``CaeBase.__init__`` (Cae3D.py:14-26): 7-int list -> named widths."""
    return {"in": channels[0], "o": channels[1], "d2": channels[2], "d4": channels[3],
            "d8": channels[4], "fc": channels[5], "cls": channels[-1]}


def _cae_half_spec(prefix, layers, channels):
    cm = cae_channel_map(channels)
    spec = []
    for i, (kind, ci, co, k, _s, _p) in enumerate(layers):
        spec += _bn_entries("%s.%d" % (prefix, 3 * i), cm[ci])
        spec += _conv_entries("%s.%d" % (prefix, 3 * i + 1), cm[co], cm[ci], (k, k, k), transposed=(kind == "convT"))
    return spec


def enc_spec(channels):
    return _cae_half_spec("encoder", ENC_LAYERS, channels)


def dec_spec(channels):
    return _cae_half_spec("decoder", DEC_LAYERS, channels)


def cae_spec(channels):
    return [("enc." + n, s, k) for n, s, k in enc_spec(channels)] + \
           [("dec." + n, s, k) for n, s, k in dec_spec(channels)]


def make_state_dict(spec, seed=0, dtype=torch.float32):
    sd = OrderedDict()
    for name, shape, kind in spec:
        g = _rng(seed, name)
        if kind == "bn_w":
            a = g.uniform(0.5, 1.5, shape)
        elif kind == "bn_b":
            a = g.normal(0.0, 0.1, shape)
        elif kind == "bn_rm":
            a = g.normal(0.0, 0.1, shape)
        elif kind == "bn_rv":
            a = g.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            sd[name] = torch.tensor(0, dtype=torch.long)
            continue
        else:
            bound = 1.0 / np.sqrt(kind[1])
            a = g.uniform(-bound, bound, shape)
        sd[name] = torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(dtype)
    return sd


def unet_inputs(batch, size, seed=0, n_in=2, out_size=None, scales=3):
    """images ~N(0,1) (B,n_in,*size); labels (B,2,*out) = U(0,1)>0.7."""
    if isinstance(size, int):
        size = (size,) * 3
    g = _rng(seed, "unet.images")
    x = g.standard_normal((batch, n_in) + tuple(size)).astype(np.float32)
    out = out_size or tuple(unet_out_size(s, scales) for s in size)
    gl = _rng(seed, "unet.labels")
    y = (gl.uniform(0, 1, (batch, 2) + tuple(out)) > 0.7).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(y)


def unet_out_size(n, scales=3):
    """valid 3x3x3 x2 per block, pool/2, upsample x2 (scales=4: 256 -> 164)."""
    m = n
    for _ in range(scales - 1):
        m = (m - 4) // 2
    m -= 4
    for _ in range(scales - 1):
        m = 2 * m - 4
    return m


def cae_inputs(batch, d=28, hw=128, seed=0):
    """Binary blob labels (B,3,d,hw,hw) + clinical (B,5,1,1,1) double.

    Blobs: threshold of separably box-smoothed noise; core subset of penumbra,
    lesion in between (loosely mimics the data contract of ``data.py:79-99``).
    """
    g = _rng(seed, "cae.labels")
    n = g.standard_normal((batch, d, hw, hw)).astype(np.float32)
    t = torch.from_numpy(n)[:, None]
    k = 9
    for _ in range(2):
        t = torch.nn.functional.avg_pool3d(t, (3, k, k), stride=1, padding=(1, k // 2, k // 2))
    t = t[:, 0]
    t = t / t.std()
    core = (t > 1.2).float()
    lesion = (t > 0.8).float()
    penu = (t > 0.4).float()
    labels = torch.stack([core, penu, lesion], dim=1).contiguous()
    gc = _rng(seed, "cae.clinical")
    clin = np.zeros((batch, 5, 1, 1, 1), dtype=np.float64)
    clin[:, 0, 0, 0, 0] = gc.uniform(0.5, 4.0, batch)   # tO->tA
    clin[:, 1, 0, 0, 0] = gc.uniform(0.5, 5.0, batch)   # tA->tR
    clin[:, 2:, 0, 0, 0] = gc.uniform(0, 1, (batch, 3))
    return labels, torch.from_numpy(clin)
