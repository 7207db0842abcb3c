"""Round-3 host-side tests (no GPU): ADVICE r2 findings and the bench line's new plumbing."""
import io
import json
import os
import sys

import torch

import stroke_prediction_amd  # noqa: F401
from stroke_prediction_amd.common.model.Unet3D import Unet3D
from stroke_prediction_amd.learner.Learner import Learner, _decode_metrics, _encode_metrics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CH = [2, 16, 32, 64, 32, 16, 32, 2]


class _Loader(list):
    batch_size = 2


def test_save_model_leaves_the_live_model_and_its_captured_steps_alone(tmp_path):
    """rank 0's save_model used to move the live model to the CPU and back and clear its captured
    graphs -- the other ranks kept replaying theirs, so the next steps issued different collectives per rank.  The file is
    now written from a copy: storages, flat buffers and the graph cache of the live model are untouched."""
    model = Unet3D(CH)
    flat_p, flat_g = model.flat_buffers()
    ptrs = [p.data_ptr() for p in model.parameters()]
    learner = Learner(_Loader(), None, model, None, None, 1, None, str(tmp_path / "run"))
    learner._graphs["sentinel"] = object()
    learner.save_model()
    assert [p.data_ptr() for p in model.parameters()] == ptrs
    assert model.flat_buffers()[0].data_ptr() == flat_p.data_ptr() and model.flat_buffers()[1].data_ptr() == flat_g.data_ptr()
    assert "sentinel" in learner._graphs
    loaded = torch.load(str(tmp_path / "run_learner.model"), weights_only=False)
    assert type(loaded).__name__ == "Unet3D" and not next(loaded.parameters()).is_cuda
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert torch.equal(a, b), k


def test_metric_history_numbers_lists_like_jsonpickle():
    """jsonpickle 0.9.6 gives lists an id too; the two phase lists precede every DTO."""
    from common.dto.MetricMeasuresDto import MetricMeasuresDto, BinaryMeasuresDto
    b = {"py/object": "common.dto.MetricMeasuresDto.BinaryMeasuresDto", "dc": 0.5, "hd": 1.0, "assd": 2.0, "precision": 0.1,
         "sensitivity": 0.2, "specificity": 0.3}
    m = {"py/object": "common.dto.MetricMeasuresDto.MetricMeasuresDto", "loss": 0.25, "core": b, "penu": {"py/id": 3},
         "lesion": {"py/id": 3}}
    # ids: 1 = the 'training' list, 2 = the MetricMeasuresDto, 3 = its first BinaryMeasuresDto
    hist = _decode_metrics(json.dumps({"training": [m], "validate": []}))
    got = hist["training"][0]
    assert isinstance(got, MetricMeasuresDto) and isinstance(got.penu, BinaryMeasuresDto)
    assert got.penu is got.core and got.lesion is got.core and got.core.hd == 1.0
    again = _decode_metrics(_encode_metrics(hist))
    assert again["training"][0].core.assd == 2.0 and again["validate"] == []


def test_device_pipeline_loaders_do_not_fork_workers():
    """a Compose bound to a device runs HIP kernels -- its DataLoader must stay in the training process."""
    from stroke_prediction_amd.common import data as D

    class DS(list):
        transform = D.Compose([D.ToTensor()], device="cuda")
    ld = D._loader(DS(range(4)), [0, 1, 2, 3], 2, 4, False, False)
    assert ld.num_workers == 0

    class DSHost(list):
        transform = D.Compose([D.ToTensor()], device=None)
    assert D._loader(DSHost(range(4)), [0, 1, 2, 3], 2, 3, False, False).num_workers == 3


def test_bench_offers_only_precision_modes_that_exist():
    """no dangling --dtype choice; every offered mode is one the models accept."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    from stroke_prediction_amd.runtime import lib as L
    for dt in bench.DTYPES:
        assert dt in L.DTYPE_CODES, dt
        assert dt in bench.PEAK_TFLOPS


def test_oracle_fp8_roundings_are_the_ocp_formats_and_the_fp8_conv_has_the_documented_gradients():
    """oracle/nets.py: round_e4m3 / round_e5m2 are torch's float8_e4m3fn / float8_e5m2 casts (saturating), and _F8Conv's
    backward is what csrc/sp_conv_zm8.hip / sp_wgrad_f8.hip compute: dx from e5m2(S dy) and e4m3 weights, dw from the bf16
    tensors or (wgrad8) from the fp8 copies"""
    import torch.nn.functional as F
    from oracle import nets
    g = torch.Generator().manual_seed(2)
    v = torch.randn(4000, generator=g) * torch.logspace(-9, 3, 4000)
    v[:4] = torch.tensor([1e9, -1e9, 0.0, 449.0])
    ref4 = v.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    ref5 = v.clamp(-57344, 57344).to(torch.float8_e5m2).float()
    assert torch.equal(nets.round_e4m3(v), ref4) and torch.equal(nets.round_e5m2(v), ref5)
    x = torch.randn(1, 4, 6, 7, 8, generator=g).requires_grad_(True)
    w = (torch.randn(3, 4, 3, 3, 3, generator=g) * 0.2).requires_grad_(True)
    b = torch.zeros(3, requires_grad=True)
    S = 2.0 ** 10
    dy = torch.randn(1, 3, 4, 5, 6, generator=g) * 1e-2
    for wg8 in (False, True):
        y = nets._F8Conv.apply(x, w, b, S, wg8)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
        dyq = nets.round_e5m2(dy * S) / S
        wq_t = nets.quant_weights_e4m3(w.detach().transpose(0, 1).contiguous()).transpose(0, 1)
        torch.testing.assert_close(gx, F.conv_transpose3d(dyq, wq_t), rtol=1e-5, atol=1e-7)
        xs, ds = (nets.round_e4m3(x.detach()), dyq) if wg8 else (x.detach(), dy)
        ref_w = torch.stack([torch.stack([torch.stack([torch.einsum("bozyx,bizyx->oi", ds, xs[:, :, a:a + 4, c:c + 5, d:d + 6])
                                                        for d in range(3)], -1) for c in range(3)], -2) for a in range(3)], -3)
        torch.testing.assert_close(gw, ref_w, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(gb, dy.sum(dim=(0, 2, 3, 4)))
