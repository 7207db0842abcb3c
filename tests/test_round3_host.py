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
