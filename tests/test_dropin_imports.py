"""Import-level drop-in: with ``stroke-prediction_amd/`` at the head of ``sys.path`` the
import blocks of the reference's two named scripts resolve, the parsers take the reference's command lines, the loader
factories honour the batch-dict contract, and ``.model`` / ``.optim`` / ``.json`` files written by the reference load
(SURVEY 8 row N3).  CPU only; GPU behaviour of the same objects is in tests/test_gpu_dropin.py."""
import ast
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

import stroke_prediction_amd  # noqa: E402,F401

# the import statements of /root/reference/train_shape_reconstruction.py:1-5 and train_unet_segmentation.py:1-5
SCRIPT_IMPORTS = """
import torch
import datetime
from learner.CaeReconstructionLearner import CaeReconstructionLearner
from common.model.Cae3D import Cae3D, Enc3D, Enc3DStep, Dec3D
from common import data, util, metrics
from learner.UnetSegmentationLearner import UnetSegmentationLearner
from common.model.Unet3D import Unet3D
from learner.CaePredictionLearner import CaePredictionLearner
from learner.CaeStepLearner import CaeStepLearner
from common.inference.CaeEncInference import CaeEncInference
"""


def test_reference_script_import_blocks_resolve():
    ns = {}
    exec(SCRIPT_IMPORTS, ns)
    assert ns["Cae3D"].__module__ == "common.model.Cae3D" and ns["util"].__name__ == "common.util"
    assert os.path.dirname(ns["util"].__file__).startswith(os.path.join(ROOT, "stroke-prediction_amd"))
    assert issubclass(ns["CaePredictionLearner"], ns["CaeEncInference"]) and issubclass(ns["CaeStepLearner"], ns["CaeReconstructionLearner"])
    for name in ("get_stroke_shape_training_data", "get_stroke_prediction_training_data", "get_testdata", "ResamplePlaneXY", "HemisphericFlip", "ElasticDeform",
                 "ToTensor", "PadImages", "RandomPatch", "HemisphericFlipFixedToCaseId", "KEY_IMAGES", "DIM_CHANNEL_TORCH3D_5"):
        assert hasattr(ns["data"], name), name
    for name in ("get_args_shape_training", "get_args_unet_training", "get_args_step_training", "get_args_shape_testing",
                 "get_args_shape_prediction_training", "get_args_sdm", "ExpParser", "CAEParser", "UnetParser", "SDMParser"):
        assert hasattr(ns["util"], name), name
    for name in ("BatchDiceLoss", "binary_measures_torch", "binary_measures_numpy"):
        assert hasattr(ns["metrics"], name), name
    from tester.UnetSegmentationTester import UnetSegmentationTester  # noqa: F401
    from tester.CaeReconstructionTester import CaeReconstructionTester  # noqa: F401
    from tester.Tester import Tester  # noqa: F401


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree exists in the build container only")
def test_every_import_of_the_named_reference_scripts_resolves():
    """parse the two scripts' real import statements (text study of the reference; nothing is executed from it)"""
    for script in ("train_shape_reconstruction.py", "train_unet_segmentation.py", "train_shape_prediction.py",
                   "train_interpolationstep_after_reconstruction.py"):
        tree = ast.parse(open(os.path.join("/root/reference", script)).read())
        block = [ast.unparse(n) for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        assert block
        code = "import sys; sys.path.insert(0, %r); import stroke_prediction_amd\n%s\nprint('ok')" % (ROOT, "\n".join(block))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp", timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (script, r.stderr[-1500:])


def test_parsers_take_reference_command_lines(capsys):
    from common import util
    a = util.get_args_shape_training(["--channelscae", "1", "16", "24", "32", "100", "800", "1", "--batchsize", "4", "--epochs", "2",
                                      "--lrsteps", "100", "150", "--outbasepath", "/tmp/x", "--fold", "0", "1", "2"])
    assert a.channelscae == [1, 16, 24, 32, 100, 800, 1] and a.batchsize == 4 and a.lrsteps == [100, 150] and a.fold == [0, 1, 2]
    assert a.globals == 5 and a.normalize == 10 and a.xyoriginal == 256 and a.xyresample == 0.5 and a.zsize == 28
    assert a.steplearning is False and a.inbasepath is None and a.validsetsize == 0.5 and a.seed == 4 and a.padding == [20, 20, 20]
    d = util.get_args_shape_training([])
    assert d.channelscae == [1, 16, 24, 32, 100, 200, 1] and d.epochs == 300 and d.outbasepath == "/tmp/tmp_out" and d.fold == list(range(29))
    u = util.get_args_unet_training(["/tmp/unet.model", "--channels", "2", "16", "32", "64", "32", "16", "32", "2", "--epochs", "3"])
    assert u.unetpath == "/tmp/unet.model" and u.channels == [2, 16, 32, 64, 32, 16, 32, 2] and u.epochs == 3 and u.hemisflipid == 15
    s = util.get_args_step_training(["/tmp/cae.model"])
    assert s.caepath == "/tmp/cae.model"
    p = util.get_args_shape_prediction_training(["/tmp/cae.model", "--initbycae"])
    assert p.initbycae and p.channelsenc == [1, 16, 24, 32, 100, 200, 1]
    t = util.get_args_shape_testing(["--path", "a", "--path", "b", "--fold", "1", "2", "--fold", "3"])
    assert t.path == ["a", "b"] and t.fold == [[1, 2], [3]] and t.normalize == 10
    assert "Namespace(" in capsys.readouterr().out          # ExpParser.parse_args prints the namespace (util.py:54-58)


def test_loader_factories_honour_the_batch_dict_contract():
    from common import data
    tr, va = data.get_stroke_shape_training_data(["m0", "m1"], ["l0", "l1", "l2"], [data.ResamplePlaneXY(0.5), data.ToTensor()],
                                                 [data.ResamplePlaneXY(0.5), data.ToTensor()], list(range(29)), 0.5, seed=4, batchsize=4)
    assert len(tr.sampler.indices) + len(va.sampler.indices) == 29 and len(va.sampler.indices) == 14
    b = next(iter(tr))
    assert set(b) >= {data.KEY_CASE_ID, data.KEY_IMAGES, data.KEY_LABELS, data.KEY_GLOBAL}
    assert tuple(b[data.KEY_IMAGES].shape) == (4, 2, 28, 128, 128)       # B x C x D x H x W (data.py:299-310)
    assert tuple(b[data.KEY_LABELS].shape) == (4, 3, 28, 128, 128) and tuple(b[data.KEY_GLOBAL].shape) == (4, 5, 1, 1, 1)
    lab = b[data.KEY_LABELS]
    assert set(np.unique(lab.cpu().numpy())) <= {0.0, 1.0}
    assert bool((lab[:, 0] <= lab[:, 2]).all()) and bool((lab[:, 2] <= lab[:, 1]).all())     # core in lesion in penumbra
    one, none = data.get_stroke_shape_training_data([], ["l0", "l1", "l2"], [data.ToTensor()], [data.ToTensor()], [0, 1, 2, 3], 0.5,
                                                    batchsize=2, split=False)
    assert none is None and len(one.sampler.indices) == 4
    te = data.get_testdata(["m0", "m1"], ["l0", "l1"], [5, 6], transform=[data.ToTensor()])
    assert te.batch_size == 1 and len(te) == 2
    # same seed, same split (split_data_loader3D shuffles the fold with RandomState(seed), data.py:127-131)
    tr2, _ = data.get_stroke_shape_training_data(["m0"], ["l0"], [data.ToTensor()], [data.ToTensor()], list(range(29)), 0.5, seed=4)
    assert sorted(tr2.sampler.indices) == sorted(tr.sampler.indices)


def test_enc3dstep_constructs_and_learns_a_step_head():
    from common.model.Cae3D import Enc3DStep, Enc3DCtp
    import common.dto.CaeDto as CaeDtoUtil
    enc = Enc3DStep(128, 28, [1, 16, 24, 32, 100, 200, 1], 5, 1.0)
    keys = set(enc.state_dict())
    assert {"reduce.0.weight", "reduce.2.weight", "step.weight", "step.bias", "encoder.1.weight"} <= keys
    dto = CaeDtoUtil.init_dto(torch.rand(2, 5, 1, 1, 1), None, None, None, None, None, None, None, None)
    step = enc._get_step(dto)            # no time to treatment given: predicted from the globals (Cae3D.py:138-142)
    assert tuple(step.shape) == (2, 1, 1, 1, 1) and bool(((step > 0.55) & (step < 0.7)).all())      # sigmoid(~0.5)
    dto.given_variables.time_to_treatment = torch.full((2, 1, 1, 1, 1), 0.25)
    assert torch.equal(enc._get_step(dto), dto.given_variables.time_to_treatment)
    with pytest.raises(NotImplementedError):
        Enc3DCtp(128, 28, [1, 16, 24, 32, 100, 200, 1], 5, 1.0)


def test_reference_pickled_models_load_as_dropin_classes():
    """``torch.load`` of whole-module files written by the reference classes (tests/golden/make_golden.py:gen_checkpoints)"""
    from common.model.Unet3D import Unet3D
    from common.model.Cae3D import Cae3D, Enc3D, Dec3D
    fx = np.load(os.path.join(GOLD, "ref_checkpoints.npz"))
    m = torch.load(os.path.join(GOLD, "ref_unet.model"), weights_only=False)
    assert type(m) is Unet3D and m.channels == [int(c) for c in fx["channels"]] and m.compute_dtype == "bf16"
    fresh = Unet3D(m.channels)
    assert list(m.state_dict()) == list(fresh.state_dict())
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(v.shape) for v in fresh.state_dict().values()]
    assert sum(p.numel() for p in m.parameters()) == sum(p.numel() for p in fresh.parameters())
    m.freeze(True)
    assert not any(p.requires_grad for p in m.parameters())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        import common.dto.UnetDto as U
        m(U.init_dto(torch.zeros(1, 2, 44, 44, 44)))
    c = torch.load(os.path.join(GOLD, "ref_cae.model"), weights_only=False)
    assert type(c) is Cae3D and type(c.enc) is Enc3D and type(c.dec) is Dec3D
    assert c.enc.channels == [int(v) for v in fx["cae_channels"]] and c.dec.channels == c.enc.channels and c.enc.alpha == 1.0
    fresh = Cae3D(Enc3D(64, 28, c.enc.channels, 5, 1.0), Dec3D(64, 28, c.enc.channels, 5, 1.0))
    assert list(c.state_dict()) == list(fresh.state_dict())
    # and back: a model saved here re-loads (engines / flat buffers are not pickled)
    import io
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_reference_optimizer_state_and_jsonpickle_history_load(tmp_path):
    from learner.Learner import _decode_metrics, _encode_metrics
    import common.dto.MetricMeasuresDto as MM
    from stroke_prediction_amd.optim import FusedAdam
    # .optim written by torch.optim.Adam under the reference (Learner.py:108) into FusedAdam
    m = torch.load(os.path.join(GOLD, "ref_unet.model"), weights_only=False)
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    opt.load_state_dict(torch.load(os.path.join(GOLD, "ref_unet.optim"), weights_only=False))
    fx = np.load(os.path.join(GOLD, "ref_checkpoints.npz"))
    second = list(m.parameters())[1]
    assert int(opt.state[second]["step"]) == int(fx["optim_step"])
    np.testing.assert_array_equal(opt.state[second]["exp_avg"].reshape(-1)[:8].numpy(), fx["exp_avg_head"])
    # .json in jsonpickle 0.9.6's layout (what Learner.py:110 writes): py/object paths, Infinity for numpy.Inf
    text = ('{"training": [{"py/object": "common.dto.MetricMeasuresDto.MetricMeasuresDto", "loss": 0.5, '
            '"core": {"py/object": "common.dto.MetricMeasuresDto.BinaryMeasuresDto", "dc": 0.25, "hd": Infinity, "assd": Infinity, '
            '"precision": null, "sensitivity": null, "specificity": null}, '
            '"penu": {"py/object": "common.dto.MetricMeasuresDto.BinaryMeasuresDto", "dc": 0.5, "hd": 3.0, "assd": 1.5, '
            '"precision": null, "sensitivity": null, "specificity": null}, '
            '"lesion": {"py/object": "common.dto.MetricMeasuresDto.BinaryMeasuresDto", "dc": null, "hd": null, "assd": null, '
            '"precision": null, "sensitivity": null, "specificity": null}}], '
            '"validate": [{"py/object": "common.dto.MetricMeasuresDto.MetricMeasuresDto", "loss": 0.75, '
            '"core": {"py/object": "common.dto.MetricMeasuresDto.BinaryMeasuresDto", "dc": 0.125, "hd": 2.0, "assd": 1.0, '
            '"precision": null, "sensitivity": null, "specificity": null}, '
            '"penu": {"py/object": "common.dto.MetricMeasuresDto.BinaryMeasuresDto", "dc": 0.5, "hd": 3.0, "assd": 1.5, '
            '"precision": null, "sensitivity": null, "specificity": null}, '
            '"lesion": {"py/object": "common.dto.MetricMeasuresDto.BinaryMeasuresDto", "dc": null, "hd": null, "assd": null, '
            '"precision": null, "sensitivity": null, "specificity": null}}]}')
    hist = _decode_metrics(text)
    t, v = hist["training"][0], hist["validate"][0]
    assert isinstance(t, MM.MetricMeasuresDto) and isinstance(t.core, MM.BinaryMeasuresDto)
    assert t.loss == 0.5 and t.core.dc == 0.25 and t.core.hd == float("inf") and t.penu.assd == 1.5 and t.lesion.dc is None
    assert v.loss == 0.75 and v.core.dc == 0.125 and v.penu.hd == 3.0
    again = json.loads(_encode_metrics(hist))
    assert again["training"][0]["py/object"] == "common.dto.MetricMeasuresDto.MetricMeasuresDto"
    assert again["training"][0]["core"]["py/object"] == "common.dto.MetricMeasuresDto.BinaryMeasuresDto"
    assert again["training"][0]["core"]["hd"] == float("inf")
    back = _decode_metrics(_encode_metrics(hist))
    assert back["validate"][0].penu.hd == 3.0 and back["training"][0].core.hd == float("inf")
