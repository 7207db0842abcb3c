"""Drop-in behaviour on the GPU: the two training scripts run end to end on synthetic cases,
``.model`` files written by the REFERENCE's classes load through ``Learner.load_model`` / the concrete testers and
reproduce the reference's outputs, and the reference's own optimiser choice (torch.optim.Adam) trains the HIP models."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stroke-prediction_amd")
GOLD = os.path.join(ROOT, "tests", "golden")
DEV = "cuda:0"

from oracle import weights as W
import stroke_prediction_amd  # noqa: F401


def _run(script, args, timeout=900):
    env = dict(os.environ, SP_SYNTHETIC_DATA="1", MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, os.path.join(PKG, script)] + args, capture_output=True, text=True, env=env, timeout=timeout,
                       cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("extra", [[], ["--fusedadam", "--graph"]])
def test_train_shape_reconstruction_script_two_epochs(tmp_path, extra):
    """train_shape_reconstruction.py:8-79 on synthetic cases: 2 epochs, checkpoints in the reference's naming scheme.
    Default flags = the reference's configuration (torch.optim.Adam, eager); second run: FusedAdam + hipGraph."""
    base = str(tmp_path / "cae")
    out = _run("train_shape_reconstruction.py", ["--epochs", "2", "--batchsize", "2", "--fold", "0", "1", "2", "3", "4", "5", "6", "7",
                                                 "--channelscae", "1", "16", "24", "32", "100", "200", "1", "--outbasepath", base] + extra)
    assert "Epoch 2/2 training loss" in out and "Epoch 2/2 validate loss" in out
    for suffix in ("_cae1.model", "_cae1.optim", "_cae1.json", "_cae1_final.model"):
        assert os.path.exists(base + suffix), suffix
    import re
    losses = [float(v) for v in re.findall(r"training loss: ([0-9.eE+-]+)", out)]
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0] + 0.05
    # the saved best model re-loads under the reference's import path and evaluates through the concrete tester
    from common import data
    from tester.CaeReconstructionTester import CaeReconstructionTester
    os.environ["SP_SYNTHETIC_DATA"] = "1"
    loader = data.get_testdata([], ['l0', 'l1', 'l2'], [0, 1], transform=[data.ResamplePlaneXY(0.5), data.ToTensor()])
    tester = CaeReconstructionTester(loader, base + "_cae1.model", path_outputs_base=str(tmp_path / "eval"))
    tester._model.to(DEV)
    n = 0
    for batch in loader:
        m, dto = tester.infer_batch(batch)
        assert 0.0 <= m.lesion.dc <= 1.0 and 0.0 <= m.core.dc <= 1.0
        assert tuple(dto.reconstructions.gtruth.interpolation.shape) == (1, 1, 28, 128, 128)
        tester.print_inference(batch, m, dto)
        n += 1
    assert n == 2 and any(f.endswith("_pred.npy") for f in os.listdir(str(tmp_path)))


def test_train_unet_segmentation_script_two_epochs(tmp_path):
    """the intended behaviour of train_unet_segmentation.py: 104 x 104 x 68 patches -> 64 x 64 x 28 outputs"""
    base = str(tmp_path / "unet")
    unetpath = str(tmp_path / "unet.model")
    out = _run("train_unet_segmentation.py", [unetpath, "--epochs", "2", "--batchsize", "3", "--fold"] + [str(i) for i in range(12)] +
               ["--outbasepath", base, "--fusedadam", "--graph"])
    assert "Epoch 2/2 training loss" in out
    for f in (base + "_unet.model", base + "_unet.optim", base + "_unet.json", base + "_unet_final.model", unetpath):
        assert os.path.exists(f), f
    from common import data
    from tester.UnetSegmentationTester import UnetSegmentationTester
    os.environ["SP_SYNTHETIC_DATA"] = "1"
    pad = [20, 20, 20]
    tf = [data.ResamplePlaneXY(0.5), data.PadImages(*pad, pad_value=0), data.ToTensor()]          # test_unet_segmentation.py:17-20
    loader = data.get_testdata(['m0', 'm1'], ['l0', 'l1'], [3], transform=tf)
    tester = UnetSegmentationTester(loader, unetpath, path_outputs_base=str(tmp_path / "seg"), padding=None)
    tester._model.to(DEV)
    for batch in loader:
        m, dto = tester.infer_batch(batch)
        assert tuple(dto.outputs.core.shape) == (1, 1, 28, 128, 128)       # whole padded volume in, whole volume out
        assert 0.0 <= m.core.dc <= 1.0 and 0.0 <= m.penu.dc <= 1.0


def test_reference_pickled_unet_loads_through_learner_and_matches(tmp_path):
    """N3: ``Learner.load_model`` (Learner.py:90-95) on a ``.model`` written by the reference's ``Unet3D``; eval-mode output
    against what the reference produced from the same file (parity mode; bf16 within storage noise)."""
    import shutil
    from stroke_prediction_amd.learner.UnetSegmentationLearner import UnetSegmentationLearner
    from stroke_prediction_amd.optim import FusedAdam
    import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil
    fx = np.load(os.path.join(GOLD, "ref_checkpoints.npz"))
    base = str(tmp_path / "prev")
    shutil.copyfile(os.path.join(GOLD, "ref_unet.model"), base + "_unet.model")
    shutil.copyfile(os.path.join(GOLD, "ref_unet.optim"), base + "_unet.optim")
    with open(base + "_unet.json", "w") as f:
        f.write('{"training": [], "validate": []}')
    from stroke_prediction_amd.common.model.Unet3D import Unet3D
    placeholder = Unet3D([int(c) for c in fx["channels"]]).to(DEV)
    opt = FusedAdam(placeholder.parameters(), lr=1e-3)

    class Loader(list):
        batch_size = 2
    learner = UnetSegmentationLearner(Loader(), None, placeholder, opt, None, 1, None, path_previous_base=base,
                                      path_outputs_base=str(tmp_path / "out"))
    model = learner._model
    assert model is not placeholder and next(model.parameters()).is_cuda
    x, _ = W.unet_inputs(1, 44, int(fx["seed"]))
    for mode, tol in (("f32", 1e-4), ("bf16", 2e-2)):
        model.compute_dtype = mode
        model._engines = {}
        model.eval()
        with torch.no_grad():
            dto = model(UnetDtoUtil.init_dto(x.to(DEV)))
        seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1).cpu().numpy()
        np.testing.assert_allclose(seg, fx["seg"], rtol=0, atol=tol)


def test_reference_pickled_cae_matches_on_gpu():
    import stroke_prediction_amd.common.dto.CaeDto as CaeDtoUtil
    fx = np.load(os.path.join(GOLD, "ref_checkpoints.npz"))
    cae = torch.load(os.path.join(GOLD, "ref_cae.model"), weights_only=False).to(DEV)
    cae.enc.compute_dtype = cae.dec.compute_dtype = "f32"
    cae.eval()
    labels, clinical = W.cae_inputs(1, 28, 64, int(fx["cae_seed"]))
    labels = labels.to(DEV)
    with torch.no_grad():
        dto = CaeDtoUtil.init_dto(clinical.float().to(DEV), torch.tensor([[[[[0.25]]]]], device=DEV), None, None, None, None, None, None, None)
        g = dto.given_variables.gtruth
        g.core, g.penu, g.lesion = labels[:, 0:1], labels[:, 1:2], labels[:, 2:3]
        dto = cae(dto)
    for k in ("core", "penu", "lesion", "interpolation"):
        rec = getattr(dto.reconstructions.gtruth, k).cpu().numpy()[:, :, 10:18, 24:40, 24:40]
        np.testing.assert_allclose(rec, fx["cae_rec/" + k], rtol=0, atol=2e-4)
        lat = getattr(dto.latents.gtruth, k).cpu().numpy()
        np.testing.assert_allclose(lat, fx["cae_lat/" + k], rtol=2e-3, atol=2e-4)
