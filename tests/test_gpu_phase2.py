"""SURVEY 8(f) N4, second half -- the phase-2 learners on the HIP path: ``CaePredictionLearner`` (a new ``Enc3D`` trained on the
U-Net segmentations against the FROZEN shape CAE: the gradient crosses the frozen decoder as data gradients only) and
``CaeStepLearner`` (``Enc3DStep``: only the 1x1x1 step layers train, through the frozen decoder and the latent interpolation),
against the fixtures recorded from the reference's own classes (tests/golden/make_golden.py:gen_cae_phase2)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Cae3D import Cae3D, Enc3D, Dec3D, Enc3DStep
from stroke_prediction_amd.common.metrics import BatchDiceLoss
from stroke_prediction_amd.learner.CaePredictionLearner import CaePredictionLearner
from stroke_prediction_amd.learner.CaeStepLearner import CaeStepLearner
import stroke_prediction_amd.common.dto.CaeDto as CaeDtoUtil

DEV = "cuda:0"


class _Loader(list):
    batch_size = 2


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def phase2_inputs(seed, d=28, hw=128):
    labels, clinical = W.cae_inputs(2, d, hw, seed)
    seg, _ = W.cae_inputs(2, d, hw, seed + 7)
    return (0.05 + 0.9 * seg[:, 0:2]).contiguous(), labels, clinical


def build(ch, seed, dtype, d, hw):
    cae = Cae3D(Enc3D(hw, d, ch, 5, 1.0, dtype=dtype), Dec3D(hw, d, ch, 5, 1.0, dtype=dtype))
    cae.load_state_dict(W.make_state_dict(W.cae_spec(ch), seed))
    enc = Enc3D(hw, d, ch, 5, 1.0, dtype=dtype)
    enc.load_state_dict(W.make_state_dict(W.enc_spec(ch), seed + 1))
    return cae.to(DEV), enc.to(DEV)


def prediction_learner(cae, enc):
    opt = torch.optim.Adam([p for p in enc.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    return CaePredictionLearner(_Loader(), None, cae, enc, opt, None, n_epochs=1, path_previous_base=None,
                                path_outputs_base="/tmp/_cae2_test", criterion=BatchDiceLoss([1.0])), opt


@pytest.mark.parametrize("dtype,tol_out,tol_loss,tol_grad", [("f32", 2e-4, 2e-5, 5e-3), ("bf16", 5e-2, 3e-3, 0.12)])
def test_prediction_learner_matches_reference_fixture(golden_dir, dtype, tol_out, tol_loss, tol_grad):
    fx = np.load(os.path.join(golden_dir, "cae_phase2_200.npz"))
    ch, seed, d, hw = [int(c) for c in fx["channels"]], int(fx["seed"]), int(fx["d"]), int(fx["hw"])
    images, labels, clinical = phase2_inputs(seed, d, hw)
    cae, enc = build(ch, seed, dtype, d, hw)
    learner, opt = prediction_learner(cae, enc)
    assert not any(p.requires_grad for p in cae.parameters())           # CaePredictionLearner.py:27
    cae.train()
    dto = learner.inference_step({"case_id": [0, 1], "images": images, "labels": labels, "clinical": clinical})
    assert dto.flag == CaeDtoUtil.FLAG_GTRUTH
    np.testing.assert_allclose(dto.given_variables.time_to_treatment.cpu().numpy(), fx["ttt"], rtol=1e-6)
    for k in ("core", "penu", "interpolation"):
        lat = getattr(dto.latents.inputs, k).detach().cpu()
        rec = getattr(dto.reconstructions.inputs, k).detach().cpu()
        ref = fx["lat_in_head/" + k]
        assert float(np.abs(lat.reshape(2, -1)[:, :64].numpy() - ref).max()) <= 10 * tol_out * float(np.abs(ref).max()) + 2e-4, k
        np.testing.assert_allclose(rec[:, 0, d // 2, 60:68, 60:68].numpy(), fx["rec_in_crop/" + k], rtol=0, atol=tol_out)
        assert getattr(dto.latents.gtruth, k) is not None and getattr(dto.reconstructions.gtruth, k) is not None
    assert dto.latents.gtruth.lesion is not None and not dto.latents.gtruth.core.requires_grad      # the frozen CAE's own calls: no autograd node
    loss = learner.loss_step(dto, 0)
    assert abs(float(loss) - float(fx["loss"])) < tol_loss
    opt.zero_grad()
    loss.backward()
    bad = []
    for n, p in enc.named_parameters():
        gn = float(fx["gnorm/" + n])
        tol = tol_grad if p.numel() > 16 else max(2e-2, 4 * tol_grad)
        # bf16: the gradient of the FIRST BatchNorm's single gamma is zero in exact arithmetic (the next BatchNorm removes any
        # scale of the one input channel; 1e-5 in the reference comes from its eps terms) -- storage noise: an absolute floor
        floor = 3e-4 if (dtype == "bf16" and p.numel() <= 16) else 1e-9      # (1.4e-4 measured since the decoder's forward folds its BatchNorms per group)
        if abs(float(p.grad.double().norm()) - gn) > tol * gn + floor:
            bad.append((n, float(p.grad.double().norm()), gn))
    assert not bad, bad
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in cae.parameters())     # nothing reached the frozen CAE's gradients
    for n, b in cae.named_buffers():
        if n.endswith("num_batches_tracked"):
            assert int(b) == int(fx["nbt/" + n]), n          # decoder: 3 (inputs) + 4 (gtruth) calls, CAE encoder: 3
        elif n.startswith("dec.decoder.0."):
            np.testing.assert_allclose(b.cpu().numpy(), fx["buf1/" + n], rtol=5e-3 if dtype == "f32" else 5e-2, atol=1e-3)
    for n, b in enc.named_buffers():
        if n.endswith("num_batches_tracked"):
            assert int(b) == int(fx["nbt/newenc." + n]) == 2
    opt.step()


def test_frozen_decoder_gradient_equals_the_unfrozen_one():
    """the data-gradient-only backward of a frozen stack (no weight-gradient kernels) hands the trainable encoder the same
    gradient as the full backward of the same, un-frozen stack (f32 mode: the BatchNorm-backward sums come from another
    reduction order only)"""
    ch, seed, d, hw = [1, 16, 24, 32, 100, 200, 1], 31, 28, 64
    images, labels, clinical = phase2_inputs(seed, d, hw)
    batch = {"case_id": [0, 1], "images": images, "labels": labels, "clinical": clinical}
    grads = {}
    for frozen in (True, False):
        cae, enc = build(ch, seed, "f32", d, hw)
        learner, opt = prediction_learner(cae, enc)
        cae.freeze(frozen)
        cae.train()
        dto = learner.inference_step(batch)
        loss = learner.loss_step(dto, 0)
        opt.zero_grad()
        loss.backward()
        grads[frozen] = {n: p.grad.detach().clone() for n, p in enc.named_parameters()}
        if not frozen:
            assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in cae.dec.parameters())
    for n in grads[True]:
        assert rel_l2(grads[True][n], grads[False][n]) < 2e-4, n


@pytest.mark.parametrize("dtype,tol_out,tol_loss,tol_grad", [("f32", 2e-4, 2e-5, 1e-2), ("bf16", 5e-2, 3e-3, 0.15)])
def test_step_learner_matches_reference_fixture(golden_dir, dtype, tol_out, tol_loss, tol_grad):
    fx = np.load(os.path.join(golden_dir, "cae_phase2_200.npz"))
    ch, seed, d, hw = [int(c) for c in fx["channels"]], int(fx["seed"]), int(fx["d"]), int(fx["hw"])
    images, labels, clinical = phase2_inputs(seed, d, hw)
    cae, _ = build(ch, seed, dtype, d, hw)
    cae.freeze(True)
    senc = Enc3DStep(hw, d, ch, 5, 1.0, dtype=dtype)
    senc.encoder = cae.enc.encoder                      # train_interpolationstep_after_reconstruction.py:25
    sd = senc.state_dict()
    for k in fx.files:
        if k.startswith("step_param/"):
            sd[k[len("step_param/"):]] = torch.from_numpy(fx[k])
    senc.load_state_dict(sd)
    cae2 = Cae3D(senc, cae.dec).to(DEV).train()
    params = [p for p in cae2.parameters() if p.requires_grad]
    assert sum(p.numel() for p in params) == sum(int(np.prod(fx[k].shape)) for k in fx.files if k.startswith("step_param/"))
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    learner = CaeStepLearner(_Loader(), None, cae2, opt, None, n_epochs=1, path_previous_base=None,
                             path_outputs_base="/tmp/_cae1step_test", criterion=BatchDiceLoss([1.0]), verbose=False)
    dto = learner.inference_step({"case_id": [0, 1], "images": images, "labels": labels, "clinical": clinical})
    assert dto.given_variables.time_to_treatment is None
    np.testing.assert_allclose(senc._get_step(dto).detach().cpu().numpy(), fx["step_value"], rtol=1e-5, atol=1e-6)
    for k in ("penu", "interpolation"):
        rec = getattr(dto.reconstructions.gtruth, k).detach().cpu()
        np.testing.assert_allclose(rec[:, 0, d // 2, 60:68, 60:68].numpy(), fx["step_rec_crop/" + k], rtol=0, atol=tol_out)
    loss = learner.loss_step(dto, 0)
    assert abs(float(loss) - float(fx["step_loss"])) < tol_loss
    opt.zero_grad()
    loss.backward()
    for n, p in senc.named_parameters():
        if p.requires_grad:
            ref = torch.from_numpy(fx["step_grad/" + n])
            assert rel_l2(p.grad.cpu(), ref) < tol_grad or float((p.grad.cpu() - ref).abs().max()) < 1e-7, (n, rel_l2(p.grad.cpu(), ref))
    opt.step()


def test_phase2_training_scripts_run(tmp_path):
    """the two phase-2 scripts end to end on the synthetic data set: phase 1 for one epoch writes the CAE, shape prediction and
    step learning load it (reference train_shape_prediction.py / train_interpolationstep_after_reconstruction.py)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pk = os.path.join(root, "stroke-prediction_amd")
    base = str(tmp_path / "run")
    common = ["--epochs", "1", "--batchsize", "2", "--fold", "0", "1", "2", "3", "--validsetsize", "0.5", "--outbasepath", base]
    chcae = ["--channelscae", "1", "16", "24", "32", "100", "200", "1"]
    env = dict(os.environ, MPLBACKEND="Agg")
    r = subprocess.run([sys.executable, os.path.join(pk, "train_shape_reconstruction.py")] + common + chcae, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    caepath = base + "_cae1_final.model"
    assert os.path.exists(caepath), os.listdir(str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(pk, "train_shape_prediction.py"), caepath] + common +
                       ["--channelsenc", "1", "16", "24", "32", "100", "200", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(base + "_cae2_final.model") and os.path.exists(base + "_cae2_enc_final.model")
    r = subprocess.run([sys.executable, os.path.join(pk, "train_interpolationstep_after_reconstruction.py"), caepath] + common + chcae,
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(base + "_cae1step_final.model")
