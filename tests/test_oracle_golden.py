"""Pin the CPU oracle (oracle/) against fixtures produced by the REAL reference
modules (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import nets, weights as W

UNET_CH = [2, 16, 32, 64, 32, 16, 32, 2]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _leafify(sd):
    for k in nets.trainable(sd):
        sd[k] = sd[k].clone().requires_grad_(True)
    return sd


@pytest.mark.parametrize("fname", ["unet_44.npz", "unet_48.npz", "unet_44x48x52.npz", "unet4_92.npz"])
def test_unet_train_steps_match_reference(golden_dir, fname):
    """three-scale ``Unet3D`` fixtures and the four-scale ``LargeUnet3D`` one (the reference class, constructed as
    tests/golden/make_golden.py:reference_large_unet explains)"""
    fx = _load(golden_dir, fname)
    seed, size = int(fx["seed"]), tuple(int(s) for s in np.atleast_1d(fx["size"]))
    size = size * 3 if len(size) == 1 else size
    ch = [int(c) for c in fx["channels"]]
    sd = _leafify(W.make_state_dict(W.unet_spec(ch), seed))
    x, y = W.unet_inputs(2, size, seed, scales=(len(ch) - 2) // 2)
    names = nets.trainable(sd)
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    for step in range(3):
        seg = nets.unet_forward(sd, x, training=True)
        loss = nets.unet_loss(seg, y)
        grads = torch.autograd.grad(loss, [sd[k] for k in names])
        assert abs(loss.item() - float(fx["loss/%d" % step])) < 2e-6
        if step == 0:
            np.testing.assert_allclose(seg.detach().numpy(), fx["seg"], rtol=1e-5, atol=1e-6)
            for k, g in zip(names, grads):
                gn = float(fx["gnorm/" + k])
                assert abs(g.double().norm().item() - gn) <= 2e-4 * gn + 1e-9, k
                np.testing.assert_allclose(g.reshape(-1)[:8].numpy(), fx["ghead/" + k], rtol=2e-3, atol=1e-7 + 1e-4 * gn)
        with torch.no_grad():
            nets.adam_step([sd[k] for k in names], grads, m, v, step + 1, lr=1e-3, betas=(0.99, 0.999),
                           weight_decay=1e-5)
        if step in (0, 2):
            for k in sd:
                if k.endswith("running_mean") or k.endswith("running_var"):
                    np.testing.assert_allclose(sd[k].numpy(), fx["buf%d/%s" % (step + 1, k)], rtol=1e-5, atol=1e-6)
    for k in names:
        np.testing.assert_allclose(sd[k].detach().reshape(-1)[:8].numpy(), fx["phead3/" + k], rtol=1e-4, atol=2e-5)
    with torch.no_grad():
        seg = nets.unet_forward(sd, x, training=False)
    np.testing.assert_allclose(seg.numpy(), fx["seg_eval3"], rtol=1e-3, atol=1e-4)


def test_unet_eval128_matches_reference(golden_dir):
    fx = _load(golden_dir, "unet_eval128.npz")
    sd = W.make_state_dict(W.unet_spec(UNET_CH), int(fx["seed"]))
    x, _ = W.unet_inputs(1, 128, int(fx["seed"]))
    with torch.no_grad():
        seg = nets.unet_forward(sd, x, training=False)
    assert tuple(seg.shape) == tuple(fx["shape"])
    np.testing.assert_allclose(seg[:, :, 42:46, 42:46, 42:46].numpy(), fx["crop"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(seg.double().mean(dim=(0, 2, 3, 4)).numpy(), fx["mean"], rtol=1e-6)


def test_unet_train128_matches_reference(golden_dir):
    """the headline size with signal: train-mode forward (batch statistics), classify gain as recorded in the fixture"""
    fx = _load(golden_dir, "unet_train128.npz")
    sd = W.make_state_dict(W.unet_spec(UNET_CH), int(fx["seed"]))
    for k, gain in zip(fx["head_gain_keys"], fx["head_gain"]):
        sd[str(k)] = sd[str(k)] * float(gain)
    x, _ = W.unet_inputs(1, 128, int(fx["seed"]))
    with torch.no_grad():
        seg = nets.unet_forward(sd, x, training=True)
    assert tuple(seg.shape) == tuple(fx["shape"])
    np.testing.assert_allclose(seg[:, :, 40:48, 40:48, 40:48].numpy(), fx["crop"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(seg[:, :, :4, :4, -4:].numpy(), fx["crop_corner"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(seg.double().mean(dim=(0, 2, 3, 4)).numpy(), fx["mean"], rtol=1e-5)
    np.testing.assert_allclose(seg.double().std(dim=(0, 2, 3, 4)).numpy(), fx["std"], rtol=1e-5)
    assert float(min(fx["std"])) > 0.05          # the fixture carries signal


def grad_sample_index(numel, n=64):
    """tests/golden/make_golden.py:grad_sample_index (the fixture's sample positions of a gradient tensor)"""
    return np.unique(np.linspace(0, numel - 1, num=min(n, numel)).round().astype(np.int64))


def test_unet_trainstep128_matches_reference(golden_dir):
    """forward + backward + Adam at the HEADLINE size (2 x 2 x 128^3; VERDICT r4 "next" 3): the oracle's loss, every parameter
    gradient (norm, head, 64-element sample), the BatchNorm buffers and the parameter norms after the step against the reference's"""
    fx = _load(golden_dir, "unet_trainstep128.npz")
    seed, B = int(fx["seed"]), int(fx["batch"])
    sd = W.make_state_dict(W.unet_spec(UNET_CH), seed)
    for k, gain in zip(fx["head_gain_keys"], fx["head_gain"]):
        sd[str(k)] = sd[str(k)] * float(gain)
    sd = _leafify(sd)
    x, y = W.unet_inputs(B, 128, seed)
    names = nets.trainable(sd)
    seg = nets.unet_forward(sd, x, training=True)
    loss = nets.unet_loss(seg, y)
    grads = torch.autograd.grad(loss, [sd[k] for k in names])
    assert abs(loss.item() - float(fx["loss"])) < 2e-6
    np.testing.assert_allclose(seg.detach()[:, :, 40:48, 40:48, 40:48].numpy(), fx["crop"], rtol=1e-4, atol=2e-6)
    for k, g in zip(names, grads):
        gn = float(fx["gnorm/" + k])
        assert abs(g.double().norm().item() - gn) <= 1e-3 * gn + 1e-12, k
        gs = g.reshape(-1)[torch.from_numpy(grad_sample_index(g.numel()))].numpy()
        np.testing.assert_allclose(gs, fx["gsample/" + k], rtol=5e-3, atol=1e-3 * gn / np.sqrt(g.numel()) + 1e-12, err_msg=k)
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    with torch.no_grad():
        nets.adam_step([sd[k] for k in names], grads, m, v, 1, lr=1e-3, betas=(0.99, 0.999), weight_decay=1e-5)
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            np.testing.assert_allclose(sd[k].numpy(), fx["buf1/" + k], rtol=1e-5, atol=1e-6)
    for k in names:
        assert abs(sd[k].detach().double().norm().item() - float(fx["pnorm1/" + k])) <= 1e-5 * float(fx["pnorm1/" + k]) + 1e-7, k


@pytest.mark.parametrize("fname", ["cae_200.npz", "cae_800.npz"])
def test_cae_step_matches_reference(golden_dir, fname):
    fx = _load(golden_dir, fname)
    ch, seed = [int(c) for c in fx["channels"]], int(fx["seed"])
    sd = _leafify(W.make_state_dict(W.cae_spec(ch), seed))
    labels, clinical = W.cae_inputs(2, int(fx["d"]), int(fx["hw"]), seed)
    step = nets.time_to_treatment(clinical)
    np.testing.assert_allclose(step.numpy(), fx["ttt"], rtol=1e-6)
    core, penu, lesion = labels[:, 0:1], labels[:, 1:2], labels[:, 2:3]
    lat, rec = nets.cae_forward(sd, core, penu, lesion, step, alpha=1.0, training=True)
    for k in ("core", "penu", "lesion", "interpolation"):
        np.testing.assert_allclose(lat[k].detach().reshape(2, -1)[:, :64].numpy(), fx["lat_head/" + k], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rec[k].detach()[:, 0, int(fx["d"]) // 2, 60:68, 60:68].numpy(), fx["rec_crop/" + k],
                                   rtol=1e-4, atol=1e-5)
        d = fx["rec_digest/" + k]
        assert abs(rec[k].double().sum().item() - d[0]) <= 1e-5 * d[1]
    for ep in (0, 30, 60):
        assert abs(nets.cae_loss(lat, rec, core, penu, lesion, ep).item() - float(fx["loss_epoch/%d" % ep])) < 2e-6
    names = nets.trainable(sd)
    grads = torch.autograd.grad(nets.cae_loss(lat, rec, core, penu, lesion, 30), [sd[k] for k in names])
    for k, g in zip(names, grads):
        gn = float(fx["gnorm/" + k])
        assert abs(g.double().norm().item() - gn) <= 1e-3 * gn + 1e-9, k
    # BN bookkeeping: encoder BNs see 3 calls, decoder BNs 4 per step (Cae3D.py:105-107,230-233)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(fx["nbt/" + k]) == (3 if k.startswith("enc.") else 4)
    assert abs(nets.cae_beta1(0) - float(fx["betas_epoch0"][0])) < 1e-12
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    with torch.no_grad():
        nets.adam_step([sd[k] for k in names], grads, m, v, 1, lr=1e-3, betas=(nets.cae_beta1(0), 0.999), weight_decay=1e-5)
    for k in names[:8]:
        np.testing.assert_allclose(sd[k].detach().reshape(-1)[:8].numpy(), fx["phead1/" + k], rtol=1e-4, atol=2e-5)
    for k in sd:
        if ("buf1/" + k) in fx.files:
            np.testing.assert_allclose(sd[k].numpy(), fx["buf1/" + k], rtol=1e-5, atol=1e-6)


def test_adam_restatement_matches_torch_optim():
    torch.manual_seed(0)
    p0 = [torch.randn(7, 5), torch.randn(11)]
    ref = [p.clone().requires_grad_(True) for p in p0]
    opt = torch.optim.Adam(ref, lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    mine = [p.clone() for p in p0]
    m = [torch.zeros_like(p) for p in p0]
    v = [torch.zeros_like(p) for p in p0]
    for step in range(1, 6):
        grads = [torch.randn_like(p) for p in p0]
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        opt.step()
        nets.adam_step(mine, grads, m, v, step, lr=1e-3, betas=(0.99, 0.999), weight_decay=1e-5)
    for a, b in zip(ref, mine):
        np.testing.assert_allclose(a.detach().numpy(), b.numpy(), rtol=1e-6, atol=1e-7)


def phase2_inputs(seed, d=28, hw=128):
    """the synthetic batch of tests/golden/make_golden.py:phase2_inputs"""
    labels, clinical = W.cae_inputs(2, d, hw, seed)
    seg, _ = W.cae_inputs(2, d, hw, seed + 7)
    return (0.05 + 0.9 * seg[:, 0:2]).contiguous(), labels, clinical


def test_cae_phase2_learners_match_reference(golden_dir):
    """SURVEY 8(f) N4 second half: CaePredictionLearner (new encoder against the frozen CAE) and CaeStepLearner (learned step)
    as restated in oracle/nets.py against what the reference's own classes computed"""
    fx = _load(golden_dir, "cae_phase2_200.npz")
    assert int(fx["ref_inference_step_raises"]) == 1      # the reference's CaeEncInference.inference_step asserts as written (dto.mode / dto.flag)
    ch, seed, d, hw = [int(c) for c in fx["channels"]], int(fx["seed"]), int(fx["d"]), int(fx["hw"])
    sd_cae = W.make_state_dict(W.cae_spec(ch), seed)
    sd_enc = _leafify(W.make_state_dict(W.enc_spec(ch), seed + 1))
    images, labels, clinical = phase2_inputs(seed, d, hw)
    step = nets.time_to_treatment(clinical)
    np.testing.assert_allclose(step.numpy(), fx["ttt"], rtol=1e-6)
    core, penu, lesion = labels[:, 0:1], labels[:, 1:2], labels[:, 2:3]
    lat_in, rec_in, lat_gt, rec_gt = nets.cae_prediction_forward(sd_cae, sd_enc, images[:, 0:1], images[:, 1:2], core, penu, lesion, step)
    for k in ("core", "penu", "interpolation"):
        np.testing.assert_allclose(lat_in[k].detach().reshape(2, -1)[:, :64].numpy(), fx["lat_in_head/" + k], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rec_in[k].detach()[:, 0, d // 2, 60:68, 60:68].numpy(), fx["rec_in_crop/" + k], rtol=1e-4, atol=1e-5)
    loss = nets.cae_prediction_loss(lat_in, rec_in, lat_gt, lesion)
    assert abs(loss.item() - float(fx["loss"])) < 2e-6
    names = nets.trainable(sd_enc)
    grads = torch.autograd.grad(loss, [sd_enc[k] for k in names])
    for k, g in zip(names, grads):
        gn = float(fx["gnorm/" + k])
        assert abs(float(g.double().norm()) - gn) <= 2e-3 * gn + 1e-9, k
    # ---- CaeStepLearner
    sd2 = dict(W.make_state_dict(W.cae_spec(ch), seed))
    for k in fx.files:
        if k.startswith("step_param/"):
            sd2["enc." + k[len("step_param/"):]] = torch.from_numpy(fx[k]).clone().requires_grad_(True)
    stepv = nets.enc_step(sd2, clinical.float(), 1.0)
    np.testing.assert_allclose(stepv.detach().numpy(), fx["step_value"], rtol=1e-5, atol=1e-7)
    _, rec = nets.cae_forward(sd2, core, penu, lesion, stepv, alpha=1.0, training=True)
    loss2 = nets.cae_step_loss(rec, lesion)
    assert abs(loss2.item() - float(fx["step_loss"])) < 2e-6
    sk = [k for k in sd2 if k.startswith("enc.reduce.") or k.startswith("enc.step.")]
    g2 = torch.autograd.grad(loss2, [sd2[k] for k in sk])
    for k, g in zip(sk, g2):
        ref = fx["step_grad/" + k[len("enc."):]]
        np.testing.assert_allclose(g.numpy(), ref, rtol=2e-3, atol=2e-3 * float(np.abs(ref).max()) + 1e-12)
