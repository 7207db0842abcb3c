#!/usr/bin/env python3
"""Generate golden fixtures by running the REAL reference modules on CPU.

Run only in the build container (needs ``/root/reference``; nothing under
``tests/`` reads that path at test time):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

The reference's model/loss/learner classes are imported unmodified.  Modules
that the hot path imports but never uses and that are absent here (nibabel,
torchvision, medpy, jsonpickle) are replaced by empty stand-in modules in
``sys.modules`` before import.  Weights and inputs come from
``oracle/weights.py`` (own generator), so fixtures hold data only: inputs are
regenerated from seeds, outputs are stored.

Versions recorded in each fixture: torch version (trilinear upsample runs with
align_corners=False under torch>=0.4).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("STROKE_REFERENCE", "/root/reference")


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def import_reference():
    for n in ("nibabel", "torchvision", "medpy", "medpy.metric", "medpy.metric.binary", "jsonpickle"):
        if n not in sys.modules:
            _stub(n)
    sys.modules["torchvision"].transforms = _stub("torchvision.transforms")
    sys.modules["medpy"].metric = sys.modules["medpy.metric"]
    sys.modules["medpy.metric"].binary = sys.modules["medpy.metric.binary"]
    sys.path.insert(0, REF)


def digest(t):
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def grads_summary(named):
    out = {}
    for n, p in named:
        g = p.grad.detach().reshape(-1)
        out["gnorm/" + n] = np.float64(g.double().norm().item())
        out["ghead/" + n] = g[:8].numpy().copy()
    return out


def reference_large_unet(channels):
    """The reference's OWN ``LargeUnet3D`` (common/model/Unet3D.py:87-146), unmodified.  Its constructor opens with
    ``super(Unet3D, self).__init__()`` (Unet3D.py:89), which raises for an object that is not a ``Unet3D``; the name
    ``Unet3D`` in that module's namespace is bound to ``LargeUnet3D`` for the duration of the call, so the line reads
    ``super(LargeUnet3D, self).__init__()`` -- evidently what was meant -- and the class's own layer construction and
    ``forward`` run as written.  No reference file is touched."""
    import common.model.Unet3D as ref
    keep = ref.Unet3D
    ref.Unet3D = ref.LargeUnet3D
    try:
        return ref.LargeUnet3D(channels)
    finally:
        ref.Unet3D = keep


def gen_unet(size, seed, fname, ch=(2, 16, 32, 64, 32, 16, 32, 2)):
    from oracle import weights as W
    from common.model.Unet3D import Unet3D
    import common.dto.UnetDto as UnetDtoUtil
    from common.metrics import BatchDiceLoss

    ch = list(ch)
    scales = (len(ch) - 2) // 2
    model = Unet3D(ch) if scales == 3 else reference_large_unet(ch)
    model.load_state_dict(W.make_state_dict(W.unet_spec(ch), seed))
    x, y = W.unet_inputs(2, size, seed, scales=scales)
    crit = BatchDiceLoss([1.0])
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3,
                           weight_decay=1e-5, betas=(0.99, 0.999))   # train_unet_segmentation.py:13-14,32
    fx = {"channels": np.array(ch), "size": np.array(size), "seed": np.array(seed), "batch": np.array(2),
          "torch_version": np.array(torch.__version__)}
    model.train()
    for step in range(3):
        dto = UnetDtoUtil.init_dto(x, y[:, 0:1], y[:, 1:2])      # UnetInference.py:16-25
        dto = model(dto)
        # UnetSegmentationLearner.loss_step :21-28 (the learner ctor itself is broken)
        loss = (crit(dto.outputs.core, dto.given_variables.core) + crit(dto.outputs.penu, dto.given_variables.penu)) / 2
        opt.zero_grad()
        loss.backward()
        if step == 0:
            fx["seg"] = torch.cat((dto.outputs.core, dto.outputs.penu), 1).detach().numpy().copy()
            fx.update(grads_summary(model.named_parameters()))
        fx["loss/%d" % step] = np.float64(loss.item())
        opt.step()
        if step in (0, 2):
            for n, b in model.named_buffers():
                if not n.endswith("num_batches_tracked"):
                    fx["buf%d/%s" % (step + 1, n)] = b.detach().numpy().copy()
    for n, p in model.named_parameters():
        fx["pnorm3/" + n] = np.float64(p.detach().double().norm().item())
        fx["phead3/" + n] = p.detach().reshape(-1)[:8].numpy().copy()
    model.eval()
    with torch.no_grad():
        dto = model(UnetDtoUtil.init_dto(x))
    fx["seg_eval3"] = torch.cat((dto.outputs.core, dto.outputs.penu), 1).numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **fx)
    print("wrote", fname, "loss", [fx["loss/%d" % i] for i in range(3)])


def gen_unet_eval128(seed, fname):
    from oracle import weights as W
    from common.model.Unet3D import Unet3D
    import common.dto.UnetDto as UnetDtoUtil
    ch = [2, 16, 32, 64, 32, 16, 32, 2]
    model = Unet3D(ch)
    model.load_state_dict(W.make_state_dict(W.unet_spec(ch), seed))
    model.freeze(True)
    model.eval()
    x, _ = W.unet_inputs(1, 128, seed)
    with torch.no_grad():
        dto = model(UnetDtoUtil.init_dto(x))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    fx = {"seed": np.array(seed), "shape": np.array(seg.shape), "torch_version": np.array(torch.__version__),
          "mean": seg.double().mean(dim=(0, 2, 3, 4)).numpy(), "std": seg.double().std(dim=(0, 2, 3, 4)).numpy(),
          "crop": seg[:, :, 42:46, 42:46, 42:46].numpy().copy(), "digest": digest(seg)}
    np.savez_compressed(os.path.join(HERE, fname), **fx)
    print("wrote", fname, fx["mean"], fx["std"])


HEAD_GAIN = {"classify.0.weight": 2.0, "classify.2.weight": 6.0}


def gen_unet_train128(seed, fname):
    """the headline spatial size WITH signal: a TRAIN-mode forward (batch statistics) of the reference's
    Unet3D at 1 x 2 x 128^3 -- every BatchNorm renormalises, so the logits spread over several units (the eval-mode fixture
    above, with untouched running statistics, has an output std of 3.6e-4: bf16-level errors hide in it)"""
    from oracle import weights as W
    from common.model.Unet3D import Unet3D
    import common.dto.UnetDto as UnetDtoUtil
    ch = [2, 16, 32, 64, 32, 16, 32, 2]
    model = Unet3D(ch)
    sd = W.make_state_dict(W.unet_spec(ch), seed)
    # the generator's head weights (U(+-1/sqrt(fan_in))) squeeze every logit into +-0.7: the classify convolutions get a gain, so
    # that the probabilities use their range (std ~0.1-0.2) and a wrong bit in any layer shows at the output
    for k, gain in HEAD_GAIN.items():
        sd[k] = sd[k] * gain
    model.load_state_dict(sd)
    model.train()
    x, _ = W.unet_inputs(1, 128, seed)
    with torch.no_grad():
        dto = model(UnetDtoUtil.init_dto(x))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    logit = torch.log(seg.double() / (1 - seg.double()))
    fx = {"seed": np.array(seed), "shape": np.array(seg.shape), "torch_version": np.array(torch.__version__),
          "mean": seg.double().mean(dim=(0, 2, 3, 4)).numpy(), "std": seg.double().std(dim=(0, 2, 3, 4)).numpy(),
          "logit_std": logit.std(dim=(0, 2, 3, 4)).numpy(), "logit_absmax": np.float64(logit.abs().max().item()),
          "crop": seg[:, :, 40:48, 40:48, 40:48].numpy().copy(), "crop_corner": seg[:, :, :4, :4, -4:].numpy().copy(),
          "digest": digest(seg), "head_gain_keys": np.array(sorted(HEAD_GAIN)), "head_gain": np.array([HEAD_GAIN[k] for k in sorted(HEAD_GAIN)])}
    for n, b in model.named_buffers():
        if n.endswith("running_mean") and n.startswith("block5"):
            fx["buf/" + n] = b.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **fx)
    print("wrote", fname, fx["mean"], fx["std"], fx["logit_std"], fx["logit_absmax"])


def grad_sample_index(numel, n=64):
    """the fixed positions of a gradient tensor the 128^3 training-step fixture keeps (the tests index with the same function)"""
    return np.unique(np.linspace(0, numel - 1, num=min(n, numel)).round().astype(np.int64))


def gen_unet_trainstep128(seed, fname, batch=2):
    """forward + backward + one Adam step of the reference at the HEADLINE spatial size (VERDICT r4 "next" 3): Unet3D in train mode on
    batch x 2 x 128^3 (Unet3D.py:56-79), the segmentation loss (UnetSegmentationLearner.py:21-28, metrics.py:16-28), loss.backward(),
    torch.optim.Adam as train_unet_segmentation.py:32 builds it.  Kept: the loss, norm / head / a fixed 64-element sample of every
    parameter gradient, every BatchNorm buffer after the step, parameter norms after the step, a crop of the output.  Same weights
    as gen_unet_train128 (classify gains: outputs with signal)."""
    from oracle import weights as W
    from common.model.Unet3D import Unet3D
    import common.dto.UnetDto as UnetDtoUtil
    from common.metrics import BatchDiceLoss
    ch = [2, 16, 32, 64, 32, 16, 32, 2]
    model = Unet3D(ch)
    sd = W.make_state_dict(W.unet_spec(ch), seed)
    for k, gain in HEAD_GAIN.items():
        sd[k] = sd[k] * gain
    model.load_state_dict(sd)
    model.train()
    x, y = W.unet_inputs(batch, 128, seed)
    crit = BatchDiceLoss([1.0])
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    dto = model(UnetDtoUtil.init_dto(x, y[:, 0:1], y[:, 1:2]))
    loss = (crit(dto.outputs.core, dto.given_variables.core) + crit(dto.outputs.penu, dto.given_variables.penu)) / 2
    opt.zero_grad()
    loss.backward()
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1).detach()
    fx = {"seed": np.array(seed), "batch": np.array(batch), "shape": np.array(seg.shape), "torch_version": np.array(torch.__version__),
          "loss": np.float64(loss.item()), "crop": seg[:, :, 40:48, 40:48, 40:48].numpy().copy(), "digest": digest(seg),
          "std": seg.double().std(dim=(0, 2, 3, 4)).numpy(),
          "head_gain_keys": np.array(sorted(HEAD_GAIN)), "head_gain": np.array([HEAD_GAIN[k] for k in sorted(HEAD_GAIN)])}
    fx.update(grads_summary(model.named_parameters()))
    for n, p in model.named_parameters():
        g = p.grad.detach().reshape(-1)
        fx["gsample/" + n] = g[torch.from_numpy(grad_sample_index(g.numel()))].numpy().copy()
    opt.step()
    for n, b in model.named_buffers():
        if not n.endswith("num_batches_tracked"):
            fx["buf1/" + n] = b.detach().numpy().copy()
    for n, p in model.named_parameters():
        fx["pnorm1/" + n] = np.float64(p.detach().double().norm().item())
    np.savez_compressed(os.path.join(HERE, fname), **fx)
    print("wrote", fname, "loss", fx["loss"], "std", fx["std"])


class _FakeLoader:
    """Just enough of DataLoader for ``Learner.__init__`` (Learner.py:40-42)."""
    batch_size = 2

    def __len__(self):
        return 1


def gen_cae(ch, seed, fname, d=28, hw=128):
    from oracle import weights as W
    from common.model.Cae3D import Cae3D, Enc3D, Dec3D
    from common.metrics import BatchDiceLoss
    from learner.CaeReconstructionLearner import CaeReconstructionLearner

    alpha = 1.0                                                    # train_shape_reconstruction.py:18
    enc = Enc3D(size_input_xy=hw, size_input_z=d, channels=ch, n_ch_global=5, alpha=alpha)
    dec = Dec3D(size_input_xy=hw, size_input_z=d, channels=ch, n_ch_global=5, alpha=alpha)
    cae = Cae3D(enc, dec)
    cae.load_state_dict(W.make_state_dict(W.cae_spec(ch), seed))
    params = [p for p in cae.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    learner = CaeReconstructionLearner(_FakeLoader(), None, cae, opt, None, n_epochs=1, path_previous_base=None,
                                       path_outputs_base="/tmp/_golden_cae", criterion=BatchDiceLoss([1.0]))
    labels, clinical = W.cae_inputs(2, d, hw, seed)
    batch = {"case_id": [0, 1], "images": None, "labels": labels, "clinical": clinical}
    fx = {"channels": np.array(ch), "seed": np.array(seed), "d": np.array(d), "hw": np.array(hw),
          "torch_version": np.array(torch.__version__)}
    cae.train()
    learner.adapt_betas(0)
    fx["betas_epoch0"] = np.array(opt.param_groups[0]["betas"])
    dto = learner.inference_step(batch)
    fx["ttt"] = dto.given_variables.time_to_treatment.detach().numpy().copy()
    for k in ("core", "penu", "lesion", "interpolation"):
        lat = getattr(dto.latents.gtruth, k)
        rec = getattr(dto.reconstructions.gtruth, k)
        fx["lat_digest/" + k] = digest(lat)
        fx["lat_head/" + k] = lat.detach().reshape(lat.shape[0], -1)[:, :64].numpy().copy()
        fx["rec_digest/" + k] = digest(rec)
        fx["rec_crop/" + k] = rec.detach()[:, 0, d // 2, 60:68, 60:68].numpy().copy()
    for ep in (0, 30, 60):
        fx["loss_epoch/%d" % ep] = np.float64(learner.loss_step(dto, ep).item())
    loss = learner.loss_step(dto, 30)
    opt.zero_grad()
    loss.backward()
    fx.update(grads_summary(cae.named_parameters()))
    opt.step()
    for n, b in cae.named_buffers():
        if n.endswith("num_batches_tracked"):
            fx["nbt/" + n] = b.numpy().copy()
        elif n.startswith("enc.encoder.0.") or n.startswith("dec.decoder.0.") or n.startswith("dec.decoder.33."):
            fx["buf1/" + n] = b.detach().numpy().copy()
    for n, p in list(cae.named_parameters())[:8]:
        fx["phead1/" + n] = p.detach().reshape(-1)[:8].numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **fx)
    print("\nwrote", fname, {k: float(v) for k, v in fx.items() if k.startswith("loss_epoch")})


def phase2_inputs(seed, d=28, hw=128):
    """synthetic batch of the phase-2 learners: labels / clinical as for phase 1, 'images' = soft U-Net-like segmentations
    of core and penumbra (another seed's blobs, squeezed into (0.05, 0.95))"""
    from oracle import weights as W
    labels, clinical = W.cae_inputs(2, d, hw, seed)
    seg, _ = W.cae_inputs(2, d, hw, seed + 7)
    images = (0.05 + 0.9 * seg[:, 0:2]).contiguous()
    return images, labels, clinical


def gen_cae_phase2(seed, fname, ch=(1, 16, 24, 32, 100, 200, 1), d=28, hw=128):
    """SURVEY 8(f) N4, second half: the reference's ``CaePredictionLearner`` (new encoder against the frozen CAE) and
    ``CaeStepLearner`` (learned interpolation step) on a synthetic batch.  The reference's own
    ``CaeEncInference.inference_step`` cannot run (it sets ``dto.mode``, the models read ``dto.flag``: the second model call
    asserts, recorded below as ``ref_inference_step_raises``); the harness makes the same five calls with ``dto.flag`` set to
    what ``dto.mode`` says -- every forward, the loss and the backward are the reference's own code."""
    from oracle import weights as W
    from common.model.Cae3D import Cae3D, Enc3D, Dec3D, Enc3DStep
    from common.metrics import BatchDiceLoss
    from learner.CaePredictionLearner import CaePredictionLearner
    from learner.CaeStepLearner import CaeStepLearner
    import common.dto.CaeDto as CaeDtoUtil
    ch = list(ch)
    kw = dict(size_input_xy=hw, size_input_z=d, channels=ch, n_ch_global=5, alpha=1.0)
    cae = Cae3D(Enc3D(**kw), Dec3D(**kw))
    cae.load_state_dict(W.make_state_dict(W.cae_spec(ch), seed))
    enc = Enc3D(**kw)
    enc.load_state_dict(W.make_state_dict(W.enc_spec(ch), seed + 1))
    images, labels, clinical = phase2_inputs(seed, d, hw)
    batch = {"case_id": [0, 1], "images": images, "labels": labels, "clinical": clinical}
    opt = torch.optim.Adam([p for p in enc.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    learner = CaePredictionLearner(_FakeLoader(), None, cae, enc, opt, None, n_epochs=1, path_previous_base=None,
                                   path_outputs_base="/tmp/_golden_cae2", criterion=BatchDiceLoss([1.0]))
    fx = {"channels": np.array(ch), "seed": np.array(seed), "d": np.array(d), "hw": np.array(hw),
          "torch_version": np.array(torch.__version__)}
    assert not any(p.requires_grad for p in cae.parameters())          # CaePredictionLearner.__init__ froze the CAE
    cae.train()
    try:
        learner.inference_step(batch)
        fx["ref_inference_step_raises"] = np.array(0)
    except AssertionError:
        fx["ref_inference_step_raises"] = np.array(1)
    # running statistics the failed attempt moved: reload, so that the fixture describes ONE clean pass
    cae.load_state_dict(W.make_state_dict(W.cae_spec(ch), seed))
    enc.load_state_dict(W.make_state_dict(W.enc_spec(ch), seed + 1))
    dto = learner.init_clinical_variables(batch, None)
    dto.flag = CaeDtoUtil.FLAG_INPUTS
    dto = learner.init_unet_segm_variables(batch, dto)
    dto = enc(dto)
    dto = cae.dec(dto)
    dto.flag = CaeDtoUtil.FLAG_GTRUTH
    dto = learner.init_gtruth_segm_variables(batch, dto)
    dto = cae(dto)
    fx["ttt"] = dto.given_variables.time_to_treatment.detach().numpy().copy()
    for k in ("core", "penu", "interpolation"):
        lat, rec = getattr(dto.latents.inputs, k), getattr(dto.reconstructions.inputs, k)
        fx["lat_in_head/" + k] = lat.detach().reshape(lat.shape[0], -1)[:, :64].numpy().copy()
        fx["lat_in_digest/" + k] = digest(lat)
        fx["rec_in_crop/" + k] = rec.detach()[:, 0, d // 2, 60:68, 60:68].numpy().copy()
        fx["rec_in_digest/" + k] = digest(rec)
        fx["lat_gt_digest/" + k] = digest(getattr(dto.latents.gtruth, k))
    loss = learner.loss_step(dto, 0)
    fx["loss"] = np.float64(loss.item())
    opt.zero_grad()
    loss.backward()
    fx.update(grads_summary(enc.named_parameters()))
    assert all(p.grad is None for p in cae.parameters())
    for n, b in list(cae.named_buffers()) + [("newenc." + n, b) for n, b in enc.named_buffers()]:
        if n.endswith("num_batches_tracked"):
            fx["nbt/" + n] = b.numpy().copy()
        elif n.startswith("dec.decoder.0.") or n.startswith("newenc.encoder.0."):
            fx["buf1/" + n] = b.detach().numpy().copy()
    # ---- CaeStepLearner: Enc3DStep around the frozen encoder stack, frozen decoder (train_interpolationstep_after_reconstruction.py:21-27)
    cae.load_state_dict(W.make_state_dict(W.cae_spec(ch), seed))
    cae.freeze(True)
    torch.manual_seed(seed)
    senc = Enc3DStep(**kw)
    senc.encoder = cae.enc.encoder
    cae2 = Cae3D(senc, cae.dec)
    cae2.train()
    for n, p in senc.named_parameters():
        if n.startswith("reduce.") or n.startswith("step."):
            fx["step_param/" + n] = p.detach().numpy().copy()
    params = [p for p in cae2.parameters() if p.requires_grad]
    opt2 = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999))
    learner2 = CaeStepLearner(_FakeLoader(), None, cae2, opt2, None, n_epochs=1, path_previous_base=None,
                              path_outputs_base="/tmp/_golden_cae1step", criterion=BatchDiceLoss([1.0]))
    dto2 = learner2.inference_step(batch)
    fx["step_value"] = senc._get_step(dto2).detach().numpy().copy()
    for k in ("penu", "interpolation"):
        rec = getattr(dto2.reconstructions.gtruth, k)
        fx["step_rec_crop/" + k] = rec.detach()[:, 0, d // 2, 60:68, 60:68].numpy().copy()
    loss2 = learner2.loss_step(dto2, 0)
    fx["step_loss"] = np.float64(loss2.item())
    opt2.zero_grad()
    loss2.backward()
    for n, p in senc.named_parameters():
        if p.requires_grad:
            fx["step_grad/" + n] = p.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, fname), **fx)
    print("wrote", fname, "loss", fx["loss"], "step loss", fx["step_loss"], "ref inference_step raises:", int(fx["ref_inference_step_raises"]))


def gen_checkpoints():
    """whole-module pickles exactly as ``Learner.save_model`` writes them (``torch.save(self._model.cpu(),
    path)``, learner/Learner.py:112-114) from the REFERENCE classes -- class paths ``common.model.Unet3D.Unet3D`` /
    ``common.model.Cae3D.Cae3D`` + tensors -- with the eval-mode outputs they produce, and the ``.optim`` / ``.json``
    companions of ``save_training`` (Learner.py:105-110).  Small channel counts keep the files small."""
    from oracle import weights as W
    from common.model.Unet3D import Unet3D
    from common.model.Cae3D import Cae3D, Enc3D, Dec3D
    import common.dto.UnetDto as UnetDtoUtil
    import common.dto.CaeDto as CaeDtoUtil
    ch = [2, 8, 8, 16, 8, 8, 8, 2]
    model = Unet3D(ch)
    model.load_state_dict(W.make_state_dict(W.unet_spec(ch), 41))
    x, _ = W.unet_inputs(1, 44, 41)
    model.eval()
    with torch.no_grad():
        dto = model(UnetDtoUtil.init_dto(x))
    torch.save(model.cpu(), os.path.join(HERE, "ref_unet.model"))
    fx = {"channels": np.array(ch), "seed": np.array(41), "size": np.array(44),
          "seg": torch.cat((dto.outputs.core, dto.outputs.penu), 1).numpy().copy()}
    # the optimiser state torch.optim.Adam saves after one step (Learner.py:108)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    x2, y2 = W.unet_inputs(2, 44, 42)
    from common.metrics import BatchDiceLoss
    crit = BatchDiceLoss([1.0])
    dto = model(UnetDtoUtil.init_dto(x2, y2[:, 0:1], y2[:, 1:2]))
    loss = (crit(dto.outputs.core, dto.given_variables.core) + crit(dto.outputs.penu, dto.given_variables.penu)) / 2
    opt.zero_grad(); loss.backward(); opt.step()
    torch.save(opt.state_dict(), os.path.join(HERE, "ref_unet.optim"))
    fx["optim_step"] = np.array(1)
    fx["exp_avg_head"] = opt.state_dict()["state"][1]["exp_avg"].reshape(-1)[:8].numpy().copy()
    cch = [1, 8, 8, 8, 8, 16, 1]
    enc = Enc3D(size_input_xy=64, size_input_z=28, channels=cch, n_ch_global=5, alpha=1.0)
    dec = Dec3D(size_input_xy=64, size_input_z=28, channels=cch, n_ch_global=5, alpha=1.0)
    cae = Cae3D(enc, dec)
    cae.load_state_dict(W.make_state_dict(W.cae_spec(cch), 43))
    labels, clinical = W.cae_inputs(1, 28, 64, 43)
    cae.eval()
    with torch.no_grad():
        cdto = CaeDtoUtil.init_dto(clinical.float(), torch.tensor([[[[[0.25]]]]]), None, None, None, None, None, None, None)
        cdto.given_variables.gtruth.core, cdto.given_variables.gtruth.penu, cdto.given_variables.gtruth.lesion = \
            labels[:, 0:1], labels[:, 1:2], labels[:, 2:3]
        cdto = cae(cdto)
    torch.save(cae.cpu(), os.path.join(HERE, "ref_cae.model"))
    fx["cae_channels"] = np.array(cch)
    fx["cae_seed"] = np.array(43)
    for k in ("core", "penu", "lesion", "interpolation"):
        fx["cae_rec/" + k] = getattr(cdto.reconstructions.gtruth, k).numpy().copy()[:, :, 10:18, 24:40, 24:40]
        fx["cae_lat/" + k] = getattr(cdto.latents.gtruth, k).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "ref_checkpoints.npz"), **fx)
    print("wrote ref_unet.model ref_unet.optim ref_cae.model ref_checkpoints.npz")


if __name__ == "__main__":
    torch.set_num_threads(8)
    import_reference()
    which = sys.argv[1:] or ["unet", "unet128", "unet128step", "cae", "cae2", "unet4", "ckpt"]
    if "unet" in which:
        gen_unet(44, 11, "unet_44.npz")
        gen_unet(48, 12, "unet_48.npz")
        gen_unet((44, 48, 52), 13, "unet_44x48x52.npz")
    if "ckpt" in which:
        gen_checkpoints()
    if "unet4" in which:      # BASELINE configs[4] topology (ch_bC = 32), smallest closed sizes
        ch4 = (2, 32, 64, 128, 256, 128, 64, 32, 32, 2)
        gen_unet(92, 31, "unet4_92.npz", ch4)
        gen_unet((92, 100, 96), 32, "unet4_92x100x96.npz", ch4)
    if "unet128" in which:
        gen_unet_eval128(14, "unet_eval128.npz")
        gen_unet_train128(15, "unet_train128.npz")
    if "unet128step" in which:
        gen_unet_trainstep128(15, "unet_trainstep128.npz")
    if "cae2" in which:
        gen_cae_phase2(23, "cae_phase2_200.npz")
    if "cae" in which:
        gen_cae([1, 16, 24, 32, 100, 200, 1], 21, "cae_200.npz")
        gen_cae([1, 16, 24, 32, 100, 800, 1], 22, "cae_800.npz")
