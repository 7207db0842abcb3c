"""Round-3 GPU parity tests: the reference-recorded 128^3 fixture through the HIP path, checkpoints written
while a captured step is live, and the precision modes added this round."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Unet3D import Unet3D, LargeUnet3D
import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil
from stroke_prediction_amd.runtime import lib as L
from stroke_prediction_amd.runtime import ops as O

CH = [2, 16, 32, 64, 32, 16, 32, 2]
CH4 = [2, 32, 64, 128, 256, 128, 64, 32, 32, 2]
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _build(ch, seed, dtype, cls=Unet3D):
    model = cls(ch, dtype=dtype)
    spec = W.unet_spec(ch)
    model.load_state_dict(W.make_state_dict(spec, seed))
    return model.to(DEV)


def _logit(p):
    p = p.double().clamp(1e-12, 1 - 1e-12)
    return torch.log(p / (1 - p))


# VERDICT r2 "missing" 5: the only reference-recorded fixture at the HEADLINE spatial size (tests/golden/make_golden.py:
# eval-mode forward of the reference's Unet3D at 1 x 2 x 128^3) against the HIP path in every precision mode.
# f32 (split-bf16 x3): crop <= 1e-4 abs, logits <= 1e-3 relative (north_star).  bf16 / f16 storage: stated per mode.
@pytest.mark.parametrize("dtype,tol_crop,tol_logit,tol_mean", [
    ("f32", 1e-4, 1e-3, 2e-6),
    ("bf16", 2e-2, 6e-2, 2e-4),
    ("f16", 3e-3, 8e-3, 3e-5),          # IEEE-half storage: 8x finer than bf16
])
def test_unet_eval128_fixture_through_the_hip_path(dtype, tol_crop, tol_logit, tol_mean):
    fx = np.load(os.path.join(GOLDEN, "unet_eval128.npz"))
    seed = int(fx["seed"])
    model = _build(CH, seed, dtype).eval()
    x, _ = W.unet_inputs(1, 128, seed)
    with torch.no_grad():
        dto = model(UnetDtoUtil.init_dto(x.to(DEV), None, None))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1).cpu()
    assert tuple(seg.shape) == tuple(int(v) for v in fx["shape"])
    crop = seg[:, :, 42:46, 42:46, 42:46]
    ref = torch.from_numpy(fx["crop"])
    assert float((crop - ref).abs().max()) <= tol_crop
    rel = float(((_logit(crop) - _logit(ref)).abs() / _logit(ref).abs().clamp_min(1.0)).max())
    assert rel <= tol_logit, rel
    np.testing.assert_allclose(seg.double().mean(dim=(0, 2, 3, 4)).numpy(), fx["mean"], rtol=0, atol=tol_mean)
    np.testing.assert_allclose(seg.double().std(dim=(0, 2, 3, 4)).numpy(), fx["std"], rtol=0.05 if dtype != "f32" else 2e-3)


# the eval-mode fixture above has an output std of 3.6e-4 (untouched running statistics), so its bf16 / f16
# rows bind only through mean and std.  unet_train128.npz is a TRAIN-mode forward of the reference at the same size with a gain
# on the classify weights: probabilities with std 0.10-0.12, logits up to +-6.3.  Tolerances of the 16-bit modes are fractions of
# that std; f32 and bf16x3 are held to the north-star numbers (probabilities 1e-4, logits 1e-3 of the logit range).
@pytest.mark.parametrize("dtype,tol_prob_std,tol_logit_range", [
    ("f32", None, 1e-3),
    ("f16x3", None, 1e-3),
    ("bf16x3", None, 1e-3),
    ("f16", 0.08, 8e-3),          # measured 0.050 of the output std / 3.9e-3 of the logit range
    ("bf16", 0.45, 4e-2),         # measured 0.32 / 2.1e-2: bf16 storage is 30 x the north-star tolerance at this size
])
def test_unet_train128_fixture_through_the_hip_path(dtype, tol_prob_std, tol_logit_range):
    fx = np.load(os.path.join(GOLDEN, "unet_train128.npz"))
    seed = int(fx["seed"])
    sd = W.make_state_dict(W.unet_spec(CH), seed)
    for k, gain in zip(fx["head_gain_keys"], fx["head_gain"]):
        sd[str(k)] = sd[str(k)] * float(gain)
    model = Unet3D(CH, dtype=dtype)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    x, _ = W.unet_inputs(1, 128, seed)
    with torch.no_grad():
        dto = model(UnetDtoUtil.init_dto(x.to(DEV), None, None))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1).cpu()
    assert tuple(seg.shape) == tuple(int(v) for v in fx["shape"])
    std = float(min(fx["std"]))
    assert std > 0.05
    worst_p, worst_l = 0.0, 0.0
    for key, crop in (("crop", seg[:, :, 40:48, 40:48, 40:48]), ("crop_corner", seg[:, :, :4, :4, -4:])):
        ref = torch.from_numpy(fx[key])
        worst_p = max(worst_p, float((crop - ref).abs().max()))
        worst_l = max(worst_l, float((_logit(crop) - _logit(ref)).abs().max()) / float(fx["logit_absmax"]))
    print("train128 %s: max |dp| %.2e (%.3f of the output std), max |dlogit| / max |logit| %.2e" % (dtype, worst_p, worst_p / std, worst_l))
    assert worst_p <= (1e-4 if tol_prob_std is None else tol_prob_std * std), worst_p
    assert worst_l <= tol_logit_range, worst_l
    tol_m = 2e-6 if tol_prob_std is None else tol_prob_std * std * 0.1
    np.testing.assert_allclose(seg.double().mean(dim=(0, 2, 3, 4)).numpy(), fx["mean"], rtol=0, atol=tol_m)
    np.testing.assert_allclose(seg.double().std(dim=(0, 2, 3, 4)).numpy(), fx["std"], rtol=2e-4 if tol_prob_std is None else 0.02)


def test_save_model_between_captured_steps_keeps_the_graph_valid(tmp_path):
    """``Learner.save_model`` no longer moves the live model, so a captured ``train_batch`` stays valid
    (and every data-parallel rank keeps issuing the same collectives).  Trajectory with a save after the capture == trajectory
    without one, bit for bit under replay."""
    from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
    from stroke_prediction_amd.common.metrics import BatchDiceLoss
    from stroke_prediction_amd.learner.UnetSegmentationLearner import UnetSegmentationLearner

    class Loader(list):
        batch_size = 2
    seed = 11
    x, y = W.unet_inputs(2, (52, 52, 52), seed)
    batch = {"case_id": [0, 1], "images": x.to(DEV), "labels": y.to(DEV), "clinical": torch.zeros(2, 5, 1, 1, 1)}
    out = {}
    for tag in ("plain", "saved"):
        model = _build(CH, seed, "f32").train()
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999), capturable=True)
        attach_flat_grads(model)
        learner = UnetSegmentationLearner(Loader([batch]), None, model, opt, None, 1, BatchDiceLoss([1.0]), None,
                                          str(tmp_path / tag), graph=True, batch_metrics=False)
        learner.GRAPH_WARMUP = 1
        losses = []
        for step in range(5):
            losses.append(learner.train_batch(batch, 0).loss)
            if tag == "saved" and step == 2:
                graphs = {k: v["graph"] for k, v in learner._graphs.items()}
                learner.save_model()
                assert {k: v["graph"] for k, v in learner._graphs.items()} == graphs and all(g is not None for g in graphs.values())
                loaded = torch.load(learner.path("save", learner.FNB_MODEL), weights_only=False)
                assert not next(loaded.parameters()).is_cuda
                for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
                    assert torch.equal(a.cpu(), b), k
        out[tag] = (losses, model.flat_buffers()[0].clone())
    # the captured steps replay the same kernels on the same buffers, and the reductions are ordered (sp_cols_sum; see
    # test_three_training_steps_are_bit_identical_from_run_to_run): the two trajectories are the same numbers
    assert [float(v) for v in out["plain"][0]] == [float(v) for v in out["saved"][0]], (out["plain"][0], out["saved"][0])
    assert torch.equal(out["plain"][1], out["saved"][1])


# ------------------------------------------------------------------------------------------------ CAE: concurrent passes
def _cae_step(ch, seed, d, hw, dtype, streams, graph=False, steps=1, batched=0, warmup=1):
    """one (or a few) CaeReconstructionLearner.train_batch steps; returns reconstructions, loss, the flat gradient after the
    first backward and every buffer -- with the 3 + 4 passes on one stream (streams = 0) or one stream each (2)"""
    from stroke_prediction_amd.common.model import Cae3D as M
    from stroke_prediction_amd.common.model.Cae3D import Cae3D, Enc3D, Dec3D
    from stroke_prediction_amd.common.metrics import BatchDiceLoss
    from stroke_prediction_amd.learner.CaeReconstructionLearner import CaeReconstructionLearner
    from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
    keep = M.CAE_STREAMS, M.CAE_BATCHED
    M.CAE_STREAMS, M.CAE_BATCHED = streams, batched
    try:
        cae = Cae3D(Enc3D(hw, d, ch, 5, 1.0, dtype=dtype), Dec3D(hw, d, ch, 5, 1.0, dtype=dtype))
        cae.load_state_dict(W.make_state_dict(W.cae_spec(ch), seed))
        cae = cae.to(DEV).train()
        opt = FusedAdam([p for p in cae.parameters()], lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999), capturable=True)
        attach_flat_grads(cae)

        class Loader(list):
            batch_size = 2
        learner = CaeReconstructionLearner(Loader(), None, cae, opt, None, 1, None, "/tmp/_cae_r3", BatchDiceLoss([1.0]),
                                           verbose=False, graph=graph, batch_metrics=False)
        learner.GRAPH_WARMUP = warmup
        labels, clinical = W.cae_inputs(2, d, hw, seed)
        batch = {"case_id": [0, 1], "images": None, "labels": labels.to(DEV), "clinical": clinical.to(DEV)}
        dto = learner.inference_step(batch)
        loss = learner.loss_step(dto, 30)
        opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        rec = {k: getattr(dto.reconstructions.gtruth, k).detach().clone() for k in ("core", "penu", "lesion", "interpolation")}
        grad = cae.flat_buffers()[1].clone()
        bufs = {k: v.detach().clone() for k, v in cae.named_buffers()}
        losses = []
        for _ in range(steps):
            losses.append(float(learner.train_batch(batch, 30).loss))
        torch.cuda.synchronize()
        return rec, float(loss.detach()), grad, bufs, losses, cae.flat_buffers()[0].clone()
    finally:
        M.CAE_STREAMS, M.CAE_BATCHED = keep


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_cae_concurrent_passes_equal_sequential_passes(dtype):
    """VERDICT r2 item 2 (structure of the CAE step): the 3 encoder / 4 decoder passes of a call on one stream each
    (SP_CAE_STREAMS) -- same reconstructions, loss and BatchNorm buffers (the running statistics are updated in pass order
    through per-layer events), gradients equal up to the order of the seven partial sums (each pass accumulates into its own
    buffer, added in issue order)."""
    ch = [1, 16, 24, 32, 100, 200, 1]
    a = _cae_step(ch, 23, 28, 64, dtype, 0)
    a2 = _cae_step(ch, 23, 28, 64, dtype, 0)
    b = _cae_step(ch, 23, 28, 64, dtype, 2)
    # The same kernels on the same data.  Two SEQUENTIAL runs already differ by the run-to-run noise of the fp64 statistics
    # atomics (1e-7 relative in a BatchNorm scale; in bf16 that flips the rounding of isolated activations and spreads over the
    # 22 layers): the concurrent run must stay within three times that distance of the sequential one (floors: f32 / bf16).
    # (bf16: two runs are sometimes bit-identical and sometimes 1e-3 apart on average -- a flipped rounding in the 200-value
    # BatchNorm of the latent moves everything behind it; the floors are that spread, a race would show as garbage)
    fl = dict(f32=(2e-5, 2e-6, 2e-4), bf16=(5e-2, 3e-3, 8e-2))[dtype]
    for k in a[0]:
        d, d0 = (a[0][k] - b[0][k]).abs(), (a[0][k] - a2[0][k]).abs()
        assert float(d.max()) <= 3 * float(d0.max()) + fl[0] and float(d.mean()) <= 3 * float(d0.mean()) + fl[1], \
            (k, float(d.max()), float(d0.max()), float(d.mean()), float(d0.mean()))
    assert abs(a[1] - b[1]) <= 3 * abs(a[1] - a2[1]) + fl[0]
    for k in a[3]:
        if k.endswith("num_batches_tracked"):
            assert int(a[3][k]) == int(b[3][k]) == (3 if k.startswith("enc.") else 4), k
        else:      # running statistics: pass order kept
            d, d0 = float((a[3][k] - b[3][k]).abs().max()), float((a[3][k] - a2[3][k]).abs().max())
            assert d <= 3 * d0 + (1e-5 if dtype == "f32" else 2e-3) * float(a[3][k].abs().max() + 1e-3), (k, d, d0)
    rel = float((a[2] - b[2]).double().norm() / a[2].double().norm())
    rel0 = float((a[2] - a2[2]).double().norm() / a[2].double().norm())
    print("concurrent vs sequential gradient %.2e, sequential vs sequential %.2e" % (rel, rel0))
    assert rel < 3 * rel0 + fl[2], (rel, rel0)


def test_cae_graph_mode_with_concurrent_passes_follows_the_eager_trajectory():
    """Learner(graph=True) captures the step with the passes as parallel graph branches: four replayed steps stay on the
    trajectory of the sequential eager steps (bf16 run-to-run noise of the fp64 statistics atomics only)."""
    ch = [1, 16, 24, 32, 100, 200, 1]
    a = _cae_step(ch, 23, 28, 64, "f32", 0, graph=False, steps=4)
    b = _cae_step(ch, 23, 28, 64, "f32", 1, graph=True, steps=4)
    np.testing.assert_allclose(a[4], b[4], rtol=0, atol=3e-4)
    assert b[4][-1] < b[4][0]
    assert float((a[5] - b[5]).abs().max()) < 8e-3


# ------------------------------------------------------------------------------------------------ RCCL through the C ABI
def test_direct_communicator_all_reduce_eager_and_captured():
    """VERDICT r2 missing 3: ``sp_allreduce_flat`` & co (include/stroke_amd.h) on a communicator created through the C ABI.
    One rank is all a one-GPU box allows (RCCL refuses two ranks per device): the sum over one rank is the identity, which
    still exercises id creation, communicator init, the collective on our own stream, the two-shot form, the stream join and --
    what the torch.distributed path could not rehearse -- the collective captured inside a hipGraph as a forked branch."""
    from stroke_prediction_amd.parallel import DirectComm
    from stroke_prediction_amd.runtime import lib as L
    assert L.load().sp_comm_available() == 1
    comm = DirectComm()
    assert comm.world == 1 and comm.rank == 0
    g = torch.Generator(device=DEV).manual_seed(3)
    buf = torch.randn(355014, generator=g, device=DEV)
    ref = buf.clone()
    comm.all_reduce_async(buf)
    comm.all_reduce_async(buf[1000:9000])          # a bucket: a slice of the flat buffer, 4-byte aligned only
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(buf, ref)
    comm.all_reduce_async(buf[:355008], two_shot=True)
    comm.wait()
    torch.cuda.synchronize()
    assert torch.equal(buf, ref)
    # captured: scale -> all-reduce on the communicator's stream (a fork) -> join -> scale
    static = ref.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        static.mul_(2.0)
        comm.all_reduce_async(static)
        comm.wait()
        static.add_(1.0)
    static.copy_(ref)
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.allclose(static, (ref * 2 + 1) * 2 + 1)
    comm.close()


# ------------------------------------------------------------------------------------------------ f16 precision mode
def test_f16_mode_train_step_matches_the_oracle_and_the_fixture():
    """a fast mode that is closer to the 1e-3 logit target than bf16 storage can be.
    ``Unet3D(dtype="f16")`` = the same kernels built for IEEE-half storage (libstroke_amd_f16.so), output gradients scaled by a
    power of two inside the backward.  Against the oracle with the same storage roundings (q=round_f16) and against the fp32
    oracle / the reference fixture: outputs, loss, every gradient tensor (rel-L2, direction), and the gradients must come out
    UNSCALED."""
    seed, size = 11, (44, 44, 44)
    x, y = W.unet_inputs(2, size, seed)
    out = {}
    for q, tag in ((nets.round_f16, "emul"), (nets._ident, "f32")):
        sd = W.make_state_dict(W.unet_spec(CH), seed)
        names = nets.trainable(sd)
        for k in names:
            sd[k].requires_grad_(True)
        seg = nets.unet_forward(sd, x, training=True, q=q)
        loss = nets.unet_loss(seg, y)
        out[tag] = (seg.detach(), float(loss.detach()), dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names]))))
    model = _build(CH, seed, "f16").train()
    dto = model(UnetDtoUtil.init_dto(x.to(DEV), y[:, 0:1].to(DEV), y[:, 1:2].to(DEV)))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    loss = nets.unet_loss(seg, y.to(DEV))
    loss.backward()
    eng = next(iter(model._engines.values()))
    assert eng.variant == "f16" and eng.loss_scale >= 2.0 and eng.conv[1][1].y.dtype == torch.float16
    s = seg.detach().cpu()
    d_emul, d_f32 = float((s - out["emul"][0]).abs().max()), float((s - out["f32"][0]).abs().max())
    print("f16 mode: max |seg - f16-emulating oracle| %.2e, max |seg - fp32 oracle| %.2e" % (d_emul, d_f32))
    assert d_emul < 1.5e-3 and d_f32 < 3e-3            # bf16 mode: 8e-3 / 2e-2 (tests/test_gpu_unet.py)
    lg, lr = _logit(s), _logit(out["f32"][0])
    # (measured 4.5e-3 .. 6.1e-3 from run to run -- the order of the statistics atomics moves the BatchNorm scales in their last
    # bits; the bf16 mode sits at 5e-2)
    assert float((lg - lr).abs().max() / lr.abs().max()) < 1e-2
    assert abs(float(loss.detach()) - out["f32"][1]) < 1e-3
    fx = np.load(os.path.join(GOLDEN, "unet_44.npz"))
    assert float((s - torch.from_numpy(fx["seg"])).abs().max()) < 3e-3
    bad = []
    for k, p in model.named_parameters():
        a, b = p.grad.detach().cpu().double(), out["f32"][2][k].double()
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        small = b.numel() <= 64
        # the bound is the spread of this quantity, not its value for one seed: at 4 x 4 x 4 outputs the parameter gradients are
        # sums with heavy cancellation and a last-bit change of one BatchNorm statistic redraws the f16 rounding noise of
        # everything behind it.  Measured over seeds 11..16 (tools/probes/f16_rel.py), first-layer weight (the worst tensor):
        # 0.113 0.148 0.111 0.110 0.086 0.179 -- and 0.140 0.150 0.111 0.114 0.085 0.182 after the concatenation kernel's
        # workgroups were re-ordered (another grouping of the same fp32 partial sums); classify.0: 0.006 .. 0.065.
        if rel > (0.5 if small else 0.25) or cos < (0.9 if small else 0.97):
            bad.append((k, rel, cos))
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ CAE: batched passes
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_cae_batched_passes_equal_sequential_passes(dtype):
    """the 3 encoder / 4 decoder passes of a call stacked along the batch axis (SP_CAE_BATCHED: one launch
    per layer for the convolution, weight gradient and data gradient of all passes; per-pass BatchNorm statistics, scale /
    shift tables and backward coefficients; running statistics updated in pass order by one grouped finalize kernel) against
    the pass-by-pass execution: reconstructions, loss, every BatchNorm buffer incl. num_batches_tracked (+= 3 / 4), gradients.
    f32: tight.  bf16: the batched path normalises every layer's input into a stored tensor where the sequential path folds
    the BatchNorm into the weights of un-padded layers or applies it on the operand load -- same function, other rounding
    points, so bf16 is held to the distance between two bf16 pipelines (as against the emulating oracle, test_gpu_cae.py)."""
    ch = [1, 16, 24, 32, 100, 200, 1]
    a = _cae_step(ch, 23, 28, 64, dtype, 0, batched=0)
    b = _cae_step(ch, 23, 28, 64, dtype, 0, batched=1)
    tol = dict(f32=(1e-4, 1e-5, 2e-5, 2e-3), bf16=(6e-2, 4e-3, 5e-3, 0.12))[dtype]
    for k in a[0]:
        d = (a[0][k] - b[0][k]).abs()
        assert float(d.max()) <= tol[0] and float(d.mean()) <= tol[1], (k, float(d.max()), float(d.mean()))
    assert abs(a[1] - b[1]) <= tol[2], (a[1], b[1])
    for k in a[3]:
        if k.endswith("num_batches_tracked"):
            assert int(a[3][k]) == int(b[3][k]) == (3 if k.startswith("enc.") else 4), k
        else:
            d = float((a[3][k] - b[3][k]).abs().max())
            # (bf16: the batched path folds the BatchNorm of its padded layers per group -- weights rounded after the fold -- where the
            # sequential one normalises on the operand load: 5.8e-3 measured on the latent's running mean)
            assert d <= (2e-5 if dtype == "f32" else 1e-2) * float(a[3][k].abs().max() + 1e-2), (k, d)
    rel = float((a[2] - b[2]).double().norm() / a[2].double().norm())
    print("batched vs sequential: flat gradient rel-L2 %.2e (%s)" % (rel, dtype))
    assert rel < tol[3], rel


# ------------------------------------------------------------------------------------------------ run-to-run reproducibility
def _three_steps(ch, dtype, size, cls, graph):
    """parameters / buffers / losses after three Learner-style steps of a fresh model"""
    from stroke_prediction_amd.optim import FusedAdam
    scales = 4 if cls is LargeUnet3D else 3
    x, y = W.unet_inputs(2, size, 11, scales=scales) if scales == 4 else W.unet_inputs(2, size, 11)
    xd, yd = x.to(DEV), y.to(DEV)
    model = _build(ch, 11, dtype, cls).train()
    opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999), capturable=graph)
    losses = []
    for _ in range(3):
        dto = model(UnetDtoUtil.init_dto(xd, yd[:, 0:1], yd[:, 1:2]))
        loss = nets.unet_loss(torch.cat((dto.outputs.core, dto.outputs.penu), 1), yd)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    return ([p.detach().clone() for p in model.parameters()], [b.detach().clone() for b in model.buffers()], losses)


@pytest.mark.parametrize("dtype,cls,ch,size", [("f32", Unet3D, [2, 16, 32, 64, 32, 16, 32, 2], (44, 48, 52)),
                                               ("bf16", Unet3D, [2, 16, 32, 64, 32, 16, 32, 2], (76, 76, 76)),
                                               ("fp8", LargeUnet3D, [2, 32, 64, 128, 256, 128, 64, 32, 32, 2], (100, 92, 96))])
def test_three_training_steps_are_bit_identical_from_run_to_run(dtype, cls, ch, size):
    """VERDICT r2 item 5b.  Every within-workgroup reduction adds its waves up in wave order (csrc/sp_common.h:sp_cols_sum; no LDS
    float atomics), the weight gradients go through ordered partial blocks, and what the workgroups then add into the fp64
    replica rows are fp32 values whose sum is exact in 53 bits whenever the addends of a row span less than ~2^21 -- so two runs
    agree bit for bit in practice: parameters, BatchNorm buffers and losses after three Adam steps (where a 1e-7 difference
    of a gradient would already have moved bottleneck weights by 2 lr)."""
    from stroke_prediction_amd.runtime import f8 as F8
    keep = F8.F8_MIN_PLANES
    F8.F8_MIN_PLANES = 8
    try:
        a = _three_steps(ch, dtype, size, cls, False)
        b = _three_steps(ch, dtype, size, cls, False)
    finally:
        F8.F8_MIN_PLANES = keep
    for (la, lb) in zip(a[2], b[2]):
        assert torch.equal(la, lb), (float(la), float(lb))
    worst = max(float((pa.double() - pb.double()).abs().max()) for pa, pb in zip(a[0] + a[1], b[0] + b[1]))
    assert worst == 0.0, worst


@pytest.mark.parametrize("graph", [False, True])
def test_cae_training_steps_are_bit_identical_from_run_to_run(graph):
    """the same for the CAE step (3 + 4 batched passes, Learner.train_batch eager and as a replayed hipGraph with its
    side-stream forks): reconstructions, gradients, buffers, losses and parameters after three steps"""
    ch = [1, 16, 24, 32, 100, 200, 1]
    # (two eager warm-up steps: the helper's own backward leaves every packed weight current, so the first one re-packs nothing --
    # the tables a capture replays are built by the first step that follows an optimizer update)
    a = _cae_step(ch, 23, 28, 64, "bf16", 0, graph=graph, steps=4, batched=1, warmup=2)
    b = _cae_step(ch, 23, 28, 64, "bf16", 0, graph=graph, steps=4, batched=1, warmup=2)
    assert a[1] == b[1] and a[4] == b[4], (a[1], b[1], a[4], b[4])
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), k
    assert torch.equal(a[2], b[2]) and torch.equal(a[5], b[5])
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def test_static_batch_skips_the_input_copy_and_changes_nothing():
    """Learner.static_batch: a batch that already lives in the captured step's input buffers is not copied again, and the
    trajectory equals the one of ordinary batches (bit for bit: reproducible reductions)."""
    from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
    from stroke_prediction_amd.common.metrics import BatchDiceLoss
    from stroke_prediction_amd.learner.UnetSegmentationLearner import UnetSegmentationLearner

    class Loader(list):
        batch_size = 2
    x, y = W.unet_inputs(2, (52, 52, 52), 3)
    out = {}
    for tag in ("plain", "static"):
        batch = {"case_id": [0, 1], "images": x.to(DEV), "labels": y.to(DEV), "clinical": None}
        model = _build(CH, 3, "bf16").train()
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999), capturable=True)
        attach_flat_grads(model)
        learner = UnetSegmentationLearner(Loader([batch]), None, model, opt, None, 1, BatchDiceLoss([1.0]), None, "/tmp/_sb_" + tag,
                                          graph=True, batch_metrics=False)
        learner.GRAPH_WARMUP = 1
        if tag == "static":
            sb = learner.static_batch(batch, 0)
            assert sb["images"].data_ptr() != batch["images"].data_ptr() and torch.equal(sb["images"], batch["images"])
            batch = sb
        losses = [learner.train_batch(batch, 0).loss for _ in range(4)]
        if tag == "static":       # the buffers are the graph's: a new batch written into them is what the next replay trains on
            g = next(iter(learner._graphs.values()))
            assert g["static"]["images"].data_ptr() == batch["images"].data_ptr() and g["graph"] is not None
            batch["images"].mul_(0.5)
            l5 = learner.train_batch(batch, 0).loss
            assert l5 != losses[-1]
        out[tag] = (losses, model.flat_buffers()[0].clone() if tag == "plain" else None)
    assert out["plain"][0] == out["static"][0], (out["plain"][0], out["static"][0])


@pytest.mark.parametrize("cout,cin,cop,cip,nparts", [(128, 128, 128, 128, 8), (250, 150, 256, 160, 12), (64, 256, 64, 256, 32)])
def test_tiled_weight_gradient_finish_matches_torch(cout, cin, cop, cip, nparts):
    """sp_wgrad_finish_folded on few, large partial blocks (the 64..384-channel layers of the 4-scale network): the tiled finish
    (one output channel x 32 input channels x all taps per workgroup, dw written as contiguous runs) against torch -- the sum
    over the blocks, the folded BatchNorm, the bias gradient and the BatchNorm-backward sums reduced over the taps"""
    g = torch.Generator(device=DEV).manual_seed(cout + cin)
    acc = torch.randn(nparts, 27, cop, cip, generator=g, device=DEV)
    scale = torch.rand(cip, generator=g, device=DEV) + 0.5
    shift = torch.randn(cip, generator=g, device=DEV) * 0.1
    dbias = torch.randn(cop, generator=g, device=DEV).double()
    w = torch.randn(cout, cin, 27, generator=g, device=DEV)
    dw = torch.randn(cout, cin, 27, generator=g, device=DEV)
    dw0 = dw.clone()
    db = torch.zeros(cout, device=DEV)
    nrep = 4
    bn = torch.zeros(nrep, cip, 2, dtype=torch.float64, device=DEV)
    tapsrc = torch.arange(27, dtype=torch.int32, device=DEV)
    L.call("sp_wgrad_finish_folded", O.ptr(acc), nparts, O.ptr(tapsrc), 27, cop, cip, cout, cin, cin * 27, 27, O.ptr(scale), O.ptr(shift),
           O.ptr(dbias), O.ptr(dw), O.ptr(db), O.ptr(w), O.ptr(bn), nrep, 0, 0, O.stream())
    a = acc.sum(0)[:, :cout, :cin].permute(1, 2, 0)                     # [co][ci][tap]
    ref = dw0 + scale[:cin].view(1, -1, 1) * a + shift[:cin].view(1, -1, 1) * dbias[:cout].float().view(-1, 1, 1)
    torch.testing.assert_close(dw, ref, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(db, dbias[:cout].float(), rtol=1e-6, atol=1e-6)
    s0 = (w.double() * dbias[:cout].view(-1, 1, 1)).sum((0, 2))
    s1 = (w.double() * a.double()).sum((0, 2))
    got = bn.sum(0)
    torch.testing.assert_close(got[:cin, 0], s0, rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(got[:cin, 1], s1, rtol=1e-5, atol=1e-2)
    assert float(got[cin:].abs().max()) == 0.0 if cip > cin else True
