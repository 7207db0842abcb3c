"""End-to-end U-Net parity on the GPU: drop-in ``Unet3D`` (HIP path) vs the CPU oracle and vs the
golden fixtures recorded from the real reference (forward, loss, every parameter gradient, BatchNorm
running statistics).  Tolerances: parity mode ("f32", split-bf16 MFMA) 1e-3 relative as BASELINE.json's
north_star asks; fast mode ("bf16") is held to bf16 storage noise and stated per assertion."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets, weights as W
from stroke_prediction_amd.common.model.Unet3D import Unet3D
import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil

CH = [2, 16, 32, 64, 32, 16, 32, 2]
DEV = "cuda:0"


def build(seed, dtype):
    model = Unet3D(CH, dtype=dtype)
    model.load_state_dict(W.make_state_dict(W.unet_spec(CH), seed))
    return model.to(DEV)


def oracle_step(seed, x, y, q=nets._ident):
    sd = W.make_state_dict(W.unet_spec(CH), seed)
    names = nets.trainable(sd)
    for k in names:
        sd[k].requires_grad_(True)
    seg = nets.unet_forward(sd, x, training=True, q=q)
    loss = nets.unet_loss(seg, y)
    grads = torch.autograd.grad(loss, [sd[k] for k in names])
    return seg.detach(), loss.item(), dict(zip(names, grads)), sd


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


# Gradient tolerances are relative L2 per parameter tensor.  LeakyReLU'(z) jumps 100x at z = 0, so ANY
# re-ordering of fp32 arithmetic flips a few elements near the kink (one flipped element of the 6^3 x 64
# map at 44^3 is already 7e-3 in L2, measured with tools/debug_unet_bwd.py); away from such flips the
# parity mode agrees to ~1e-5.  The bf16 path is compared with the oracle run with the SAME bf16 storage
# points (q=round_bf16): against the pure-fp32 trajectory 1-3 % of the signs differ per layer, which
# says nothing about the kernels.
@pytest.mark.parametrize("dtype,size,seed,tol_seg,tol_grad", [
    ("f32", (44, 44, 44), 11, 1e-4, 3e-2),
    ("f32", (44, 48, 52), 13, 1e-4, 3e-2),
    # bf16 against the EMULATING oracle, measured (tools/probes/bf16_grad_probe.py; 44^3 / 48^3 / 60^3): tensors > 64 elements worst
    # 0.19-0.24, median 0.15-0.18 -- two bf16 pipelines with different rounding points decorrelate at the LeakyReLU kinks; the bound
    # below plus the median bound in the body replace round 3's 0.3.  (The modes whose gradients ARE close to fp32: f16x3 / bf16x3,
    # tests/test_gpu_bf16x3.py.)
    ("bf16", (44, 44, 44), 11, 8e-3, 0.26),
    ("bf16", (48, 48, 48), 12, 8e-3, 0.26),
])
def test_unet_train_step_matches_oracle(dtype, size, seed, tol_seg, tol_grad):
    x, y = W.unet_inputs(2, size, seed)
    seg_ref, loss_ref, g_ref, sd_ref = oracle_step(seed, x, y, nets.round_bf16 if dtype == "bf16" else nets._ident)
    if dtype == "bf16":   # and the fast path stays close to the true fp32 reference on the outputs
        seg32, _, g32, _ = oracle_step(seed, x, y)
    model = build(seed, dtype)
    model.train()
    dto = model(UnetDtoUtil.init_dto(x.to(DEV), y[:, 0:1].to(DEV), y[:, 1:2].to(DEV)))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    assert tuple(seg.shape) == tuple(seg_ref.shape)
    # probabilities: absolute tolerance (values in (0,1)); logits relative tolerance for the parity mode
    torch.testing.assert_close(seg.detach().cpu(), seg_ref, rtol=0, atol=tol_seg)
    if dtype == "bf16":
        torch.testing.assert_close(seg.detach().cpu(), seg32, rtol=0, atol=2e-2)
    if dtype == "f32":
        logit = lambda p: torch.log(p / (1 - p))
        lr, lg = logit(seg_ref.double()), logit(seg.detach().cpu().double())
        assert float((lg - lr).abs().max() / lr.abs().max()) < 1e-3     # north_star: logits within 1e-3 rel
    loss = nets.unet_loss(seg, y.to(DEV))           # Dice recipe in torch (plumbing) on the HIP outputs
    assert abs(loss.item() - loss_ref) < (1e-5 if dtype == "f32" else 5e-3)
    loss.backward()
    bad = []
    if dtype == "bf16":
        big = sorted(rel_l2(p.grad.cpu(), g_ref[n]) for n, p in model.named_parameters() if p.numel() > 64)
        assert big[len(big) // 2] < 0.2, big
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        e = rel_l2(p.grad.cpu(), g_ref[name])
        cos = float(torch.nn.functional.cosine_similarity(p.grad.cpu().reshape(1, -1).double(),
                                                          g_ref[name].reshape(1, -1).double()))
        small = p.numel() <= 64 and dtype == "bf16"      # 2..64-element BatchNorm / bias gradients are noisier
        # parity mode: a single LeakyReLU branch flip (run-to-run: the fp64 BatchNorm atomics change the last bit of a
        # scale, see test_gpu_parallel_exact.py) moves a 16..64-element BatchNorm gradient by up to ~4 %: 2x head-room
        tol = 0.6 if small else (2 * tol_grad if (dtype == "f32" and p.numel() <= 64) else tol_grad)
        if e > tol or cos < (0.88 if small else 0.95):
            # Noise floor of a small, cancellation-heavy gradient under bf16 storage = how far the storage-point
            # emulation itself lands from the pure-fp32 oracle.  (The first BatchNorm's gamma is the extreme case: the
            # following conv -> BatchNorm makes the loss nearly invariant to it, its true gradient is ~0 and the
            # emulated and fp32 oracles differ by 2.5x / 27x per element at 48^3.)  Within twice that floor of either
            # reference the HIP value carries as much information as the emulation does.
            if small:
                noise = float((g_ref[name].double() - g32[name].double()).norm())
                d_emul = float((p.grad.cpu().double() - g_ref[name].double()).norm())
                d_fp32 = float((p.grad.cpu().double() - g32[name].double()).norm())
                if min(d_emul, d_fp32) <= 2.0 * noise:
                    continue
            bad.append((name, e, cos, rel_l2(p.grad.cpu(), g32[name]) if dtype == "bf16" else None))
    assert not bad, bad
    # BatchNorm running statistics followed the reference update rule (momentum 0.1, unbiased variance)
    for name, b in model.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == 1
        else:
            torch.testing.assert_close(b.cpu(), sd_ref[name], rtol=5e-3 if dtype == "f32" else 3e-2,
                                       atol=1e-4 if dtype == "f32" else 3e-3)


@pytest.mark.parametrize("fname", ["unet_44.npz", "unet_48.npz", "unet_44x48x52.npz"])
def test_unet_matches_reference_fixture(golden_dir, fname):
    """HIP path (parity mode) directly against outputs/gradients recorded from the REAL reference."""
    fx = np.load(os.path.join(golden_dir, fname))
    seed = int(fx["seed"])
    size = tuple(int(s) for s in np.atleast_1d(fx["size"]))
    size = size * 3 if len(size) == 1 else size
    x, y = W.unet_inputs(2, size, seed)
    model = build(seed, "f32")
    model.train()
    dto = model(UnetDtoUtil.init_dto(x.to(DEV)))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    np.testing.assert_allclose(seg.detach().cpu().numpy(), fx["seg"], rtol=0, atol=1e-4)
    loss = nets.unet_loss(seg, y.to(DEV))
    assert abs(loss.item() - float(fx["loss/0"])) < 1e-5
    loss.backward()
    for name, p in model.named_parameters():
        gn = float(fx["gnorm/" + name])
        assert abs(float(p.grad.double().norm()) - gn) <= 3e-2 * gn + 1e-9, name     # kink flips, see above
        np.testing.assert_allclose(p.grad.reshape(-1)[:8].cpu().numpy(), fx["ghead/" + name], rtol=3e-2,
                                   atol=6e-2 * gn + 1e-9)


def test_unet_eval_mode_and_freeze():
    seed = 11
    x, _ = W.unet_inputs(1, 64, seed)
    sd = W.make_state_dict(W.unet_spec(CH), seed)
    with torch.no_grad():
        ref = nets.unet_forward(sd, x, training=False)
    model = build(seed, "f32")
    model.freeze(True)
    model.eval()
    dto = model(UnetDtoUtil.init_dto(x.to(DEV)))
    seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
    assert not seg.requires_grad
    torch.testing.assert_close(seg.cpu(), ref, rtol=0, atol=1e-4)
    for name, b in model.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == 0


def test_rejects_cpu_and_too_small_inputs():
    model = Unet3D(CH)
    with pytest.raises(RuntimeError):
        model(UnetDtoUtil.init_dto(torch.zeros(1, 2, 44, 44, 44)))
    model = model.to(DEV)
    with pytest.raises(ValueError):          # BASELINE config #1 as written (32^3) cannot pass valid convs
        model(UnetDtoUtil.init_dto(torch.zeros(2, 2, 32, 32, 32, device=DEV)))


def test_checkpoint_round_trip_and_tester(tmp_path):
    """N3 / N1 of SURVEY 8f: the whole module pickles like the reference's (``torch.save(model)`` in
    ``Learner.save_model`` ``Learner.py:93,113``), reloads under the reference's import path and gives the same
    outputs; ``Tester`` (``tester/Tester.py:16-28``) drives it forward-only, one case at a time, on a non-cubic volume."""
    from stroke_prediction_amd.tester.Tester import Tester
    seed = 13
    model = build(seed, "f32")
    x, y = W.unet_inputs(2, (44, 48, 52), seed)
    model.train()
    dto = model(UnetDtoUtil.init_dto(x.to(DEV), y[:, 0:1].to(DEV), y[:, 1:2].to(DEV)))   # engines + BatchNorm buffers now live
    nets.unet_loss(torch.cat((dto.outputs.core, dto.outputs.penu), 1), y.to(DEV)).backward()
    path = str(tmp_path / "ckpt_unet.model")
    torch.save(model.cpu(), path)                                 # Learner.save_model moves to the CPU first
    model.to(DEV)
    loaded = torch.load(path, weights_only=False)
    assert type(loaded).__module__.endswith("common.model.Unet3D") and type(loaded).__name__ == "Unet3D"
    for (n1, a), (n2, b) in zip(sorted(model.state_dict().items()), sorted(loaded.state_dict().items())):
        assert n1 == n2 and torch.equal(a.cpu(), b.cpu()), n1

    class Loader(list):
        batch_size = 1

    class UnetTester(Tester):
        def inference_step(self, batch):
            from stroke_prediction_amd.common.inference.UnetInference import UnetInference
            return UnetInference.inference_step(self, batch)

    batches = Loader({"images": x[i:i + 1], "labels": y[i:i + 1], "case_id": [i]} for i in range(2))
    tester = UnetTester(batches, path)
    tester._model.to(DEV)                                          # the reference's scripts move the loaded model to the GPU
    assert not any(p.requires_grad for p in tester._model.parameters()) and not tester._model.training
    model.eval()
    for i, batch in enumerate(batches):
        _, out = tester.infer_batch(batch)
        with torch.no_grad():
            ref = model(UnetDtoUtil.init_dto(x[i:i + 1].to(DEV)))
        torch.testing.assert_close(out.outputs.core, ref.outputs.core, rtol=0, atol=1e-6)
        torch.testing.assert_close(out.outputs.penu, ref.outputs.penu, rtol=0, atol=1e-6)


def test_graph_captured_step_with_stream_overlap_matches_eager():
    """The optimiser step replayed from a hipGraph -- where the weight-gradient kernels run on a second stream beside the
    data-gradient convolutions (ops.overlap_level() == 2 only during capture) -- gives the gradients of the eager,
    single-stream step (bf16, same weights and batch)."""
    from stroke_prediction_amd.optim import attach_flat_grads
    from stroke_prediction_amd.runtime import ops as O
    from stroke_prediction_amd.common.metrics import BatchDiceLoss, mean_of_channel_losses
    seed = 21
    model = build(seed, "bf16")
    model.train()
    attach_flat_grads(model)
    x, y = W.unet_inputs(2, (52, 52, 52), seed)
    xd, yd = x.to(DEV), y.to(DEV)
    crit = BatchDiceLoss([1.0])

    def fwd_bwd():
        dto = model(UnetDtoUtil.init_dto(xd, yd[:, 0:1], yd[:, 1:2]))
        loss = mean_of_channel_losses(crit, (dto.outputs.core, dto.outputs.penu), (dto.given_variables.core, dto.given_variables.penu))
        model._flat_grad.zero_()
        loss.backward()
        return loss

    assert O.overlap_level() == 0                       # eager: one stream
    fwd_bwd()
    torch.cuda.synchronize()
    g0 = model._flat_grad.clone()
    le = fwd_bwd()
    torch.cuda.synchronize()
    ge = model._flat_grad.clone()
    seen = []
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        seen.append(O.overlap_level())
        lg = fwd_bwd()
    assert seen == [2]                                  # the captured step forks the weight gradients
    model._flat_grad.zero_()
    graph.replay()
    torch.cuda.synchronize()
    gg = model._flat_grad.clone()
    assert abs(float(lg) - float(le)) < 1e-5
    # bf16 run-to-run noise (order of the fp64 statistics atomics -> a storage rounding or LeakyReLU kink flips, amplified by
    # the 4^3 bottleneck of this small volume) is measured on two eager runs; the replayed graph must sit inside it
    scale = float(ge.abs().max())
    noise = float((ge - g0).abs().max())
    err = float((gg - ge).abs().max())
    print("graph vs eager %.3e, eager vs eager %.3e, scale %.3e" % (err, noise, scale))
    assert err <= max(3.0 * noise, 2e-3 * scale) and err <= 1e-1 * scale, (err, noise, scale)
    # running statistics advanced once per call in both modes (no double counting inside the graph)
    del graph


def test_metrics_on_device_match_numpy():
    """N2: the device measures equal the oracle's restatement of medpy (oracle/measures.py)."""
    from stroke_prediction_amd.common import metrics as M
    from oracle import measures as OM
    g = torch.Generator().manual_seed(5)
    res = torch.rand(2, 1, 20, 24, 28, generator=g)
    tgt = (torch.rand(2, 1, 20, 24, 28, generator=g) > 0.6).float()
    ref = OM.binary_measures(res.numpy(), tgt.numpy())
    got = M.binary_measures_torch(res.to(DEV), tgt.to(DEV), True, distances=True)
    fast = M.binary_measures_torch(res.to(DEV), tgt.to(DEV), True, distances=False)
    for k in ("dc", "precision", "sensitivity", "specificity"):
        assert abs(ref[k] - getattr(got, k)) < 1e-12, k
    for k in ("hd", "assd"):
        assert abs(ref[k] - getattr(got, k)) <= 1e-5 * max(1.0, ref[k]), k
    assert fast.dc == ref["dc"] and fast.hd == np.inf


def test_full_size_directional_derivative():
    """BASELINE.json configs[1] volume size (2 x 128^3 -> 2 x 88^3), where the CPU oracle is too slow to run in a test:
    a size-independent property instead.  For a random direction d in parameter space the analytic gradient must
    reproduce the central finite difference of the loss, <grad L, d> ~= (L(p + e d) - L(p - e d)) / 2e -- this checks
    every backward kernel against the forward kernels at the full tile / grid configuration of the headline workload
    (parity mode: the difference quotient needs fp32 forward accuracy).  Also: the bf16 fast path agrees with the
    parity path on loss and outputs at this size."""
    seed = 21
    torch.manual_seed(seed)
    x, y = W.unet_inputs(1, (128, 128, 128), seed)
    xd, yd = x.to(DEV), y.to(DEV)
    model = build(seed, "f32")
    model.train()

    def loss_of(m):
        dto = m(UnetDtoUtil.init_dto(xd))
        seg = torch.cat((dto.outputs.core, dto.outputs.penu), 1)
        return nets.unet_loss(seg, yd), seg

    loss, seg32 = loss_of(model)
    loss.backward()
    params = [p for p in model.parameters()]
    g = torch.Generator().manual_seed(seed)
    dirs = [torch.randn(p.shape, generator=g).to(DEV) * p.detach().abs().mean() for p in params]
    analytic = float(sum((p.grad.double() * d.double()).sum() for p, d in zip(params, dirs)))
    eps = 2e-3
    vals = []
    with torch.no_grad():
        for sgn in (+1.0, -1.0):
            for p, d in zip(params, dirs):
                p.add_(sgn * eps * d)
            model._ensure_flat()
            # (no manual cache invalidation: in-place edits bump the parameters' version counters, runtime/flat.py)
            vals.append(float(loss_of(model)[0].detach()))
            for p, d in zip(params, dirs):
                p.add_(-sgn * eps * d)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(analytic - numeric) <= 0.03 * abs(numeric) + 1e-6, (analytic, numeric)
    # fast path at the same size: same loss to bf16 accuracy, outputs within 2e-2
    fast = build(seed, "bf16")
    fast.train()
    lf, segf = loss_of(fast)
    assert abs(float(lf.detach()) - float(loss.detach())) < 5e-3
    assert float((segf - seg32).abs().max()) < 3e-2
