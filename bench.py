#!/usr/bin/env python3
"""Headline benchmark: train-step voxels/sec of the 3-D U-Net (BASELINE.json configs[1]):
channels 2 16 32 64 32 16 32 2, batch 4 per GPU, 2x128^3 synthetic volumes, bf16 storage / MFMA.

A step = forward + (Dice+Dice)/2 + zero_grad + backward + [RCCL all-reduce] + Adam, i.e.
``Learner.train_batch`` without the CPU-side medpy metrics (excluded on both sides).
Prints ONE JSON line on rank 0 (contract in the task statement) with ``roofline`` (dominant kernel,
HIP-event timed inside the timed region) and ``cpu_baseline`` (the CPU oracle on the host cores).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHANNELS = [2, 16, 32, 64, 32, 16, 32, 2]
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 2500.0 / 3.0}     # dense bf16 MFMA (MI355X_MICROARCH.md); f32 mode = 3 MFMAs/product
TRAIN_GFLOP_PER_SAMPLE_128 = 345.7                        # SURVEY.md 8d (fwd + dgrad + wgrad)


def cpu_baseline(size, steps=1, batch=2):
    """The CPU oracle (fp32 restatement of the reference path) timed on the host cores: 1 warm-up + `steps`."""
    from oracle import nets, weights as W
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # more threads than this only adds contention on this path
    sd = W.make_state_dict(W.unet_spec(CHANNELS), 1234)
    names = nets.trainable(sd)
    for k in names:
        sd[k].requires_grad_(True)
    x, y = W.unet_inputs(batch, size, 1234)
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    times = []
    for step in range(steps + 1):
        t0 = time.perf_counter()
        seg = nets.unet_forward(sd, x, training=True)
        loss = nets.unet_loss(seg, y)
        grads = torch.autograd.grad(loss, [sd[k] for k in names])
        with torch.no_grad():
            nets.adam_step([sd[k] for k in names], grads, m, v, step + 1, lr=1e-3, betas=(0.99, 0.999), weight_decay=1e-5)
        times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]
    vox = batch * size[0] * size[1] * size[2]
    return {"value": vox / t, "unit": "voxels/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "CPU oracle (fp32), batch %d x 2x%d^3, 1 warm-up + %d timed steps, median %.2f s/step"
                      % (batch, size[0], steps, t)}


def torch_gpu_baseline(size, batch, steps=3, bf16=True):
    """Optional extra leg (--torch-gpu-baseline): the same torch restatement run by stock PyTorch-ROCm (MIOpen
    convolutions, bf16 autocast, eager autograd, torch-style Adam) on this GPU.  A reference point for what the
    reference code base itself would reach on an MI355X; never the thing shipped or the headline value."""
    from oracle import nets, weights as W
    dev = "cuda:0"
    sd = {k: v.to(dev) for k, v in W.make_state_dict(W.unet_spec(CHANNELS), 1234).items()}
    names = nets.trainable(sd)
    for k in names:
        sd[k].requires_grad_(True)
    x, y = W.unet_inputs(batch, size, 1234)
    x, y = x.to(dev), y.to(dev)
    opt = torch.optim.Adam([sd[k] for k in names], lr=1e-3, weight_decay=1e-5, betas=(0.99, 0.999))
    times = []
    for step in range(steps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
            seg = nets.unet_forward(sd, x, training=True)
        loss = nets.unet_loss(seg.float(), y)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = sorted(times[2:])[len(times[2:]) // 2]
    vox = batch * size[0] * size[1] * size[2]
    return {"value": vox / t, "unit": "voxels/s", "ms_per_step": 1e3 * t, "kind": "torch restatement on the same GPU (MIOpen, %s, eager)"
            % ("bf16 autocast" if bf16 else "fp32"), "batch": batch}


def bench_unet_infer(args, world, rank, dev):
    """Tester.infer_batch / Learner.validate_batch -- eval-mode forward (running BatchNorm statistics),
    torch.no_grad, reference call path model(dto) (Tester.py:16-28, Learner.py:132-142)."""
    import torch.distributed as dist
    from stroke_prediction_amd.common.model.Unet3D import Unet3D
    import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil
    from stroke_prediction_amd.runtime.unet_engine import unet_out_dims
    size = (args.size,) * 3
    out = unet_out_dims(size)
    torch.manual_seed(1234)
    model = Unet3D(CHANNELS, dtype=args.dtype).to(dev).eval()
    model.freeze(True)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn((args.batch, 2) + size, generator=g, device=dev)
    labels = torch.zeros((args.batch, 2) + out, device=dev)

    def step():
        with torch.no_grad():
            dto = model(UnetDtoUtil.init_dto(images, labels[:, 0:1], labels[:, 1:2]))
        return dto.outputs.core

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3)):
        step()
    fence()
    graph, _ = capture_step(step, not args.no_graph and world == 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    vox = world * args.batch * size[0] * size[1] * size[2] * args.steps
    res = {"metric": "inference voxels/sec, 3D U-Net Bx2x%d^3 (eval forward)" % args.size, "value": vox / dt, "unit": "voxels/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "3D U-Net --channels 2 16 32 64 32 16 32 2, batch %d/GPU, 2x%d^3 -> 2x%d^3, eval forward "
                                  "(SURVEY 8 row N1)" % (args.batch, args.size, out[0]),
                      "launch": "hipGraph" if graph is not None else "eager"}}
    if rank == 0:
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


def capture_step(step, enabled):
    """Capture one optimiser step (hundreds of kernel launches) into a hipGraph; returns (graph, static_loss) or (None, None)."""
    if not enabled:
        return None, None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread may query events while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            gloss = step()
        graph.replay()
        torch.cuda.synchronize()
        return graph, gloss
    except Exception as e:      # keep the measurement alive: fall back to eager launches
        print("graph capture failed (%s): eager launches" % (str(e).splitlines()[0][:200]), file=sys.stderr)
        torch.cuda.synchronize()
        return None, None


def bench_cae(args, world, rank, dev):
    """BASELINE configs[2]: CAE --channelscae 1 16 24 32 100 800 1, batch 4, 1 x D x 128 x 128 (D = 28 native; the
    reference cannot close the loss at D = 128), one Learner.train_batch without the CPU metrics."""
    import torch.distributed as dist
    from stroke_prediction_amd.common.model.Cae3D import Cae3D, Enc3D, Dec3D
    from stroke_prediction_amd.common.metrics import BatchDiceLoss
    from stroke_prediction_amd.learner.CaeReconstructionLearner import CaeReconstructionLearner
    from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
    from oracle import weights as W          # synthetic blob labels only (input generator, not the checker)
    ch = [1, 16, 24, 32, 100, 800, 1]
    d, hw = args.cae_depth, 128
    torch.manual_seed(1234)
    cae = Cae3D(Enc3D(hw, d, ch, 5, 1.0, dtype=args.dtype), Dec3D(hw, d, ch, 5, 1.0, dtype=args.dtype)).to(dev).train()
    # N = 1: the step is replayed as one hipGraph.  N > 1: eager launches by default -- measured equal on this path
    # (the GPU, not the launching thread, is the bottleneck: 5.11 ms either way) and an RCCL collective inside a
    # capture is the one thing that cannot be rehearsed on a 1-GPU box; SP_DIST_GRAPH=1 captures it too.
    use_graph = not args.no_graph and (world == 1 or bool(os.environ.get("SP_DIST_GRAPH")))
    opt = FusedAdam([p for p in cae.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5, betas=(0.9, 0.999),
                    grad_scale=1.0 / world, capturable=use_graph)
    attach_flat_grads(cae)

    class _L:
        batch_size = args.batch
    with contextlib.redirect_stdout(sys.stderr):
        learner = CaeReconstructionLearner(_L(), None, cae, opt, None, 1, None, "/tmp/_bench_cae", BatchDiceLoss([1.0]), verbose=False)
    labels, clinical = W.cae_inputs(args.batch, d, hw, 1234 + rank)
    batch = {"case_id": list(range(args.batch)), "images": None, "labels": labels.to(dev), "clinical": clinical.to(dev)}

    def step():
        dto = learner.inference_step(batch)
        loss = learner.loss_step(dto, 30)
        opt.zero_grad()
        loss.backward()
        if world > 1:
            dist.all_reduce(cae.flat_buffers()[1])
        opt.step()
        return loss

    for _ in range(max(args.warmup, 3 if use_graph else 0)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    graph, gloss = capture_step(step, use_graph)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
            loss = gloss
        else:
            loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vox = world * args.batch * d * hw * hw * args.steps
    res = {"metric": "train-step voxels/sec, CAE Bx1xDx128x128 (3 encoder + 4 decoder passes)", "value": vox / dt,
           "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "CAE --channelscae 1 16 24 32 100 800 1, batch %d/GPU, 1x%dx128x128 (configs[2])" % (args.batch, d),
                      "global_batch": world * args.batch, "parallelism": "dp%d" % world, "loss": float(loss.detach()),
                      "launch": "hipGraph" if graph is not None else "eager"}}
    if rank == 0:
        print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--workload", default="unet", choices=["unet", "cae", "unet-infer"],
                    help="unet = BASELINE configs[1] (headline); cae = configs[2]: CAE 1 16 24 32 100 800 1, 3 enc + 4 dec passes; "
                         "unet-infer = SURVEY 8 row N1: eval-mode forward only (Tester / validate_batch), no gradients")
    ap.add_argument("--cae-depth", type=int, default=28, help="CAE volume depth (28 native, 124 = closest closed size to 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--torch-gpu-baseline", action="store_true",
                    help="also time the torch restatement (MIOpen, bf16 autocast) on this GPU; reported beside cpu_baseline")
    ap.add_argument("--dp-mode", default="fast", choices=["fast", "exact"],
                    help="multi-GPU: fast = local BatchNorm/Dice + gradient mean; exact = global-batch BatchNorm and Dice sums")
    ap.add_argument("--layers", action="store_true", help="print per-layer conv kernel timings to stderr")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
    # SP_BENCH_DEVICE / SP_BENCH_BACKEND: rehearsal of the multi-rank code path on a one-GPU box (all ranks on one
    # device, gloo instead of RCCL, which refuses two ranks per device); never set in a measured run
    if os.environ.get("SP_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["SP_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or os.environ.get("SP_FORCE_SYNC"):      # SP_FORCE_SYNC: 1-rank RCCL group, rehearses capture on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if os.environ.get("SP_BENCH_BACKEND"):
            dist.init_process_group(os.environ["SP_BENCH_BACKEND"])
        else:
            dist.init_process_group("nccl", device_id=dev)

    import stroke_prediction_amd  # noqa: F401  (puts the drop-in packages on sys.path)
    from stroke_prediction_amd.common.model.Unet3D import Unet3D
    from stroke_prediction_amd.common.metrics import BatchDiceLoss, mean_of_channel_losses
    import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil
    from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
    from stroke_prediction_amd.parallel import DataParallelSync
    from stroke_prediction_amd.runtime import ops as O
    from stroke_prediction_amd.runtime.unet_engine import unet_out_dims

    if args.workload == "cae":
        return bench_cae(args, world, rank, dev)
    if args.workload == "unet-infer":
        return bench_unet_infer(args, world, rank, dev)
    size = (args.size,) * 3
    out = unet_out_dims(size)
    torch.manual_seed(1234)                      # identical random-init weights on every rank
    model = Unet3D(CHANNELS, dtype=args.dtype).to(dev).train()
    sync = DataParallelSync(model, mode=args.dp_mode)
    # N = 1: the step is replayed as one hipGraph.  N > 1: eager launches by default -- measured equal on this path
    # (the GPU, not the launching thread, is the bottleneck: 5.11 ms either way) and an RCCL collective inside a
    # capture is the one thing that cannot be rehearsed on a 1-GPU box; SP_DIST_GRAPH=1 captures it too.
    use_graph = not args.no_graph and (world == 1 or bool(os.environ.get("SP_DIST_GRAPH")))
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5,
                    betas=(0.99, 0.999), grad_scale=sync.grad_scale, capturable=use_graph)   # train_unet_segmentation.py:13-14,32
    attach_flat_grads(model)
    sys.stdout = open(os.devnull, "w") if rank != 0 else sys.stdout
    with contextlib.redirect_stdout(sys.stderr):      # the reference's constructor prints; stdout carries the JSON line only
        crit = BatchDiceLoss([1.0])
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn((args.batch, 2) + size, generator=g, device=dev)
    labels = (torch.rand((args.batch, 2) + out, generator=g, device=dev) > 0.7).float()

    def step():
        dto = model(UnetDtoUtil.init_dto(images, labels[:, 0:1], labels[:, 1:2]))
        # UnetSegmentationLearner.loss_step: (Dice(core) + Dice(penu)) / 2
        loss = mean_of_channel_losses(crit, (dto.outputs.core, dto.outputs.penu), (dto.given_variables.core, dto.given_variables.penu))
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3 if use_graph else 0)):
        step()
    fence()
    # the whole optimiser step (~330 launches) as ONE hipGraph: same kernels, same order, no Python between them
    graph, gloss = capture_step(step, use_graph)
    launch_mode = "hipGraph" if graph is not None else "eager"
    fence()
    prof = None
    if not args.no_kernel_timing:
        # per-kernel HIP-event timing needs eager launches: a separate pass over the same steps, not the timed region
        O.PROFILE = []
        for _ in range(min(args.steps, 3)):
            step()
        fence()
        prof, O.PROFILE = O.PROFILE, None
        prof_steps = min(args.steps, 3)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
            loss = gloss
        else:
            loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    vox = world * args.batch * size[0] * size[1] * size[2] * args.steps
    res = {
        "metric": "train-step voxels/sec, 3D U-Net Bx2x128^3", "value": vox / dt, "unit": "voxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "3D U-Net --channels 2 16 32 64 32 16 32 2, batch %d/GPU, 2x%d^3 -> 2x%d^3, "
                               "fwd+Dice+bwd+Adam (configs[1])" % (args.batch, args.size, out[0]),
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world, "loss": float(loss.detach()),
                   "launch": launch_mode, "dp_mode": args.dp_mode if world > 1 else None},
    }
    # ---- roofline of the dominant kernel, from HIP events recorded around every launch of the timed region
    if prof:
        agg = {}
        per_layer = {}
        for tag, flops, e0, e1, detail in prof:
            a = agg.setdefault(tag, [0.0, 0.0, 0])
            ms = e0.elapsed_time(e1)
            a[0] += ms * 1e-3
            a[1] += flops
            a[2] += 1
            pl = per_layer.setdefault((tag, detail), [0.0, 0.0, 0])
            pl[0] += ms * 1e-3; pl[1] += flops; pl[2] += 1
        if args.layers:
            for (tag, detail), (tt, ff, nn) in sorted(per_layer.items(), key=lambda kv: -kv[1][0]):
                print("  %-11s %-34s %7.1f us/launch  %6.1f TFLOP/s  x%d/step" % (tag, detail, 1e6 * tt / nn, ff / tt / 1e12, nn // prof_steps),
                      file=sys.stderr)
        dom = max(agg, key=lambda k: agg[k][0])
        t, fl, n = agg[dom]
        dt_prof = dt / args.steps * prof_steps
        peak = PEAK_TFLOPS[args.dtype]
        traffic = None
        try:    # HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE)
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f)[dom]["bytes_per_launch"]
        except Exception:
            pass
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": fl / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                           "frac": fl / t / 1e12 / peak, "traffic": traffic, "launches": n,
                           "avg_launch_us": 1e6 * t / n, "share_of_step": t / dt_prof,
                           "timing": "HIP events around every launch, %d eager steps right before the timed region" % prof_steps}
        if traffic:     # the same launches against the HBM roof (PMC bytes per launch / measured launch time; peak 8 TB/s)
            gbps = traffic / (t / n) / 1e9
            res["roofline"]["hbm"] = {"achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0}
        res["kernels"] = {k: {"time_s_per_step": v[0] / prof_steps, "tflops": v[1] / v[0] / 1e12, "launches_per_step": v[2] / prof_steps}
                          for k, v in agg.items()}
        res["train_step_tflops"] = TRAIN_GFLOP_PER_SAMPLE_128 * (args.size / 128.0) ** 3 * 1e9 * world * args.batch * args.steps / dt / 1e12 \
            if args.size == 128 else None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(size)
    if rank == 0 and world == 1 and args.torch_gpu_baseline and args.workload == "unet":
        try:
            res["torch_gpu_baseline"] = torch_gpu_baseline(size, args.batch, bf16=(args.dtype == "bf16"))
        except Exception as e:
            res["torch_gpu_baseline"] = {"error": str(e).splitlines()[0][:200]}
    if rank == 0:
        print(json.dumps(res))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
