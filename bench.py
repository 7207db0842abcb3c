#!/usr/bin/env python3
"""Headline benchmark: train-step voxels/sec of the 3-D U-Net (BASELINE.json configs[1]):
channels 2 16 32 64 32 16 32 2, batch 4 per GPU, 2x128^3 synthetic volumes, bf16 storage / MFMA.

A step = forward + (Dice+Dice)/2 + zero_grad + backward + [RCCL all-reduce] + Adam, i.e.
``Learner.train_batch`` without the CPU-side medpy metrics (excluded on both sides).
Prints ONE JSON line on rank 0 (contract in the task statement) with ``roofline`` (dominant kernel,
HIP-event timed inside the timed region) and ``cpu_baseline`` (the CPU oracle on the host cores).
"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import stroke_prediction_amd  # noqa: F401  (puts the drop-in packages on sys.path)
    from stroke_prediction_amd.common.model.Unet3D import Unet3D
    from stroke_prediction_amd.common.metrics import BatchDiceLoss
    import stroke_prediction_amd.common.dto.UnetDto as UnetDtoUtil
    from stroke_prediction_amd.optim import FusedAdam, attach_flat_grads
    from stroke_prediction_amd.parallel import DataParallelSync
    from stroke_prediction_amd.runtime import ops as O
    from stroke_prediction_amd.runtime.unet_engine import unet_out_dims

    size = (args.size,) * 3
    out = unet_out_dims(size)
    torch.manual_seed(1234)                      # identical random-init weights on every rank
    model = Unet3D(CHANNELS, dtype=args.dtype).to(dev).train()
    sync = DataParallelSync(model)
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5,
                    betas=(0.99, 0.999), grad_scale=sync.grad_scale)   # train_unet_segmentation.py:13-14,32
    attach_flat_grads(model)
    sys.stdout = open(os.devnull, "w") if rank != 0 else sys.stdout
    crit = BatchDiceLoss([1.0])
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.randn((args.batch, 2) + size, generator=g, device=dev)
    labels = (torch.rand((args.batch, 2) + out, generator=g, device=dev) > 0.7).float()

    def step():
        dto = model(UnetDtoUtil.init_dto(images, labels[:, 0:1], labels[:, 1:2]))
        loss = (crit(dto.outputs.core, dto.given_variables.core) + crit(dto.outputs.penu, dto.given_variables.penu)) / 2
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_kernel_timing:
        O.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof, O.PROFILE = O.PROFILE, None
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
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world, "loss": float(loss)},
    }
    # ---- roofline of the dominant kernel, from HIP events recorded around every launch of the timed region
    if prof:
        agg = {}
        for tag, flops, e0, e1 in prof:
            a = agg.setdefault(tag, [0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += flops
            a[2] += 1
        dom = max(agg, key=lambda k: agg[k][0])
        t, fl, n = agg[dom]
        peak = PEAK_TFLOPS[args.dtype]
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": fl / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                           "frac": fl / t / 1e12 / peak, "traffic": None, "launches": n,
                           "avg_launch_us": 1e6 * t / n, "share_of_step": t / dt}
        res["kernels"] = {k: {"time_s_per_step": v[0] / args.steps, "tflops": v[1] / v[0] / 1e12, "launches_per_step": v[2] / args.steps}
                          for k, v in agg.items()}
        res["train_step_tflops"] = TRAIN_GFLOP_PER_SAMPLE_128 * (args.size / 128.0) ** 3 * 1e9 * world * args.batch * args.steps / dt / 1e12 \
            if args.size == 128 else None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(size)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
