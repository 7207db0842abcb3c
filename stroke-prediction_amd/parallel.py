"""Batch data parallelism over the GPUs of one node: one process per GPU, RCCL over xGMI through
``torch.distributed`` (backend "nccl" is RCCL on ROCm).  The reference has no distributed code; the
path shards naturally by batch and needs ONE exchange per step: the sum of the flat
fp32 gradient buffer (1.42 MB for the default U-Net -> latency-bound, a single all-reduce).

mode "fast" (default): local BatchNorm statistics and local Dice sums per rank, gradients averaged --
what plain DDP would do; differs from the single-process reference at O(1/B_local).

mode "exact": reproduces the single-process reference on the GLOBAL batch.  Three extra exchanges, all tiny:
BatchNorm forward sums (sum x, sum x^2) before every normalisation, BatchNorm backward sums (sum g, sum g*x),
and the three Dice sums per output; local gradients are then partial sums of the global gradient, so the flat
all-reduce SUMS and the optimiser must NOT divide (``grad_scale`` = 1); BatchNorm gamma/beta gradients, which
every rank already holds in full, are pre-scaled by 1/world.
"""
import os

import torch
import torch.distributed as dist


class DataParallelSync:
    def __init__(self, model, process_group=None, mode="fast"):
        assert mode in ("fast", "exact")
        self.model = model
        self.group = process_group
        self.mode = mode
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # SP_FORCE_SYNC: install the exchange even on a 1-rank group (rehearses RCCL + hipGraph capture on one GPU)
        if self.world > 1 or (dist.is_initialized() and os.environ.get("SP_FORCE_SYNC")):
            model.grad_sync = self._sync
            self.broadcast_parameters()
            if mode == "exact":
                from stroke_prediction_amd.runtime import layers
                layers.SYNC.update(group=process_group, world=self.world, on=True)

    def close(self):
        from stroke_prediction_amd.runtime import layers
        layers.SYNC.update(group=None, world=1, on=False)
        self.model.grad_sync = None

    @property
    def grad_scale(self):
        """pass to FusedAdam(grad_scale=...).  fast: the all-reduce SUMS local-mean gradients, the optimiser divides;
        exact: the summed local gradients ARE the global gradient."""
        return 1.0 if self.mode == "exact" else 1.0 / self.world

    def sync(self):
        """for models whose autograd node does not call ``grad_sync`` itself (the CAE: 7 nodes per step)"""
        if self.model.grad_sync is not None:
            self._sync(self.model.flat_buffers()[1])

    def broadcast_parameters(self, src=0):
        flat, _ = self.model.flat_buffers()
        dist.broadcast(flat, src, group=self.group)
        for _, b in self.model.named_buffers():
            dist.broadcast(b, src, group=self.group)

    def _sync(self, flat_grad):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
