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
    def __init__(self, model, process_group=None, mode="fast", optimizer=None, bucketed=True):
        """optimizer: optional -- a ``step`` pre-hook is registered that exchanges whatever part of the gradient has not
        been exchanged yet (models whose backward is several autograd nodes, the CAE, finish their exchange there at
        the latest).  bucketed=False: ONE blocking all-reduce at the end of backward (round-1 behaviour)."""
        assert mode in ("fast", "exact")
        self.model = model
        self.group = process_group
        self.mode = mode
        self.bucketed = bucketed
        self._works = []
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.nbuckets_last = 0
        # SP_FORCE_SYNC: install the exchange even on a 1-rank group (rehearses RCCL + hipGraph capture on one GPU)
        if self.world > 1 or (dist.is_initialized() and os.environ.get("SP_FORCE_SYNC")):
            model.grad_sync = self._sync
            model.grad_bucket_ready = self._bucket if bucketed else None
            self.broadcast_parameters()
            if mode == "exact":
                from stroke_prediction_amd.runtime import layers
                layers.SYNC.update(group=process_group, world=self.world, on=True)
            if optimizer is not None:
                optimizer.register_step_pre_hook(lambda *a, **k: self.sync())

    def close(self):
        from stroke_prediction_amd.runtime import layers
        layers.SYNC.update(group=None, world=1, on=False)
        self.model.grad_sync = None
        self.model.grad_bucket_ready = None

    @property
    def grad_scale(self):
        """pass to FusedAdam(grad_scale=...).  fast: the all-reduce SUMS local-mean gradients, the optimiser divides;
        exact: the summed local gradients ARE the global gradient."""
        return 1.0 if self.mode == "exact" else 1.0 / self.world

    def sync(self):
        """exchange what has not been exchanged yet (no-op when backward already did)"""
        if self.model.grad_sync is not None:
            self.model._after_backward()

    def broadcast_parameters(self, src=0):
        flat, _ = self.model.flat_buffers()
        dist.broadcast(flat, src, group=self.group)
        for _, b in self.model.named_buffers():
            dist.broadcast(b, src, group=self.group)

    # ---- the exchange.  RCCL (backend "nccl"): an async all_reduce is enqueued on the process group's own stream
    # behind an event of the current stream, so a bucket's ring runs over xGMI while the compute stream goes on with
    # the remaining data / weight gradients; ``Work.wait()`` makes the compute stream (Adam) wait for it.
    def _bucket(self, flat_grad, lo, hi):
        self._works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _sync(self, flat_grad, lo=0, hi=None):
        hi = flat_grad.numel() if hi is None else hi
        self.nbuckets_last = len(self._works) + (1 if hi > lo else 0)
        if hi > lo:
            dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        for w in self._works:
            w.wait()
        self._works = []
