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
import sys

import torch
import torch.distributed as dist


class DirectComm:
    """An RCCL communicator of this process created through the C ABI (``sp_comm_*`` / ``sp_allreduce_flat``,
    include/stroke_amd.h) instead of torch.distributed's: the 128-byte unique id travels over the existing process group
    (any backend), the collectives run on a stream of ours -- forked from the launching stream and joined before the
    optimiser -- so they can sit inside a captured step as a parallel graph branch.  ``two_shot``: reduce-scatter +
    all-gather over a padded copy-free view (every xGMI link carries 1/world of the buffer at once) instead of one
    all-reduce; an element count that is no multiple of the rank count sends its short tail through a plain all-reduce
    (``two_shot_plan``).  UNMEASURED on more than one GPU (this build's boxes have one): OPT-IN, ``DataParallelSync(direct=True)`` or
    SP_DIST_DIRECT=1, two-shot on top with SP_DIST_TWO_SHOT=1."""

    def __init__(self, group=None, device=None):
        import ctypes as C
        from stroke_prediction_amd.runtime import lib as L
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            L.call("sp_comm_unique_id", uid)
        if self.world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = C.create_string_buffer(box[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            L.call("sp_comm_init_rank", C.byref(self._comm), self.world, uid, self.rank)
        self.stream = torch.cuda.Stream(device=self.device)
        self._L = L
        self.two_shot = os.environ.get("SP_DIST_TWO_SHOT", "").strip().lower() in ("1", "true", "yes", "on")

    @staticmethod
    def two_shot_plan(n, world):
        """(chunk, tail): elements [0, world * chunk) go through reduce-scatter + all-gather (rank r owns [r * chunk, (r + 1) * chunk)),
        the last `tail` = n - world * chunk < world elements through a plain all-reduce; chunk == 0: all of it does"""
        chunk = n // world
        return chunk, n - chunk * world

    def all_reduce_async(self, t, two_shot=None):
        """sum of the fp32 tensor t over the ranks, in place, on the communicator's stream (ordered after everything enqueued
        on the current stream so far); ``wait`` makes the current stream wait for it"""
        assert t.dtype in (torch.float32, torch.float64) and t.is_contiguous() and t.is_cuda
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        n = t.numel()
        two_shot = self.two_shot if two_shot is None else two_shot
        chunk, tail = self.two_shot_plan(n, self.world)
        if t.dtype == torch.float64:       # the accumulators of the exact mode (BatchNorm / Dice sums)
            self._L.call("sp_allreduce_flat_f64", self._comm, t.data_ptr(), n, self.stream.cuda_stream)
        elif two_shot and chunk > 0:       # (a single rank runs it too: the rehearsal of the call sequence on a one-GPU box)
            self._L.call("sp_reduce_scatter_flat", self._comm, t.data_ptr(), chunk, self.rank, self.stream.cuda_stream)
            self._L.call("sp_allgather_flat", self._comm, t.data_ptr(), chunk, self.rank, self.stream.cuda_stream)
            if tail:
                self._L.call("sp_allreduce_flat", self._comm, t.data_ptr() + 4 * chunk * self.world, tail, self.stream.cuda_stream)
        else:
            self._L.call("sp_allreduce_flat", self._comm, t.data_ptr(), n, self.stream.cuda_stream)

    def wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def close(self):
        if self._comm:
            torch.cuda.synchronize(self.device)
            self._L.call("sp_comm_destroy", self._comm)
            self._comm = None


class DataParallelSync:
    def __init__(self, model, process_group=None, mode="fast", optimizer=None, bucketed=True, direct=None):
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
        if direct is None:
            # OPT-IN (the path has never run with more than one rank): SP_DIST_DIRECT=1 or direct=True gives the exchange a
            # communicator of our own -- capturable: it then sits INSIDE a Learner(graph=True) step as forked branches at the bucket
            # boundaries, bucket k travelling under the backward of bucket k+1, which pays for flat gradients of several MB (the CAE's
            # 18.9 MB, the 4-scale net's 23.4 MB).  Default: torch.distributed's all-reduce after the replayed backward graph.
            env = os.environ.get("SP_DIST_DIRECT", "").strip().lower()
            direct = env in ("1", "true", "yes", "on")
            if direct:
                print("stroke_prediction_amd.parallel: gradient exchange on a communicator of our own (SP_DIST_DIRECT) -- "
                      "unmeasured on more than one GPU", file=sys.stderr)
        # direct: the gradient exchange goes through sp_allreduce_flat on a communicator of our own (DirectComm)
        self.direct = DirectComm(process_group) if (direct and dist.is_initialized() and torch.cuda.is_available()
                                                    and dist.get_backend(process_group) != "gloo") else None
        # SP_FORCE_SYNC: install the exchange even on a 1-rank group (rehearses RCCL + hipGraph capture on one GPU)
        if self.world > 1 or (dist.is_initialized() and os.environ.get("SP_FORCE_SYNC")):
            model.grad_sync = self._sync
            model.grad_bucket_ready = self._bucket if bucketed else None
            self.broadcast_parameters()
            if mode == "exact":
                from stroke_prediction_amd.runtime import layers
                layers.SYNC.update(group=process_group, world=self.world, on=True, direct=self.direct)
            if optimizer is not None:
                optimizer.register_step_pre_hook(lambda *a, **k: self.sync())

    def close(self):
        if self.direct is not None:
            self.direct.close()
            self.direct = None
        from stroke_prediction_amd.runtime import layers
        layers.SYNC.update(group=None, world=1, on=False, direct=None)
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
        if self.direct is not None:
            self.direct.all_reduce_async(flat_grad[lo:hi])
            self._works.append(None)
            return
        self._works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _sync(self, flat_grad, lo=0, hi=None):
        hi = flat_grad.numel() if hi is None else hi
        self.nbuckets_last = len(self._works) + (1 if hi > lo else 0)
        if self.direct is not None:
            if hi > lo:
                self.direct.all_reduce_async(flat_grad[lo:hi])
            self.direct.wait()
            self._works = []
            return
        if hi > lo:
            dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        for w in self._works:
            w.wait()
        self._works = []
