"""Template-method training loop (API of the reference ``learner/Learner.py:16-226``).

Hooks kept by name: ``inference_step``, ``loss_step``, ``batch_metrics_step``, ``train_batch``,
``validate_batch``, ``adapt_lr``, ``adapt_betas``, ``print_epoch``, ``plot_epoch``, ``visualize_epoch``,
``run_training``, ``path``, ``save_model`` / ``load_model``, ``save_training`` / ``load_training``.
The three lines that matter for speed -- ``zero_grad`` / ``backward`` / ``step`` (Learner.py:120-122) -- drive
the HIP path: backward is the model's fused autograd node, the step is ``FusedAdam`` when the caller
passes one (any ``torch.optim`` optimizer still works).  torch-0.3 idioms of the reference
(``Variable``, ``.numpy()[0]``, ``numpy.Inf``) are replaced by their current equivalents.
"""
import json
import math

import os

import torch

import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common.dto.Dto import Dto
from common.dto.MetricMeasuresDto import MetricMeasuresDto, BinaryMeasuresDto
from common.inference.Inference import Inference


# The reference writes the metric history with jsonpickle 0.9.6 (Learner.py:103,110; requirements.txt:9): objects become
# {"py/object": "<module>.<Class>", <attributes>}, non-finite floats the JSON extensions Infinity / NaN.  jsonpickle is
# not a dependency here; the two functions below write and read that layout for the three DTO classes involved, so a
# training interrupted under the reference resumes here and vice versa.
_DTO_PATH = "common.dto.MetricMeasuresDto."


def _encode_metrics(history):
    def enc(v):
        if isinstance(v, Dto):
            d = {"py/object": _DTO_PATH + type(v).__name__ if isinstance(v, (MetricMeasuresDto, BinaryMeasuresDto))
                 else "common.dto.Dto.Dto"}
            d.update({k: enc(x) for k, x in v})
            return d
        if torch.is_tensor(v):
            return float(v)
        return v
    return json.dumps({phase: [enc(m) for m in ms] for phase, ms in history.items()})


def _decode_metrics(text):
    seen = []

    def dec(v):
        if isinstance(v, dict) and "py/id" in v:            # jsonpickle back-reference to the n-th object of the document
            return seen[v["py/id"] - 1]
        if isinstance(v, dict) and ("py/object" in v or "__dto__" in v):      # (__dto__: this package's round-1 files)
            kind = (v.get("py/object") or v.get("__dto__")).rsplit(".", 1)[-1]
            slot = len(seen)
            seen.append(None)
            vals = {k: dec(x) for k, x in v.items() if k not in ("py/object", "__dto__")}
            if any(isinstance(x, dict) and "py/id" in x and seen[x["py/id"] - 1] is None for x in v.values()):
                raise ValueError("metric history: back-reference to an object that is still being decoded")
            if kind == "BinaryMeasuresDto":
                obj = BinaryMeasuresDto(*(vals.get(k) for k in ("dc", "hd", "assd", "precision", "sensitivity", "specificity")))
            elif kind == "MetricMeasuresDto":
                obj = MetricMeasuresDto(*(vals.get(k) for k in ("loss", "core", "penu", "lesion")))
            else:
                obj = Dto(**vals)
            seen[slot] = obj
            return obj
        if isinstance(v, list):
            out = []
            seen.append(out)            # jsonpickle 0.9.6 numbers lists too (its _mkref counts every container)
            out.extend(dec(x) for x in v)
            return out
        return float("inf") if v == "inf" else v
    return {phase: dec(ms) for phase, ms in json.loads(text).items()}


class Learner(Inference):
    FNB_MODEL = 'model'
    FNB_OPTIM = 'optimizer'
    FNB_TRAIN = 'training'
    FNB_PLOTS = 'plots'
    FNB_IMAGE = 'visual'
    FNB_MARKS = '_learner'
    EXT_MODEL = '.model'
    EXT_OPTIM = '.optim'
    EXT_TRAIN = '.json'
    EXT_IMAGE = '.png'

    GRAPH_WARMUP = 3      # eager steps before a (batch shape, graph_key) is captured: allocations and caches settle

    def __init__(self, dataloader_training, dataloader_validation, model, optimizer, scheduler, n_epochs: int,
                 path_previous_base: str = None, path_outputs_base: str = '/tmp/stroke-prediction',
                 graph: bool = False, batch_metrics: bool = True, sync_loss: bool = True, distance_metrics_every: int = 16):
        """graph / batch_metrics / sync_loss / distance_metrics_every are additions to the reference signature (Learner.py:33-35).
        distance_metrics_every = k: a TRAINING batch's metrics (Learner.py:124) are the on-device confusion counts (Dice, precision,
        sensitivity, specificity: 0.04 ms); the surface distances (Hausdorff / ASSD: 2.2 ms of kernels per call, 70 % of a 3 ms step)
        are measured on every k-th training batch and held in between, so the epoch means are means over the sampled batches;
        k = 1 is the reference's every-batch behaviour, 0 never measures them (they stay inf); validation batches always do.  graph=True: ``train_batch`` replays forward + loss + zero_grad + backward + step as
        ONE hipGraph per (batch shapes, ``graph_key(epoch)``) -- the batch is copied into static device buffers, the
        optimiser's lr / betas are read from device scalars (``FusedAdam(capturable=True)``), so ``adapt_lr`` /
        ``adapt_betas`` keep working under replay.  batch_metrics=False skips ``batch_metrics_step`` (the reference's
        per-batch medpy metrics, Learner.py:124,136; reported as zeros).  sync_loss=False keeps the batch loss on the device
        (no host round trip per step); the epoch mean is converted once."""
        Inference.__init__(self, model)
        self._graph_enabled = bool(graph)
        self._batch_metrics = bool(batch_metrics)
        self._sync_loss = bool(sync_loss)
        self._distance_every = int(distance_metrics_every)
        self._train_batches = 0
        self._held_distances = {}
        self._graphs = {}
        assert dataloader_training.batch_size > 1, 'For normalization layers batch_size > 1 is required.'
        self._dataloader_training = dataloader_training
        self._dataloader_validation = dataloader_validation
        self._optimizer = optimizer
        self._scheduler = scheduler
        self._n_epochs = n_epochs
        self._path_outputs_base = path_outputs_base
        self._path_previous_base = path_previous_base
        if path_previous_base is None:
            self._metric_dtos = {'training': [], 'validate': []}
        else:
            self.load_model(self.is_cuda)
            self.load_training()
            print('Continue training', path_previous_base, '...')
        assert len(self._metric_dtos['training']) == len(self._metric_dtos['validate']), 'Incomplete training data!'

    # ------------------------------------------------------------------ file naming (Learner.py:59-78)
    def path(self, mode: str, type: str, suffix: str = ''):
        base = {'load': self._path_previous_base, 'save': self._path_outputs_base}.get(mode)
        ext = {self.FNB_MODEL: self.EXT_MODEL, self.FNB_OPTIM: self.EXT_OPTIM, self.FNB_TRAIN: self.EXT_TRAIN,
               self.FNB_PLOTS: self.EXT_IMAGE, self.FNB_IMAGE: self.EXT_IMAGE}.get(type)
        if base is None or ext is None:
            return None
        return base + self.FNB_MARKS + suffix + ext

    # ------------------------------------------------------------------ hooks
    def loss_step(self, dto: Dto, epoch):
        raise NotImplementedError

    def get_start_epoch(self):
        return 0

    def get_start_min_loss(self):
        return float('inf')

    def batch_metrics_step(self, dto: Dto, epoch) -> MetricMeasuresDto:
        return MetricMeasuresDtoInit.init_dto()

    def print_epoch(self, epoch, phase, epoch_metrics: MetricMeasuresDto):
        pass

    def plot_epoch(self, plotter, epochs):
        pass

    def visualize_epoch(self, epoch):
        pass

    def adapt_lr(self, epoch):
        if self._scheduler is not None:
            self._scheduler.step()

    def adapt_betas(self, epoch):
        pass

    # ------------------------------------------------------------------ checkpoints (Learner.py:90-114)
    def load_model(self, cuda=True):
        model = torch.load(self.path('load', self.FNB_MODEL), weights_only=False)
        self._model = model.cuda() if cuda else model

    def load_training(self):
        path_training = self.path('load', self.FNB_TRAIN)
        path_optimizer = self.path('load', self.FNB_OPTIM)
        print('Loading:', path_training, path_optimizer)
        self._optimizer.load_state_dict(torch.load(path_optimizer, weights_only=False))
        with open(path_training, 'r') as fp:
            self._metric_dtos = _decode_metrics(fp.read())

    @staticmethod
    def _is_rank0():
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def save_training(self):
        if not self._is_rank0():             # data-parallel replicas are identical: one writer
            return
        torch.save(self._optimizer.state_dict(), self.path('save', self.FNB_OPTIM))
        with open(self.path('save', self.FNB_TRAIN), 'w') as fp:
            fp.write(_encode_metrics(self._metric_dtos))

    def save_model(self, suffix=''):
        """Learner.py:112-114 writes ``model.cpu()`` and moves the model back.  Here the LIVE model is never moved: a CPU
        copy is pickled.  Moving it would re-create the parameter storages, invalidate the captured step (hipGraph) on rank 0
        only, and rank 0 would then re-warm with eager steps whose bucketed gradient exchange issues other collectives than
        the graph replays of the other ranks (a hang in any multi-rank run_training, which saves after epoch 0)."""
        if not self._is_rank0():
            return
        import copy
        torch.save(copy.deepcopy(self._model).cpu(), self.path('save', self.FNB_MODEL, suffix))

    # ------------------------------------------------------------------ the hot three lines
    def _optimise(self, batch: dict, epoch):
        """Learner.py:117-122: forward, loss, zero_grad / backward / step."""
        dto = self.inference_step(batch)
        loss = self.loss_step(dto, epoch)

        self._optimizer.zero_grad()
        loss.backward(self._root_grad(loss))
        self._optimizer.step()
        return dto, loss

    _ROOT_ONES = {}

    @classmethod
    def _root_grad(cls, loss):
        """the ``1`` that ``loss.backward()`` would create with a fill launch per call (5 us on the step's dependent chain): one cached
        tensor per device and dtype (autograd reads it, never writes it); non-scalar losses keep autograd's own behaviour (an error)"""
        if loss.dim() != 0 or not loss.is_cuda:
            return None
        key = (loss.device, loss.dtype)
        one = cls._ROOT_ONES.get(key)
        if one is None:
            one = cls._ROOT_ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
        return one

    def graph_key(self, epoch):
        """whatever ``loss_step`` bakes into the captured step besides tensors (an epoch-dependent Python constant):
        a new value means a new capture"""
        return None

    def static_batch(self, batch: dict, epoch=0) -> dict:
        """graph mode: the batch with its tensors replaced by the captured step's own input buffers (filled from ``batch``).
        An input pipeline that writes every batch straight into these tensors (``t.copy_(host_batch)``) and passes the
        dict to ``train_batch`` saves the device-to-device copy of each step; any other batch of the same shapes is
        copied in as before."""
        if not self._graph_enabled:
            return batch
        dev = next(self._model.parameters()).device
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        key = (tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tensors.items())), self.graph_key(epoch))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 4:
                self._graphs.clear()
            g = self._graphs[key] = dict(static={k: torch.empty(v.shape, dtype=v.dtype, device=dev) for k, v in tensors.items()},
                                         warm=0, graph=None, dto=None, loss=None)
        out = dict(batch)
        for k, v in tensors.items():
            g["static"][k].copy_(v, non_blocking=True)
            out[k] = g["static"][k]
        return out

    def _optimise_graph(self, batch: dict, epoch):
        if not getattr(self._optimizer, "capturable", False):
            raise RuntimeError("Learner(graph=True) needs an optimiser whose step can be captured and whose hyper-parameters "
                               "live on the device: stroke_prediction_amd.optim.FusedAdam(..., capturable=True)")
        dev = next(self._model.parameters()).device
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        key = (tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tensors.items())), self.graph_key(epoch))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 4:
                self._graphs.clear()
            g = self._graphs[key] = dict(static={k: torch.empty(v.shape, dtype=v.dtype, device=dev) for k, v in tensors.items()},
                                         warm=0, graph=None, dto=None, loss=None)
        for k, v in tensors.items():
            if v.data_ptr() != g["static"][k].data_ptr():      # (a batch from static_batch() IS the static buffers: nothing to copy)
                g["static"][k].copy_(v, non_blocking=True)
        sbatch = dict(batch)
        sbatch.update(g["static"])
        # Data-parallel replicas (parallel.DataParallelSync installed model.grad_sync): the gradient all-reduce stays OUTSIDE
        # the capture -- forward + loss + backward are one graph, then the exchange of the flat gradient buffer (one RCCL
        # call, 1.4 MB: latency-bound either way) and the fused Adam launch run eagerly.  SP_DIST_GRAPH=1 captures the
        # collectives too (works with a one-rank communicator; not rehearsable with several ranks on a one-GPU box).
        sync_fn = getattr(self._model, "grad_sync", None)
        owner = getattr(sync_fn, "__self__", None)                   # the parallel.DataParallelSync that installed the exchange
        direct = getattr(owner, "direct", None) is not None        # ... through a communicator of our own (capturable)
        # how the gradient exchange meets the graph:
        #  * direct communicator (DataParallelSync's default from 8 MB of gradients on, or SP_DIST_GRAPH=1): the WHOLE step is one
        #    graph; the bucketed exchange sits inside it as forked branches (bucket k's all-reduce beside the backward of bucket
        #    k+1), and so do the BatchNorm / Dice sum exchanges of the exact mode;
        #  * torch.distributed exchange (the 1.4 MB U-Net): forward + loss + backward are one graph, then ONE all-reduce of the
        #    flat buffer and the fused Adam launch follow eagerly; the exact mode (collectives inside forward and backward) is
        #    then not captured at all.
        capture_all = sync_fn is None or direct or bool(os.environ.get("SP_DIST_GRAPH"))
        split = not capture_all
        if sync_fn is not None and not direct:
            from stroke_prediction_amd.runtime import layers as _layers
            if _layers.SYNC["on"]:                # exact mode over torch.distributed: collectives inside forward and backward
                return self._optimise(batch, epoch)
        if g["graph"] is None:
            if g["warm"] < self.GRAPH_WARMUP:
                g["warm"] += 1
                return self._optimise(sbatch, epoch)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if split:
                bucket_fn = self._model.grad_bucket_ready
                self._model.grad_sync = self._model.grad_bucket_ready = None
            try:
                # thread_local: an RCCL watchdog thread may query events while this thread captures
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    if split:
                        g["dto"] = self.inference_step(sbatch)
                        g["loss"] = self.loss_step(g["dto"], epoch)
                        self._optimizer.zero_grad()
                        g["loss"].backward(self._root_grad(g["loss"]))
                    else:
                        g["dto"], g["loss"] = self._optimise(sbatch, epoch)
            finally:
                if split:
                    self._model.grad_sync, self._model.grad_bucket_ready = sync_fn, bucket_fn
            g["graph"], g["split"] = graph, split
        if hasattr(self._optimizer, "push_hyper"):
            self._optimizer.push_hyper()          # lr / betas as the schedulers left them
        g["graph"].replay()
        if g.get("split"):
            _, flat_grad = self._model.flat_buffers()
            sync_fn(flat_grad, 0, flat_grad.numel())
            root = self._model._flat_root() if hasattr(self._model, "_flat_root") else self._model
            root._grads_synced = True             # an optimiser pre-hook (DataParallelSync(optimizer=...)) must not exchange again
            self._optimizer.step()
        return g["dto"], g["loss"]

    def train_batch(self, batch: dict, epoch) -> MetricMeasuresDto:
        dto, loss = self._optimise_graph(batch, epoch) if self._graph_enabled else self._optimise(batch, epoch)

        batch_metrics = self._train_metrics(dto, epoch) if self._batch_metrics else MetricMeasuresDtoInit.init_dto(*([0.0] * 13))
        batch_metrics.loss = float(loss.detach()) if self._sync_loss else loss.detach().clone()
        return batch_metrics

    def _train_metrics(self, dto, epoch):
        """``batch_metrics_step`` with the surface distances on every ``distance_metrics_every``-th call only (held in between)"""
        from common import metrics
        k = self._distance_every
        want = k > 0 and self._train_batches % k == 0
        self._train_batches += 1
        keep = metrics.DISTANCE_METRICS
        metrics.DISTANCE_METRICS = bool(keep and want)
        try:
            bm = self.batch_metrics_step(dto, epoch)
        finally:
            metrics.DISTANCE_METRICS = keep
        for name in ("core", "penu", "lesion"):
            m = getattr(bm, name, None)
            if not isinstance(m, BinaryMeasuresDto) or m.dc is None:
                continue
            if want:
                self._held_distances[name] = (m.hd, m.assd)
            elif name in self._held_distances:
                m.hd, m.assd = self._held_distances[name]
        return bm

    def validate_batch(self, batch: dict, epoch) -> MetricMeasuresDto:
        with torch.no_grad():
            dto = self.inference_step(batch)
            loss = self.loss_step(dto, epoch)
        batch_metrics = self.batch_metrics_step(dto, epoch) if self._batch_metrics else MetricMeasuresDtoInit.init_dto(*([0.0] * 13))
        batch_metrics.loss = float(loss.detach())
        return batch_metrics

    def _run_phase(self, loader, step_fn, epoch):
        acc = MetricMeasuresDtoInit.init_dto()
        for batch in loader:
            acc.add(step_fn(batch, epoch))
        acc.div(len(loader))
        if torch.is_tensor(acc.loss):        # sync_loss=False: one host round trip per epoch
            acc.loss = float(acc.loss)
        return acc

    def run_training(self):
        min_loss = self.get_start_min_loss()
        epoch = self.get_start_epoch() - 1
        for epoch in range(self.get_start_epoch(), self._n_epochs):
            self.adapt_lr(epoch)
            self.adapt_betas(epoch)

            self._model.train()
            metrics = self._run_phase(self._dataloader_training, self.train_batch, epoch)
            self.print_epoch(epoch, 'training', metrics)
            self._metric_dtos['training'].append(metrics)

            self._model.eval()
            if self._dataloader_validation is None:
                metrics = MetricMeasuresDtoInit.init_dto(*([0.0] * 13))
            else:
                metrics = self._run_phase(self._dataloader_validation, self.validate_batch, epoch)
            self.print_epoch(epoch, 'validate', metrics)
            self._metric_dtos['validate'].append(metrics)

            last = self._metric_dtos['validate'][-1].loss
            if last is not None and last < min_loss:
                min_loss = last
                self.save_model()
                self.save_training()        # allows to continue an interrupted training
                print('(New optimum: Training saved)', end=' ')
                self.visualize_epoch(epoch)
            if epoch % 50 == 0:
                self.visualize_epoch(epoch)

            if epoch > 0:
                import matplotlib
                matplotlib.use('Agg')
                import matplotlib.pyplot as plt
                fig, plot = plt.subplots()
                self.plot_epoch(plot, range(1, epoch + 2))
                fig.savefig(self.path('save', self.FNB_PLOTS), bbox_inches='tight', dpi=150)
                plt.close(fig)

        self.save_model('_final')
        self.visualize_epoch(epoch)
