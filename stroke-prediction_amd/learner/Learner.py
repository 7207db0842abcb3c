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

import torch

import common.dto.MetricMeasuresDto as MetricMeasuresDtoInit
from common.dto.Dto import Dto
from common.dto.MetricMeasuresDto import MetricMeasuresDto, BinaryMeasuresDto
from common.inference.Inference import Inference


def _encode_metrics(history):
    def enc(v):
        if isinstance(v, Dto):
            d = {k: enc(x) for k, x in v}
            d["__dto__"] = type(v).__name__
            return d
        if isinstance(v, float) and math.isinf(v):
            return "inf"
        return v
    return json.dumps({phase: [enc(m) for m in ms] for phase, ms in history.items()})


def _decode_metrics(text):
    def dec(v):
        if isinstance(v, dict) and "__dto__" in v:
            kind = v.pop("__dto__")
            vals = {k: dec(x) for k, x in v.items()}
            if kind == "BinaryMeasuresDto":
                return BinaryMeasuresDto(**vals)
            if kind == "MetricMeasuresDto":
                return MetricMeasuresDto(**vals)
            return Dto(**vals)
        return float("inf") if v == "inf" else v
    return {phase: [dec(m) for m in ms] for phase, ms in json.loads(text).items()}


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

    def __init__(self, dataloader_training, dataloader_validation, model, optimizer, scheduler, n_epochs: int,
                 path_previous_base: str = None, path_outputs_base: str = '/tmp/stroke-prediction'):
        Inference.__init__(self, model)
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

    def save_training(self):
        torch.save(self._optimizer.state_dict(), self.path('save', self.FNB_OPTIM))
        with open(self.path('save', self.FNB_TRAIN), 'w') as fp:
            fp.write(_encode_metrics(self._metric_dtos))

    def save_model(self, suffix=''):
        was_cuda = self.is_cuda
        torch.save(self._model.cpu(), self.path('save', self.FNB_MODEL, suffix))
        if was_cuda:
            self._model.cuda()

    # ------------------------------------------------------------------ the hot three lines
    def train_batch(self, batch: dict, epoch) -> MetricMeasuresDto:
        dto = self.inference_step(batch)
        loss = self.loss_step(dto, epoch)

        self._optimizer.zero_grad()
        loss.backward()
        self._optimizer.step()

        batch_metrics = self.batch_metrics_step(dto, epoch)
        batch_metrics.loss = float(loss.detach())
        return batch_metrics

    def validate_batch(self, batch: dict, epoch) -> MetricMeasuresDto:
        with torch.no_grad():
            dto = self.inference_step(batch)
            loss = self.loss_step(dto, epoch)
        batch_metrics = self.batch_metrics_step(dto, epoch)
        batch_metrics.loss = float(loss.detach())
        return batch_metrics

    def _run_phase(self, loader, step_fn, epoch):
        acc = MetricMeasuresDtoInit.init_dto()
        for batch in loader:
            acc.add(step_fn(batch, epoch))
        acc.div(len(loader))
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
