"""Drop-in ``Unet3D`` (reference ``common/model/Unet3D.py:30-84``) running on hand-written gfx950 kernels.

Same constructor, ``forward(dto) -> dto``, ``freeze`` and ``state_dict`` keys
(``block{1..5}.bn_conv_relu_2x.{0,1,3,4}.*``, ``classify.{0,2}.*``) as the reference, so weights
interchange with it.  The ``torch.nn`` sub-modules below are parameter CONTAINERS only: the forward
and backward passes are one ``torch.autograd.Function`` that drives ``runtime.unet_engine`` (MFMA
implicit-GEMM convolutions with fused BatchNorm/bias/LeakyReLU, fused pool/upsample/skip kernels)
through the C ABI.  There is no CPU path: calling it without the HIP library or a GPU raises.
"""
import torch
import torch.nn as nn

from common.dto.UnetDto import UnetDto
from stroke_prediction_amd.runtime import lib as _L
from stroke_prediction_amd.runtime.flat import FlatParamsMixin


def crop(tensor_in, crop_as, dims=[]):
    """Centre crop of ``tensor_in`` to ``crop_as`` along ``dims`` (reference Unet3D.py:6-11)."""
    assert len(dims) > 0, "Specify dimensions to be cropped"
    out = tensor_in
    for d in dims:
        n = crop_as.size(d)
        out = out.narrow(d, (tensor_in.size(d) - n) // 2, n)
    return out


class Block3x3x3(nn.Module):
    """Parameter container with the reference's key layout (Unet3D.py:14-27)."""

    def __init__(self, n_input, n_channels):
        super().__init__()
        self.bn_conv_relu_2x = nn.ModuleDict({
            "0": nn.BatchNorm3d(n_input),
            "1": nn.Conv3d(n_input, n_channels, 3, stride=1, padding=0),
            "3": nn.BatchNorm3d(n_channels),
            "4": nn.Conv3d(n_channels, n_channels, 3, stride=1, padding=0),
        })

    def forward(self, *_):
        raise RuntimeError("Block3x3x3 holds parameters only; Unet3D.forward runs the fused HIP path")


class _UnetFn(torch.autograd.Function):
    """Whole-network autograd node: forward and backward are sequences of HIP kernel launches."""

    @staticmethod
    def forward(ctx, model, images, *params):
        engine = model._engine(images)
        model._begin_step()
        with _L.use(engine.variant):
            seg = engine.forward(images, model._param_dict(), model._buffer_dict(), model.training)
        ctx.model, ctx.engine = model, engine
        ctx.generation, ctx.training = engine.generation, model.training
        ctx.save_for_backward(seg)
        return seg

    @staticmethod
    def backward(ctx, dseg):
        model, engine = ctx.model, ctx.engine
        (seg,) = ctx.saved_tensors
        # the activations and BatchNorm statistics of a pass live in the engine (one set per input shape): a later
        # forward of the same shape has overwritten them
        if ctx.generation != engine.generation:
            raise RuntimeError("Unet3D.backward: another forward pass of the same input shape ran on this model after the "
                               "one being differentiated (its activations are gone); run backward before the next "
                               "forward -- gradient accumulation over micro-batches needs one backward per forward")
        if not ctx.training:
            raise RuntimeError("Unet3D.backward after an eval-mode forward is not supported: the fused backward uses the "
                               "batch-statistics BatchNorm formula; call model.train() for passes that need gradients")
        names, views, inplace = model._grad_targets()
        with _L.use(engine.variant):
            engine.backward(dseg, seg, model._param_dict(), dict(zip(names, views)), ready=model._grads_ready_from)
        model._after_backward()
        return (None, None) + tuple(None if inplace else v for v in views)


class Unet3D(FlatParamsMixin, nn.Module):
    FLAT_NBT = True      # every BatchNorm runs exactly once per forward: their step counters advance together
    N_SCALES = 3
    ENGINE_CACHE = 4     # engines kept per model (one per input shape and precision mode), least recently used evicted first

    def __init__(self, channels=[2, 32, 64, 128, 64, 32, 32, 2], channel_dim=1, channels_crop=[2, 3, 4],
                 dtype="bf16"):
        super().__init__()
        S = self.N_SCALES
        assert len(channels) == 2 * S + 2, "%s takes %d channel counts (input, %d blocks, head, classes)" % (
            type(self).__name__, 2 * S + 2, 2 * S - 1)
        self.channels = list(channels)
        self.channel_dim = channel_dim
        self.channels_crop = channels_crop
        self.compute_dtype = dtype           # "bf16" (fast) | "f32" (split-bf16 x3 MFMA, parity mode) | "fp8" (bf16 storage,
                                             # e4m3 / e5m2 MFMA operands where the fp8 kernel applies: runtime/f8.py) | "fp8b"
                                             # (the bf16 forward with the fp8 backward: gradient directions of the bf16 mode) | "f16"
                                             # (IEEE-half storage: libstroke_amd_f16.so, 3 more mantissa bits at the bf16 speed)
                                             # | "bf16x3" (forward on bf16 PAIRS, three MFMAs per product: logits within 1e-3 of
                                             # the fp32 reference; backward = the bf16 one on the hi halves)
        n_in, widths, ch_bC, n_classes = channels[0], list(channels[1:2 * S]), channels[-2], channels[-1]
        for i in range(1, S + 1):            # down path: block_i(b_{i-1} -> b_i)
            setattr(self, "block%d" % i, Block3x3x3(n_in if i == 1 else widths[i - 2], widths[i - 1]))
        for u in range(S + 1, 2 * S):        # up path: block_u(b_{u-1} + b_{2S-u} -> b_u)
            setattr(self, "block%d" % u, Block3x3x3(widths[u - 2] + widths[2 * S - u - 1], widths[u - 1]))
        self.classify = nn.ModuleDict({
            "0": nn.Conv3d(widths[-1], ch_bC, 1, stride=1, padding=0),
            "2": nn.Conv3d(ch_bC, n_classes, 1, stride=1, padding=0),
        })
        self._engines = {}

    def output_size(self, size):
        """spatial size of the segmentation for an input of spatial ``size`` (valid convolutions: 128^3 -> 88^3)"""
        from stroke_prediction_amd.runtime.unet_engine import unet_out_dims
        return unet_out_dims(tuple(size), self.N_SCALES)

    # ------------------------------------------------------------------ engine cache
    def _engine(self, images):
        from stroke_prediction_amd.runtime.unet_engine import UnetEngine
        if not images.is_cuda or not next(self.parameters()).is_cuda:
            raise RuntimeError("Unet3D (stroke_prediction_amd) runs on the MI355X HIP path only: move the model "
                               "and its inputs to the GPU (.cuda()); there is no CPU fallback")
        self._ensure_flat()
        if self.compute_dtype not in _L.DTYPE_CODES:
            raise ValueError("Unet3D: unknown precision mode %r (one of %s)" % (self.compute_dtype, sorted(_L.DTYPE_CODES)))
        dt = _L.DTYPE_CODES[self.compute_dtype]
        key = (tuple(images.shape), self.compute_dtype, images.device.index)
        eng = self._engines.get(key)
        if eng is not None:
            self._engines[key] = self._engines.pop(key)      # most recently used last
        if eng is None:
            # at most ENGINE_CACHE engines (each holds the activations of one input shape): the least recently used one goes, the
            # others -- and the hipGraphs captured over their buffers (Learner(graph=True)) -- stay valid
            while len(self._engines) >= self.ENGINE_CACHE:
                self._engines.pop(next(iter(self._engines)))
            variant = _L.VARIANT_OF[self.compute_dtype]
            with _L.use(variant):
                eng = UnetEngine(self.channels, images.shape[0], tuple(images.shape[2:]), dt, images.device,
                                 f8=(self.compute_dtype in ("fp8", "fp8b")), f8_fwd=(self.compute_dtype == "fp8"), variant=variant, hl=(self.compute_dtype in ("bf16x3", "f16x3")))
            self._engines[key] = eng
        return eng

    def forward(self, dto: UnetDto):
        images = dto.given_variables.input_modalities
        if images.dtype != torch.float32:
            images = images.float()
        params = [p for _, p in self.named_parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            segmentation = _UnetFn.apply(self, images, *params)
        else:
            eng = self._engine(images)
            with _L.use(eng.variant):
                segmentation = eng.forward(images, self._param_dict(), self._buffer_dict(), self.training)
        dto.outputs.core = segmentation[:, 0, :, :, :].unsqueeze(1)
        dto.outputs.penu = segmentation[:, 1, :, :, :].unsqueeze(1)
        return dto

    def freeze(self, freeze=False):
        requires_grad = not freeze
        for param in self.parameters():
            param.requires_grad = requires_grad

    def __setstate__(self, state):
        """Unpickling (``torch.load`` of a whole-module ``.model`` file, Learner.py:93, Tester.py:17).  A file written by the
        REFERENCE's classes resolves to this class by its import path but carries the reference's attribute set: what this
        implementation keeps beside the parameters (channel list, precision mode, engine cache) is rebuilt from the
        parameter shapes -- the state_dict keys are the same, so the weights are used as they are."""
        super().__setstate__(state)
        d = self.__dict__
        if "channels" not in d:
            nb = len([k for k in self._modules if k.startswith("block")])
            S = (nb + 1) // 2
            conv = lambda i: dict(self._modules["block%d" % i].named_parameters())["bn_conv_relu_2x.1.weight"]
            widths = [conv(i).shape[0] for i in range(1, nb + 1)]
            head = dict(self._modules["classify"].named_parameters())
            d["channels"] = [conv(1).shape[1]] + widths + [head["0.weight"].shape[0], head["2.weight"].shape[0]]
            assert S == self.N_SCALES, "pickled network has %d scales, %s has %d" % (S, type(self).__name__, self.N_SCALES)
        d.setdefault("channel_dim", 1)
        d.setdefault("channels_crop", [2, 3, 4])
        d.setdefault("compute_dtype", "bf16")
        d["_engines"] = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engines"] = {}          # engines hold device buffers and ctypes handles: rebuilt on demand
        for k in ("_flat_param", "_flat_grad", "_flat_views", "_flat_pviews", "_flat_names", "_flat_device", "_flat_nbt",
                  "grad_sync", "grad_bucket_ready", "_bucket_hi", "_grads_synced"):
            state.pop(k, None)
        return state


class LargeUnet3D(Unet3D):
    """The 4-scale network of the reference (``LargeUnet3D`` Unet3D.py:87-146: blocks 1-4 down, 5-7 up, same classify head;
    state_dict keys ``block{1..7}.bn_conv_relu_2x.*``, ``classify.{0,2}.*``).  The reference class itself cannot be
    constructed (it calls ``super(Unet3D, self).__init__()``, Unet3D.py:89); this is its topology on the same engine.
    256^3 -> 164^3."""
    N_SCALES = 4

    def __init__(self, channels=[2, 32, 64, 128, 256, 128, 64, 32, 32, 2], channel_dim=1, channels_crop=[2, 3, 4], dtype="bf16"):
        super().__init__(channels, channel_dim, channels_crop, dtype)
