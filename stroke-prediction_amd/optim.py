"""Fused Adam on flat buffers (``sp_adam_step_flat``), a drop-in for the ``torch.optim.Adam`` the
reference scripts build (train_unet_segmentation.py:32, train_shape_reconstruction.py:40): same
constructor, ``param_groups`` / ``defaults`` (``adapt_betas`` edits ``param_group['betas']``,
CaeReconstructionLearner.py:28-40), L2-coupled weight decay, bias correction, no amsgrad.

When the parameters are the views of a ``FlatParamsMixin`` model the whole step is ONE kernel over the
flat parameter / gradient / moment buffers; otherwise it falls back to one launch per tensor.
``state_dict`` keeps torch.optim.Adam's per-parameter layout (step, exp_avg, exp_avg_sq).
"""
import torch
from torch.optim.optimizer import Optimizer

from stroke_prediction_amd.runtime import ops as O


class FusedAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0,
                 capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.grad_scale = grad_scale
        self.capturable = capturable      # step count in device memory: the step can be captured in a hipGraph
        self._flat = {}

    # ------------------------------------------------------------------ flat detection
    def _flat_group(self, gi, group):
        """(p_flat, g_flat, m_flat, v_flat) if the group's params and grads tile two contiguous buffers."""
        ps = group["params"]
        if not ps or any(p.grad is None or p.dtype != torch.float32 or not p.is_cuda for p in ps):
            return None
        p0, g0 = ps[0].data_ptr(), ps[0].grad.data_ptr()
        off = 0
        for p in ps:
            if p.data_ptr() != p0 + off or p.grad.data_ptr() != g0 + off or not p.is_contiguous():
                return None
            off += p.numel() * 4
        n = off // 4
        key = (gi, p0, g0, n)
        st = self._flat.get(gi)
        if st is None or st["key"] != key:
            dev = ps[0].device
            m = torch.zeros(n, dtype=torch.float32, device=dev)
            v = torch.zeros(n, dtype=torch.float32, device=dev)
            o = 0
            for p in ps:                      # keep (or adopt) per-parameter state as views of the flat moments
                s = self.state[p]
                k = p.numel()
                if "exp_avg" in s:
                    m[o:o + k].copy_(s["exp_avg"].reshape(-1))
                    v[o:o + k].copy_(s["exp_avg_sq"].reshape(-1))
                s["exp_avg"], s["exp_avg_sq"] = m[o:o + k].view(p.shape), v[o:o + k].view(p.shape)
                s.setdefault("step", 0)
                o += k
            # flat aliases of the parameter / gradient storage (torch owns the memory)
            pf = torch.as_strided(ps[0].data, (n,), (1,))
            gf = torch.as_strided(ps[0].grad, (n,), (1,))
            # step count: carried over from the flat group this one replaces (re-flatten after .cpu()/.cuda(): the
            # device counter is the truth in capturable mode), else from the per-parameter state
            step0 = int(self.state[ps[0]].get("step", 0))
            if st is not None and self.capturable:
                step0 = max(step0, int(st["step_dev"].item()))
            st = dict(key=key, p=pf, g=gf, m=m, v=v,
                      step_dev=torch.full((1,), step0, dtype=torch.int32, device=dev),
                      lr_dev=torch.zeros(8, dtype=torch.float32, device=dev), hyper=None)
            self._flat[gi] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        O.bump_param_epoch()          # packed weight fragments cached by the conv runners are stale after this
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            st = self._flat_group(gi, group)
            if st is not None and self.capturable:
                from stroke_prediction_amd.runtime import lib as L
                # hyper-parameters live in device memory: while a hipGraph is being captured nothing is copied (a
                # captured copy would freeze today's values); ``push_hyper`` refreshes them before each replay
                if not torch.cuda.is_current_stream_capturing():
                    self._push_hyper(st, group)
                st["step_dev"].add_(1)
                L.call("sp_adam_step_flat_hyp", O.ptr(st["p"]), O.ptr(st["g"]), O.ptr(st["m"]), O.ptr(st["v"]),
                       st["p"].numel(), O.ptr(st["lr_dev"]), O.ptr(st["step_dev"]), self.grad_scale, O.stream())
                continue
            if st is not None:
                step = int(self.state[group["params"][0]]["step"]) + 1
                O.adam_step_flat(st["p"], st["g"], st["m"], st["v"], group["lr"], b1, b2, group["eps"],
                                 group["weight_decay"], step, self.grad_scale)
                for p in group["params"]:
                    self.state[p]["step"] = step
                continue
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam runs on the GPU only")
                s = self.state[p]
                if "exp_avg" not in s:
                    s["step"] = 0
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                s["step"] = int(s["step"]) + 1
                O.adam_step_flat(p.data, p.grad.contiguous(), s["exp_avg"], s["exp_avg_sq"], group["lr"], b1, b2,
                                 group["eps"], group["weight_decay"], s["step"], self.grad_scale)
        return loss

    @staticmethod
    def _push_hyper(st, group):
        hyp = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
               float(group["weight_decay"]))
        if st["hyper"] != hyp:
            st["lr_dev"].copy_(torch.tensor(hyp + (0.0,) * (st["lr_dev"].numel() - 5), dtype=torch.float32), non_blocking=False)
            st["hyper"] = hyp

    def push_hyper(self):
        """capturable mode: copy lr / betas / eps / weight_decay of every flat group to the device if they changed
        (schedulers and ``adapt_betas`` edit ``param_groups`` on the host).  Call before replaying a captured step."""
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is not None:
                self._push_hyper(st, group)

    def state_dict(self):
        """torch.optim.Adam layout; the step count of capturable mode lives on the device and is read back first."""
        if self.capturable:
            self.sync_step_from_device()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """The loaded exp_avg / exp_avg_sq / step replace the flat moments: drop the flat groups so that the next
        step re-adopts them from ``self.state`` (the cache key alone -- parameter addresses -- would not change)."""
        out = super().load_state_dict(state_dict)
        self._flat = {}
        for group in self.param_groups:
            for p in group["params"]:
                s = self.state.get(p)
                if s is not None and "step" in s and torch.is_tensor(s["step"]):
                    s["step"] = int(s["step"].item())
        return out

    def sync_step_from_device(self):
        """capturable mode: copy the device step counters into the per-parameter state (before state_dict())."""
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is not None:
                step = int(st["step_dev"].item())
                for p in group["params"]:
                    self.state[p]["step"] = step

    def zero_grad(self, set_to_none=False):
        """Keeps ``p.grad`` attached (the flat gradient buffer is the kernels' accumulation target):
        one memset when the gradients are views of a flat buffer, per-tensor otherwise."""
        for gi, group in enumerate(self.param_groups):
            st = self._flat_group(gi, group)
            if st is not None:
                st["g"].zero_()
                continue
            for p in group["params"]:
                if p.grad is not None:
                    if p.grad.grad_fn is not None:
                        p.grad = p.grad.detach()
                    p.grad.zero_()


def attach_flat_grads(model):
    """Point every ``p.grad`` of a FlatParamsMixin model at its slice of the flat gradient buffer (zeroed),
    so backward accumulates in place and FusedAdam / the all-reduce see one contiguous operand."""
    model._ensure_flat()
    model._flat_grad.zero_()
    for (_, p), v in zip(model.named_parameters(), model._flat_views):
        p.grad = v
    return model._flat_grad
