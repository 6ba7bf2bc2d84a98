"""RAdamScheduleFree over the engine's flat fp32 buffers (host scalars here, arithmetic in optim.hip).

Stands in for ``schedulefree.RAdamScheduleFree`` as the reference configures it
(src/models/trocr_model.py:412-421: lr 1e-4, betas (0.9, 0.999), eps 1e-8, weight_decay 0) including the
``.train()`` / ``.eval()`` parameter swap its hooks rely on (:423-451).  schedulefree==1.4.1 is not
installed in the build container, so this restates the published algorithm (Schedule-Free AdamW +
RAdam rectification, silent SGD phase on): parity UNPINNED -- tests compare against
oracle/trocr_oracle.py::radam_schedulefree_step, which restates the same text.

``gradient_clip_val`` (scripts/train_trocr.py:175) is fused into the same kernel: the host passes
``max_grad_norm`` and the kernel scales by min(1, max_norm / (||g|| + 1e-6)) like torch's clip_grad_norm_.
"""
from __future__ import annotations

import ctypes as C
import math

from . import _lib as L


class RAdamScheduleFree:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 r=0.0, weight_lr_power=2.0, silent_sgd_phase=True):
        import torch
        self.model = model
        self.lr, (self.beta1, self.beta2), self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.r, self.weight_lr_power, self.silent_sgd_phase = r, weight_lr_power, silent_sgd_phase
        self.k = 0
        self.lr_max = -1.0
        self.weight_sum = 0.0
        self.scheduled_lr = 0.0
        self.train_mode = True
        flat = model.flat_params
        self.z = flat.clone()
        self.v = torch.zeros_like(flat)
        self._sq = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self._scratch = torch.zeros(2048, dtype=torch.float32, device=flat.device)
        self._ema = None            # (shadow tensor, decay): kzv.ema.EMACallback attaches itself so that step() fuses its update
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]

    # ---- per-step scalars (identical to oracle.RAdamScheduleFreeState.next_scalars) ----
    def _next_scalars(self):
        step = self.k + 1
        beta2_t = self.beta2 ** step
        bc2 = 1.0 - beta2_t
        rho_inf = 2.0 / (1.0 - self.beta2) - 1.0
        rho_t = rho_inf - 2.0 * step * beta2_t / bc2
        if rho_t > 4.0:
            rect = math.sqrt((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t))
        else:
            rect = float(not self.silent_sgd_phase)
        lr = self.lr * rect
        self.scheduled_lr = lr
        self.lr_max = max(lr, self.lr_max)
        weight = (step ** self.r) * (self.lr_max ** self.weight_lr_power)
        self.weight_sum += weight
        ckp1 = weight / self.weight_sum if self.weight_sum != 0 else 0.0
        self.k = step
        return lr, ckp1, bc2, rho_t > 4.0

    def step(self, max_grad_norm: float = 0.0, grad_scale: float = 1.0):
        """One optimizer step on the model's flat buffers (must be in train mode, like schedulefree)."""
        if not self.train_mode:
            raise RuntimeError("RAdamScheduleFree.step() called in eval mode; call optimizer.train() first")
        lib = L.load()
        m = self.model
        lr, ckp1, bc2, adaptive = self._next_scalars()
        self._last_scale = abs(grad_scale)
        st = L.stream_handle()
        n = m.flat_params.numel()
        if max_grad_norm > 0:
            L.check(lib.kzv_grad_sqnorm(m.flat_grads.data_ptr(), n, self._sq.data_ptr(), self._scratch.data_ptr(), st), "grad_sqnorm")
        s = L.kzv_opt_step(lr_t=lr, ckp1=ckp1, beta1=self.beta1, beta2=self.beta2, eps=self.eps,
                           weight_decay=self.weight_decay, bias_correction2=bc2, adaptive=int(adaptive),
                           max_grad_norm=max_grad_norm, grad_scale=grad_scale, one_minus_beta2=1.0 - self.beta2)
        ema, decay = self._ema if self._ema is not None else (None, 0.0)
        L.check(lib.kzv_clip_and_step_ema(m.flat_params.data_ptr(), self.z.data_ptr(), self.v.data_ptr(),
                                          m.flat_grads.data_ptr(), n, self._sq.data_ptr(), C.byref(s), L.ptr(ema), decay, st), "clip_and_step")
        self.ema_fused_steps = getattr(self, "ema_fused_steps", 0) + (1 if ema is not None else 0)
        m.sync_weights()

    def attach_ema(self, shadow, decay: float):
        """EMA shadow updated inside the optimizer kernel (the parameters are in registers there): kzv.ema.EMACallback."""
        self._ema = (shadow, float(decay)) if shadow is not None else None

    def grad_norm(self) -> float:
        """Total L2 norm of the gradient the last step clipped, i.e. AFTER the data-parallel mean (grad_scale applied)."""
        return float(self._sq.sqrt().item()) * getattr(self, "_last_scale", 1.0)

    def zero_grad(self, set_to_none: bool = False):
        self.model.zero_grad()

    def _lerp(self, w: float):
        lib = L.load()
        m = self.model
        L.check(lib.kzv_lerp_params(m.flat_params.data_ptr(), self.z.data_ptr(), m.flat_params.numel(), w, L.stream_handle()), "lerp")
        m.sync_weights()

    def eval(self):
        if self.train_mode:
            self._lerp(1.0 - 1.0 / self.beta1)   # y -> x
            self.train_mode = False

    def train(self):
        if not self.train_mode:
            self._lerp(1.0 - self.beta1)         # x -> y
            self.train_mode = True

    def state_dict(self):
        """Flat-buffer form of schedulefree's state (kzv/checkpoint.py turns it into the per-parameter layout)."""
        return {"k": self.k, "lr_max": self.lr_max, "weight_sum": self.weight_sum, "train_mode": self.train_mode,
                "scheduled_lr": self.scheduled_lr, "z": self.z.detach().cpu(), "v": self.v.detach().cpu(),
                "lr": self.lr, "betas": (self.beta1, self.beta2), "eps": self.eps, "weight_decay": self.weight_decay,
                "r": self.r, "weight_lr_power": self.weight_lr_power, "silent_sgd_phase": self.silent_sgd_phase}

    def load_state_dict(self, sd):
        self.k, self.lr_max, self.weight_sum = int(sd["k"]), float(sd["lr_max"]), float(sd["weight_sum"])
        self.scheduled_lr = float(sd.get("scheduled_lr", 0.0))
        self.train_mode = bool(sd["train_mode"])
        self.z.copy_(sd["z"])
        self.v.copy_(sd["v"])
