"""EMA of the model parameters over the engine's flat buffer.

Behaviour of the reference's ``EMACallback`` (src/callbacks/ema.py:4-98): shadow initialised from the
parameters at fit start (:45-49); after every batch ``shadow = decay*shadow + (1-decay)*param`` (:51-58);
parameters swapped for the shadow during validation and restored afterwards (:60-73); shadow saved in /
restored from the checkpoint dict under ``ema_shadow`` (:75-98).  The update is fused into the optimizer kernel
(kzv_clip_and_step_ema: the freshly stepped parameters are in registers there) when the model has an optimizer, else ONE
streaming kernel (kzv_lerp_params: s += (1-decay)*(p - s)) over the flat buffer -- never a Python loop over tensors.
"""
from __future__ import annotations

from . import _lib as L
from . import params as P


class EMACallback:
    def __init__(self, decay: float = 0.9999):
        if not (0.0 <= decay <= 1.0):
            raise ValueError("Decay must be between 0 and 1.")     # ema.py:18-19
        self.decay = decay
        self.shadow = None
        self.backup = None

    def on_fit_start(self, model) -> None:
        self.shadow = model.flat_params.clone()
        self._attach(model)

    def _attach(self, model) -> None:
        opt = model.optimizers() if hasattr(model, "optimizers") else None
        self._fused_with = opt if (opt is not None and hasattr(opt, "attach_ema")) else None
        if self._fused_with is not None:
            self._fused_with.attach_ema(self.shadow, self.decay)
            self._seen = getattr(self._fused_with, "ema_fused_steps", 0)

    def on_train_batch_end(self, model) -> None:
        if self.shadow is None:
            self.on_fit_start(model)
        fused = getattr(self, "_fused_with", None)
        if fused is not None and fused is model.optimizers() and fused._ema is not None and fused._ema[0] is self.shadow:
            done = getattr(fused, "ema_fused_steps", 0)
            if done > self._seen:            # this batch's optimizer step already updated the shadow
                self._seen = done
                return
        L.check(L.load().kzv_lerp_params(self.shadow.data_ptr(), model.flat_params.data_ptr(), self.shadow.numel(),
                                         1.0 - self.decay, L.stream_handle()), "ema update")

    def on_validation_start(self, model) -> None:
        if self.shadow is None:
            return
        self.backup = model.flat_params.clone()
        model.flat_params.copy_(self.shadow)
        model.sync_weights()

    def on_validation_end(self, model) -> None:
        if self.backup is None:
            return
        model.flat_params.copy_(self.backup)
        self.backup = None
        model.sync_weights()

    def on_save_checkpoint(self, model, checkpoint: dict) -> dict:
        if self.shadow is not None:   # HF-named CPU tensors, like the reference's name -> tensor dict
            checkpoint["ema_shadow"] = {k: v.detach().cpu().clone() for k, v in P.state_dict_from_flat(model.cfg, self.shadow).items()
                                        if k not in P.TIED_ALIASES}
        return checkpoint

    def on_load_checkpoint(self, model, checkpoint: dict) -> None:
        if "ema_shadow" not in checkpoint:
            print("Warning: EMA shadow parameters not found in checkpoint. Initializing EMA from current model parameters.")
            self.on_fit_start(model)
            return
        self.shadow = model.flat_params.clone()
        views = P.state_dict_from_flat(model.cfg, self.shadow)
        for k, v in checkpoint["ema_shadow"].items():
            name = P.canonical_hf_name(k)
            if name in views:
                views[name].copy_(v.to(self.shadow.device))
        self._attach(model)
