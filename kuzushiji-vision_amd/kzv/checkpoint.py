"""Checkpoint interchange with the reference.

The reference trains under ``pl.Trainer`` with ``ModelCheckpoint(dirpath=.../checkpoints,
filename="trocr-{epoch:02d}-{val_loss:.2f}", monitor="val_loss", save_top_k=3, save_last=True)``
(scripts/train_trocr.py:136-143); what Lightning 2.5 writes is a pickled dict:

  epoch, global_step, pytorch-lightning_version, state_dict, loops, callbacks, optimizer_states, lr_schedulers,
  hparams_name, hyper_parameters  (+ ``ema_shadow`` added by src/callbacks/ema.py:75-85)

* ``state_dict``: ``TrOCRModel.state_dict()`` -- HF key names in module registration order, tied tensors under both
  names (``decoder.lm_head.decoder.{weight,bias}`` share storage with the word embeddings / ``lm_head.bias``).  ViT layer
  keys are spelled per transformers version: 4.57 (the reference's pin, ``spelling="hf4"``, default) or 5.x (``"hf5"``).
* ``optimizer_states[0]`` = ``schedulefree.RAdamScheduleFree.state_dict()``: ``{"state": {i: {"z", "exp_avg_sq"}},
  "param_groups": [{lr, betas, eps, r, k, train_mode, weight_sum, lr_max, scheduled_lr, weight_lr_power, weight_decay,
  foreach, silent_sgd_phase, params: [0..n-1]}]}`` with i indexing ``model.parameters()`` (shared tensors once).
  schedulefree==1.4.1 is absent from the build container: this layout is restated from its published source and is
  UNPINNED, like the optimizer arithmetic.  The parameter ORDER is pinned for the 5.x spelling by
  tests/golden/micro_lightning.ckpt (generated from the reference model itself, tools/gen_ckpt_fixture.py) and restated from
  the 4.57 sources for "hf4".
* ``hyper_parameters``: the constructor arguments ``save_hyperparameters()`` records (src/models/trocr_model.py:208-219).

Everything here is host code on CPU tensors; the engine's flat buffers are sliced through kzv/params.py.
"""
from __future__ import annotations

import numpy as np

from . import params as P

PL_VERSION = "2.5.1.post0"      # pyproject.toml:38 of the reference


def reference_state_dict_order(cfg, spelling: str = "hf4") -> list[str]:
    """Canonical (4.57-spelled) names of every reference ``state_dict`` key, in the reference's registration order for
    the transformers version that spells ViT layers ``spelling``.  TrOCRModel registers ``decoder`` before ``encoder``
    before ``encoder_decoder_proj`` (src/models/trocr_model.py:222-253); ViTEncoder's own parameters (cls_token,
    position_embeddings) come before its sub-modules."""
    r = "decoder.roberta."
    emb5 = ["word_embeddings.weight", "token_type_embeddings.weight", "LayerNorm.weight", "LayerNorm.bias", "position_embeddings.weight"]
    emb4 = ["word_embeddings.weight", "position_embeddings.weight", "token_type_embeddings.weight", "LayerNorm.weight", "LayerNorm.bias"]
    out = [r + "embeddings." + n for n in (emb5 if spelling == "hf5" else emb4)]
    for i in range(cfg.dec_layers):
        h = r + f"encoder.layer.{i}."
        for blk in ("attention", "crossattention"):
            for n in ("query", "key", "value"):
                out += [h + f"{blk}.self.{n}.weight", h + f"{blk}.self.{n}.bias"]
            out += [h + f"{blk}.output.dense.weight", h + f"{blk}.output.dense.bias",
                    h + f"{blk}.output.LayerNorm.weight", h + f"{blk}.output.LayerNorm.bias"]
        out += [h + "intermediate.dense.weight", h + "intermediate.dense.bias", h + "output.dense.weight", h + "output.dense.bias",
                h + "output.LayerNorm.weight", h + "output.LayerNorm.bias"]
    out += ["decoder.lm_head.bias", "decoder.lm_head.dense.weight", "decoder.lm_head.dense.bias",
            "decoder.lm_head.layer_norm.weight", "decoder.lm_head.layer_norm.bias",
            "decoder.lm_head.decoder.weight", "decoder.lm_head.decoder.bias"]
    out += ["encoder.cls_token", "encoder.position_embeddings",
            "encoder.patch_embeddings.projection.weight", "encoder.patch_embeddings.projection.bias"]
    for i in range(cfg.enc_layers):
        h = f"encoder.encoder.layer.{i}."
        att = []
        for n in ("query", "key", "value"):
            att += [h + f"attention.attention.{n}.weight", h + f"attention.attention.{n}.bias"]
        att += [h + "attention.output.dense.weight", h + "attention.output.dense.bias"]
        ln = [h + "layernorm_before.weight", h + "layernorm_before.bias", h + "layernorm_after.weight", h + "layernorm_after.bias"]
        mlp = [h + "intermediate.dense.weight", h + "intermediate.dense.bias", h + "output.dense.weight", h + "output.dense.bias"]
        out += att + (ln + mlp if spelling == "hf5" else mlp + ln)     # 5.x: attention, layernorms, mlp; 4.57: attention, intermediate, output, layernorms
    out += ["encoder.layernorm.weight", "encoder.layernorm.bias"]
    if cfg.has_proj:
        out += ["encoder_decoder_proj.weight", "encoder_decoder_proj.bias"]
    return out


def reference_parameter_order(cfg, spelling: str = "hf4") -> list[str]:
    """``[n for n, _ in model.named_parameters()]``: the state_dict order without the tied aliases."""
    return [k for k in reference_state_dict_order(cfg, spelling) if k not in P.TIED_ALIASES]


def _spell(name: str, spelling: str) -> str:
    return P.to_hf5_name(name) if spelling == "hf5" else name


def build_checkpoint(cfg, flat_params, hparams: dict, epoch: int, global_step: int, optimizer: dict | None = None,
                     spelling: str = "hf4", extra: dict | None = None) -> dict:
    """Lightning-shaped checkpoint dict from the engine's flat fp32 buffers (CPU torch tensors or numpy).
    ``optimizer`` = {"z": flat, "v": flat, "k", "lr_max", "weight_sum", "scheduled_lr", "train_mode", "lr", "betas", "eps",
    "weight_decay", "r", "weight_lr_power", "silent_sgd_phase"} (kzv.optim.RAdamScheduleFree.state_dict())."""
    import torch

    def views(flat):
        t = torch.as_tensor(np.asarray(flat) if not isinstance(flat, torch.Tensor) else flat).detach().cpu().clone()
        return P.state_dict_from_flat(cfg, t)
    pv = views(flat_params)
    sd = {_spell(k, spelling): pv[k] for k in reference_state_dict_order(cfg, spelling)}     # aliases share storage with their twins
    ck = {"epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": PL_VERSION, "state_dict": sd,
          "loops": {}, "callbacks": {}, "optimizer_states": [], "lr_schedulers": [], "hparams_name": "kwargs",
          "hyper_parameters": dict(hparams)}
    if optimizer is not None:
        zv, vv = views(optimizer["z"]), views(optimizer["v"])
        order = reference_parameter_order(cfg, spelling)
        state = {i: {"z": zv[k], "exp_avg_sq": vv[k]} for i, k in enumerate(order)}
        group = {"lr": optimizer["lr"], "betas": tuple(optimizer["betas"]), "eps": optimizer["eps"], "r": optimizer.get("r", 0.0),
                 "k": int(optimizer["k"]), "train_mode": bool(optimizer["train_mode"]), "weight_sum": float(optimizer["weight_sum"]),
                 "lr_max": float(optimizer["lr_max"]), "scheduled_lr": float(optimizer.get("scheduled_lr", 0.0)),
                 "weight_lr_power": optimizer.get("weight_lr_power", 2.0), "weight_decay": optimizer["weight_decay"],
                 "foreach": None, "silent_sgd_phase": optimizer.get("silent_sgd_phase", True), "params": list(range(len(order)))}
        ck["optimizer_states"] = [{"state": state, "param_groups": [group]}]
    if extra:
        ck.update(extra)
    return ck


_HF_BUFFERS = ("position_ids", "token_type_ids")      # HF registered buffers (persistent in some transformers versions)


def parameter_names_of(state_dict) -> list[str]:
    """Parameter order of a LOADED reference checkpoint: its state_dict keys in order, shared storages once (the first
    name wins, as ``named_parameters`` does), registered buffers dropped."""
    seen, out = set(), []
    for k, v in state_dict.items():
        if k.endswith(_HF_BUFFERS):
            continue
        key = (v.untyped_storage().data_ptr(), v.storage_offset(), tuple(v.shape)) if hasattr(v, "untyped_storage") else id(v)
        if key in seen:
            continue
        seen.add(key)
        out.append(k)
    return out


def read_checkpoint(ck: dict, cfg):
    """-> (state_dict under canonical names, optimizer dict in kzv.optim layout or None).  Accepts either ViT spelling;
    the optimizer state is mapped through the checkpoint's OWN parameter order."""
    import torch
    # HF registered buffers (persisted by some transformers versions) are not parameters: same filter as parameter_names_of
    sd = {P.canonical_hf_name(k): v for k, v in ck["state_dict"].items() if not k.endswith(_HF_BUFFERS)}
    opt = None
    if ck.get("optimizer_states"):
        osd = ck["optimizer_states"][0]
        if "state" not in osd or "param_groups" not in osd:
            raise ValueError("optimizer_states[0] is not in schedulefree's {'state', 'param_groups'} layout (checkpoints written by "
                             "this engine before round 2 used a flat layout that is no longer loadable: re-save from the weights)")
        names = [P.canonical_hf_name(k) for k in parameter_names_of(ck["state_dict"])]
        if len(names) != len(osd["state"]):
            raise ValueError(f"optimizer state has {len(osd['state'])} entries, the state_dict {len(names)} parameters")
        _, total = P.param_offsets(cfg)
        z, v = torch.zeros(total), torch.zeros(total)
        zv, vv = P.state_dict_from_flat(cfg, z), P.state_dict_from_flat(cfg, v)
        for i, name in enumerate(names):
            st = osd["state"][i]
            zv[name].copy_(st["z"].reshape(zv[name].shape))
            vv[name].copy_(st["exp_avg_sq"].reshape(vv[name].shape))
        g = osd["param_groups"][0]
        opt = {"z": z, "v": v, "k": g["k"], "lr_max": g["lr_max"], "weight_sum": g["weight_sum"], "train_mode": g["train_mode"],
               "scheduled_lr": g.get("scheduled_lr", 0.0), "lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"],
               "weight_decay": g["weight_decay"], "r": g.get("r", 0.0), "weight_lr_power": g.get("weight_lr_power", 2.0),
               "silent_sgd_phase": g.get("silent_sgd_phase", True)}
    return sd, opt
