#!/usr/bin/env python3
"""Generate tests/golden/micro_lightning.ckpt (build container only): a checkpoint shaped like the ones Lightning's
ModelCheckpoint writes for the reference (scripts/train_trocr.py:136-143), built FROM THE REFERENCE MODEL ITSELF
(tools/gen_golden.py shims), so that key names, key order and tensor sharing are the reference's own:

  * ``state_dict`` = ``TrOCRModel.state_dict()`` of a seeded micro model (transformers 5.x ViT spelling in this container;
    the test also loads a copy re-spelled per 4.57),
  * ``optimizer_states[0]`` in schedulefree's per-parameter layout ({"z", "exp_avg_sq"} per ``model.parameters()`` index;
    schedulefree itself is absent, so the VALUES are made here: z = p + 0.01, exp_avg_sq = p**2),
  * ``hyper_parameters`` as ``save_hyperparameters()`` records them, ``ema_shadow`` as src/callbacks/ema.py:75-85 adds it,
  * ``reference/*``: the logits and loss the reference computes with these weights on a seeded batch (the load test's oracle).

It also checks the other direction in the container: a checkpoint written by kzv/checkpoint.py (both spellings) loads into
the reference ``TrOCRModel.load_state_dict(strict=True)`` and reproduces the same logits.

Run:  python -B tools/gen_ckpt_fixture.py
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import torch  # noqa: E402

from gen_golden import build_reference, install_shims  # noqa: E402
from kzv import checkpoint as CK  # noqa: E402
from kzv import params as P  # noqa: E402
from kzv.config import micro_config  # noqa: E402
from kzv.data import synthetic_batch  # noqa: E402


def main():
    install_shims()
    cfg = micro_config()
    with tempfile.TemporaryDirectory() as tmp:
        model, flat = build_reference(cfg, tmp, seed=7)
        names = [n for n, _ in model.named_parameters()]
        # the restated orders must be the reference's own (5.x spelling here)
        assert [P.canonical_hf_name(n) for n in names] == CK.reference_parameter_order(cfg, "hf5")
        assert [P.canonical_hf_name(k) for k in model.state_dict()] == CK.reference_state_dict_order(cfg, "hf5")
        px, lab = synthetic_batch(cfg, 3, 10, seed=5, min_chars=2, max_chars=9)
        with torch.no_grad():
            out = model(torch.from_numpy(px), torch.from_numpy(lab))
        sd = model.state_dict()
        state = {i: {"z": (p.detach() + 0.01).clone(), "exp_avg_sq": (p.detach() ** 2).clone()} for i, (n, p) in enumerate(model.named_parameters())}
        group = {"lr": 1e-4, "betas": (0.9, 0.999), "eps": 1e-8, "r": 0.0, "k": 17, "train_mode": True, "weight_sum": 3.5e-8,
                 "lr_max": 9e-5, "scheduled_lr": 9e-5, "weight_lr_power": 2.0, "weight_decay": 0, "foreach": None,
                 "silent_sgd_phase": True, "params": list(range(len(names)))}
        ck = {"epoch": 3, "global_step": 17, "pytorch-lightning_version": CK.PL_VERSION, "state_dict": sd, "loops": {}, "callbacks": {},
              "optimizer_states": [{"state": state, "param_groups": [group]}], "lr_schedulers": [], "hparams_name": "kwargs",
              "hyper_parameters": {"encoder_config": cfg.encoder_config_dict(), "decoder_path": "decoder_dir", "learning_rate": 1e-4,
                                   "beta1": 0.9, "beta2": 0.999, "epsilon": 1e-8, "weight_decay": 0},
              "ema_shadow": {n: (p.detach() * 0.5).clone() for n, p in model.named_parameters()},
              "reference/pixel_seed": 5, "reference/logits": out["logits"].clone(), "reference/loss": float(out["loss"]),
              "reference/labels": torch.from_numpy(lab)}
        path = os.path.join(ROOT, "tests", "golden", "micro_lightning.ckpt")
        torch.save(ck, path)
        print("wrote", path, os.path.getsize(path), "bytes;", len(names), "parameters,", len(sd), "state_dict keys")

        # the other direction: engine-written checkpoints load into the reference, strictly
        for spelling in ("hf5",):          # the container's transformers spells 5.x; "hf4" keys are checked by name below
            mine = CK.build_checkpoint(cfg, torch.from_numpy(flat), ck["hyper_parameters"], 1, 2, spelling=spelling)
            assert list(mine["state_dict"].keys()) == list(sd.keys())
            model.load_state_dict(mine["state_dict"], strict=True)
            with torch.no_grad():
                again = model(torch.from_numpy(px), torch.from_numpy(lab))
            assert torch.equal(again["logits"], out["logits"])
            w = mine["state_dict"]
            assert w["decoder.lm_head.decoder.weight"].data_ptr() == w["decoder.roberta.embeddings.word_embeddings.weight"].data_ptr()
        hf4 = CK.build_checkpoint(cfg, torch.from_numpy(flat), ck["hyper_parameters"], 1, 2, spelling="hf4")
        assert "encoder.encoder.layer.0.attention.attention.query.weight" in hf4["state_dict"]
        print("engine-written checkpoint loads into the reference (strict=True) and reproduces its logits")


if __name__ == "__main__":
    main()
