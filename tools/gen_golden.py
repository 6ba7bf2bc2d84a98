#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

Imports /root/reference/src/models/trocr_model.py::TrOCRModel with the three import shims
recorded in SURVEY.md section 8(c) (stub pytorch_lightning, stub schedulefree, a
ModuleList-of-ViTLayer stand-in for the transformers-4.57 ``ViTEncoder`` class that 5.x removed),
loads this repo's seeded weight recipe (kzv.params.recipe_flat) into it through the reference's
own state_dict names, runs eval-mode forward + backward, and stores inputs' seeds + outputs.
The reference never travels: only the .npz outputs are committed.

Run:  python -B tools/gen_golden.py        (from the repo root)
"""
from __future__ import annotations

import hashlib
import inspect
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from kzv import params as P  # noqa: E402
from kzv.config import ModelConfig, tiny_config, vit_b_config  # noqa: E402
from kzv.data import build_decoder_dir, synthetic_batch  # noqa: E402

REF_SRC = "/root/reference/src"


def install_shims():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self._hp = types.SimpleNamespace()

        def save_hyperparameters(self, *a, **k):
            loc = inspect.currentframe().f_back.f_locals
            for key, val in loc.items():
                if key not in ("self", "__class__"):
                    setattr(self._hp, key, val)

        @property
        def hparams(self):
            return self._hp

        def log(self, *a, **k):
            pass

        def optimizers(self):
            return None

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    sf = types.ModuleType("schedulefree")
    sf.RAdamScheduleFree = type("RAdamScheduleFree", (torch.optim.AdamW,), {})
    sys.modules["schedulefree"] = sf
    import transformers.models.vit.modeling_vit as mv
    from transformers.modeling_outputs import BaseModelOutput

    class ViTEncoder(nn.Module):  # what transformers 4.57 exported under this name
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.layer = nn.ModuleList([mv.ViTLayer(config) for _ in range(config.num_hidden_layers)])

        def forward(self, hidden_states, **kw):
            for lyr in self.layer:
                hidden_states = lyr(hidden_states)
            return BaseModelOutput(last_hidden_state=hidden_states)

    if not hasattr(mv, "ViTEncoder"):
        mv.ViTEncoder = ViTEncoder


def build_reference(cfg: ModelConfig, tmp: str, seed: int):
    sys.path.insert(0, REF_SRC)
    from models.trocr_model import TrOCRModel, get_2d_sincos_pos_embed
    from transformers import RobertaConfig, RobertaForMaskedLM

    d = build_decoder_dir(os.path.join(tmp, f"dec_{cfg.vocab}_{cfg.dec_hidden}"), cfg)
    torch.manual_seed(0)
    RobertaForMaskedLM(RobertaConfig(**cfg.decoder_config_dict())).save_pretrained(d)
    model = TrOCRModel(cfg.encoder_config_dict(), d)
    model.eval()
    # the recipe's sin/cos table must equal the reference's own (KAT)
    ref_pos = get_2d_sincos_pos_embed(cfg.enc_hidden, (cfg.grid_h, cfg.grid_w))
    assert np.array_equal(ref_pos.astype(np.float32), P.position_table(cfg)[1:]), "sincos restatement differs"
    flat = P.recipe_flat(cfg, seed)
    sd = P.state_dict_from_flat(cfg, flat)
    ref_sd = model.state_dict()
    mapped = {}
    for k, v in sd.items():
        k5 = P.to_hf5_name(k)
        assert k5 in ref_sd, f"reference has no key {k5}"
        assert tuple(ref_sd[k5].shape) == tuple(v.shape), (k5, ref_sd[k5].shape, v.shape)
        mapped[k5] = torch.from_numpy(np.ascontiguousarray(v))
    missing = [k for k in ref_sd if k not in mapped]
    assert not missing, f"recipe does not cover reference keys: {missing[:5]}"
    model.load_state_dict(mapped, strict=True)
    return model, flat


def capture(model, cfg):
    """Forward hooks on the reference modules -> stage activations (names match oracle.stages)."""
    st = {}
    hooks = []

    def add(mod, name, sel=lambda o: o):
        hooks.append(mod.register_forward_hook(lambda m, i, o: st.__setitem__(name, sel(o).detach().numpy().copy())))

    add(model.encoder.patch_embeddings, "patch_embed")
    for i, lyr in enumerate(model.encoder.encoder.layer):
        add(lyr, f"enc_layer{i}", lambda o: o[0] if isinstance(o, tuple) else o)
    add(model.encoder, "enc_out")
    add(model.encoder_decoder_proj, "proj_out")
    add(model.decoder.roberta.embeddings, "dec_embed")
    for i, lyr in enumerate(model.decoder.roberta.encoder.layer):
        add(lyr, f"dec_layer{i}", lambda o: o[0] if isinstance(o, tuple) else o)
    return st, hooks


def run_case(cfg, seed, batch, label_len, data_seed, tmp, labels_override=None):
    model, flat = build_reference(cfg, tmp, seed)
    px, labels = synthetic_batch(cfg, batch, label_len, seed=data_seed, min_chars=3, max_chars=label_len)
    if labels_override is not None:
        labels = labels_override(labels)
    st, hooks = capture(model, cfg)
    out = model(torch.from_numpy(px), torch.from_numpy(labels))
    out["loss"].backward()
    for h in hooks:
        h.remove()
    grads = {}
    for k5, p in model.named_parameters():
        grads[P.canonical_hf_name(k5)] = p.grad.detach().numpy().copy() if p.grad is not None else None
    return dict(model=model, flat=flat, px=px, labels=labels, logits=out["logits"].detach().numpy(),
                loss=float(out["loss"]), stages=st, grads=grads)


def main():
    install_shims()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        # ---- tiny: full tensors ------------------------------------------------
        cfg = tiny_config()

        def with_specials(lab):
            lab = lab.copy()
            lab[0, 0] = cfg.bos_id            # exercise ids 2/3
            n0 = int((lab[0] != cfg.pad_id).sum())
            lab[0, min(n0, lab.shape[1] - 1)] = cfg.eos_id
            lab[1, 5:] = cfg.pad_id           # short sample: long pad tail
            lab[2, :] = np.where(lab[2] == cfg.pad_id, 7, lab[2])  # no padding at all
            return lab

        r = run_case(cfg, seed=42, batch=3, label_len=24, data_seed=1, tmp=tmp, labels_override=with_specials)
        save = {"seed": 42, "data_seed": 1, "pixel_values": r["px"], "labels": r["labels"],
                "logits": r["logits"], "loss": np.float64(r["loss"]),
                "weights_sha256": hashlib.sha256(r["flat"].tobytes()).hexdigest()}
        for k, v in r["stages"].items():
            save["stage/" + k] = v
        for k, v in r["grads"].items():
            if v is not None:
                save["grad/" + k] = v
        np.savez_compressed(os.path.join(out_dir, "tiny_fwd_bwd.npz"), **save)
        print("tiny: loss", r["loss"], "logits", r["logits"].shape, "stages", sorted(r["stages"]))

        # tokenizer KAT: chars -> ids 5.., pad to max_length, no BOS/EOS; whitespace -> UNK
        tok = r["model"].tokenizer
        chars = "".join(chr(c) for c in (0x4E00, 0x4E01, 0x4E02))
        enc = tok(chars, max_length=8, padding="max_length", truncation=True)["input_ids"]
        enc_ws = tok(chars[0] + " " + chars[1], max_length=8, padding="max_length", truncation=True)["input_ids"]
        enc_trunc = tok(chars * 5, max_length=8, padding="max_length", truncation=True)["input_ids"]
        dec = tok.batch_decode([enc], skip_special_tokens=True)[0]
        np.savez(os.path.join(out_dir, "tokenizer_kat.npz"), text=chars, ids=np.array(enc), ids_ws=np.array(enc_ws),
                 ids_trunc=np.array(enc_trunc), decoded=dec,
                 specials=np.array([tok.unk_token_id, tok.pad_token_id, tok.bos_token_id, tok.eos_token_id,
                                    tok.mask_token_id]))
        print("tokenizer KAT", enc, enc_ws, repr(dec))

        # CER KATs from the reference's own method (src/models/trocr_model.py:400-410); editdistance is
        # absent here, so bind a Levenshtein under that module name for this call only.
        ed = types.ModuleType("editdistance")
        sys.path.insert(0, os.path.join(ROOT))
        from oracle.trocr_oracle import levenshtein
        ed.eval = levenshtein
        sys.modules["editdistance"] = ed
        pairs = [("ac", "ab"), ("cot", "cat"), ("test", "test"), ("", "abc"), ("abc", ""), ("", ""), ("kitten", "sitting")]
        cers = [r["model"].calculate_cer(p, t) for p, t in pairs]
        np.savez(os.path.join(out_dir, "cer_kat.npz"), preds=np.array([p for p, _ in pairs]),
                 targets=np.array([t for _, t in pairs]), cer=np.array(cers))
        print("cer KAT", cers)

        # ---- ViT-B geometry (configs[1]), B=2: summary record -------------------
        cfgb = vit_b_config()
        rb = run_case(cfgb, seed=42, batch=2, label_len=128, data_seed=1, tmp=tmp)
        rng = np.random.default_rng(0)
        lg = rb["logits"]
        idx = np.stack([rng.integers(0, lg.shape[0], 4096), rng.integers(0, lg.shape[1], 4096),
                        rng.integers(0, lg.shape[2], 4096)], axis=1)
        saveb = {"seed": 42, "data_seed": 1, "batch": 2, "label_len": 128,
                 "loss": np.float64(rb["loss"]), "argmax": lg.argmax(-1).astype(np.int32),
                 "logit_idx": idx.astype(np.int32), "logit_val": lg[idx[:, 0], idx[:, 1], idx[:, 2]],
                 "top2_gap": np.sort(lg, axis=-1)[..., -1] - np.sort(lg, axis=-1)[..., -2],
                 "weights_sha256": hashlib.sha256(rb["flat"].tobytes()).hexdigest(),
                 "enc_out_sample": rb["stages"]["enc_out"][:, ::16, ::32],
                 "proj_out_sample": rb["stages"]["proj_out"][:, ::16, ::16]}
        names, norms = [], []
        for k, v in rb["grads"].items():
            if v is not None:
                names.append(k)
                norms.append(float(np.sqrt((v.astype(np.float64) ** 2).sum())))
        saveb["grad_names"] = np.array(names)
        saveb["grad_norms"] = np.array(norms)
        np.savez_compressed(os.path.join(out_dir, "vitb_b2_summary.npz"), **saveb)
        print("vit-b: loss", rb["loss"], "params", sum(p.numel() for p in rb["model"].parameters()))


if __name__ == "__main__":
    main()
