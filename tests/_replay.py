"""Test helper: the dropout masks a HIP training step draws, fetched from the library itself.

``kzv_debug_dropout_mask`` (include/kzv.h) evaluates the same device hash as every fused dropout epilogue; this module
enumerates the step's call sites (KZV_SITE_* in include/kzv.h) with their element-index geometry and returns the masks
under the names oracle/trocr_oracle.py::_drop expects, so the oracle can replay a dropout-ON step exactly."""
from __future__ import annotations

import torch

from kzv import _lib as L

SITE_ENC_EMB = 1
SITE_DEC_EMB = 1000


def site_enc(i: int, k: int) -> int:      # KZV_SITE_ENC_LAYER
    return 16 + 4 * i + k


def site_dec(i: int, k: int) -> int:      # KZV_SITE_DEC_LAYER
    return 1016 + 8 * i + k


def _mask(seed: int, site: int, p: float, rows: int, cols: int, ld: int | None = None):
    lib = L.load()
    out = torch.empty(rows, cols, dtype=torch.float32, device="cuda")
    key = lib.kzv_drop_key(seed, site)
    L.check(lib.kzv_debug_dropout_mask(key, p, rows, cols, cols if ld is None else ld, out.data_ptr(), L.stream_handle()), "mask")
    return out


def step_masks(cfg, seed: int, B: int, T: int) -> dict[str, torch.Tensor]:
    """Masks of kzv_forward_loss(train=1, seed) for a batch of B crops and a decoder of T (= active) positions."""
    Se, He, Hd, npatch = cfg.enc_seq, cfg.enc_hidden, cfg.dec_hidden, cfg.num_patches
    m: dict[str, torch.Tensor] = {}

    def hidden(name, site, p, rows, cols, shape):
        if p > 0:
            m[name] = _mask(seed, site, p, rows, cols).reshape(shape).cpu()

    def probs(name, site, p, heads, sq, sk):       # attention-probability sites: the 4 x 4-block generator
        if p > 0:
            lib = L.load()
            out = torch.empty(B * heads * sq, sk, dtype=torch.float32, device="cuda")
            L.check(lib.kzv_debug_attn_dropout_mask(lib.kzv_drop_key(seed, site), p, B * heads, sq, sk, out.data_ptr(), L.stream_handle()), "attn mask")
            m[name] = out.reshape(B, heads, sq, sk).cpu()

    hidden("enc_emb", SITE_ENC_EMB, cfg.enc_hidden_dropout, B * Se, He, (B, Se, He))
    for i in range(cfg.enc_layers):
        probs(f"enc{i}_attn", site_enc(i, 0), cfg.enc_attn_dropout, cfg.enc_heads, Se, Se)
        hidden(f"enc{i}_o", site_enc(i, 1), cfg.enc_hidden_dropout, B * Se, He, (B, Se, He))
        hidden(f"enc{i}_mlp", site_enc(i, 2), cfg.enc_hidden_dropout, B * Se, He, (B, Se, He))
    hidden("dec_emb", SITE_DEC_EMB, cfg.dec_hidden_dropout, B * T, Hd, (B, T, Hd))
    for i in range(cfg.dec_layers):
        probs(f"dec{i}_sa", site_dec(i, 0), cfg.dec_attn_dropout, cfg.dec_heads, T, T)
        hidden(f"dec{i}_sa_o", site_dec(i, 1), cfg.dec_hidden_dropout, B * T, Hd, (B, T, Hd))
        probs(f"dec{i}_ca", site_dec(i, 2), cfg.dec_attn_dropout, cfg.dec_heads, T, npatch)
        hidden(f"dec{i}_ca_o", site_dec(i, 3), cfg.dec_hidden_dropout, B * T, Hd, (B, T, Hd))
        hidden(f"dec{i}_ffn", site_dec(i, 4), cfg.dec_hidden_dropout, B * T, Hd, (B, T, Hd))
    torch.cuda.synchronize()
    return m
