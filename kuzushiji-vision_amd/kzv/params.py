"""Flat parameter table of the engine and its mapping to the reference's ``state_dict`` names.

The HIP engine keeps ONE fp32 master buffer (and one fp32 gradient buffer of the same layout).
Q/K/V weights of an attention block are stored fused ([3H, H]) so one MFMA GEMM produces all
three; the decoder's cross-attention K/V weights of all layers are stored as one [L*2H, H] block
so a single GEMM over the encoder states feeds every layer.  ``state_dict_views`` exposes
contiguous slices under the reference's HF names:

  encoder.*          -- src/models/trocr_model.py:132-152 (+ HF ViTLayer names, 4.57 spelling)
  encoder_decoder_proj.* -- :250-251
  decoder.*          -- HF RobertaForCausalLM (modeling_roberta.py:56-155, 186-398, 877-893);
                        lm_head.decoder.weight is TIED to word_embeddings (:684-687)
"""
from __future__ import annotations

import numpy as np

from .config import ModelConfig


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def param_table(cfg: ModelConfig) -> list[tuple[str, tuple[int, ...]]]:
    """Canonical (engine) order.  MUST match csrc/model.cpp:build_param_table -- a test checks it."""
    He, Fe, Hd, Fd = cfg.enc_hidden, cfg.enc_ffn, cfg.dec_hidden, cfg.dec_ffn
    t: list[tuple[str, tuple[int, ...]]] = []
    t.append(("enc.patch.w", (He, cfg.patch_dim)))
    t.append(("enc.patch.b", (He,)))
    t.append(("enc.cls", (He,)))
    t.append(("enc.pos", (cfg.enc_seq, He)))
    for i in range(cfg.enc_layers):
        p = f"enc.{i}."
        t += [(p + "ln1.w", (He,)), (p + "ln1.b", (He,)),
              (p + "qkv.w", (3 * He, He)), (p + "qkv.b", (3 * He,)),
              (p + "o.w", (He, He)), (p + "o.b", (He,)),
              (p + "ln2.w", (He,)), (p + "ln2.b", (He,)),
              (p + "fc1.w", (Fe, He)), (p + "fc1.b", (Fe,)),
              (p + "fc2.w", (He, Fe)), (p + "fc2.b", (He,))]
    t += [("enc.lnf.w", (He,)), ("enc.lnf.b", (He,))]
    if cfg.has_proj:
        t += [("proj.w", (Hd, He)), ("proj.b", (Hd,))]
    t += [("dec.word", (cfg.vocab, Hd)), ("dec.pos", (cfg.max_pos, Hd)), ("dec.type", (cfg.type_vocab, Hd)),
          ("dec.emb_ln.w", (Hd,)), ("dec.emb_ln.b", (Hd,))]
    t += [("dec.cross_kv.w", (cfg.dec_layers * 2 * Hd, Hd)), ("dec.cross_kv.b", (cfg.dec_layers * 2 * Hd,))]
    for i in range(cfg.dec_layers):
        p = f"dec.{i}."
        t += [(p + "sa_qkv.w", (3 * Hd, Hd)), (p + "sa_qkv.b", (3 * Hd,)),
              (p + "sa_o.w", (Hd, Hd)), (p + "sa_o.b", (Hd,)),
              (p + "sa_ln.w", (Hd,)), (p + "sa_ln.b", (Hd,)),
              (p + "ca_q.w", (Hd, Hd)), (p + "ca_q.b", (Hd,)),
              (p + "ca_o.w", (Hd, Hd)), (p + "ca_o.b", (Hd,)),
              (p + "ca_ln.w", (Hd,)), (p + "ca_ln.b", (Hd,)),
              (p + "fc1.w", (Fd, Hd)), (p + "fc1.b", (Fd,)),
              (p + "fc2.w", (Hd, Fd)), (p + "fc2.b", (Hd,)),
              (p + "out_ln.w", (Hd,)), (p + "out_ln.b", (Hd,))]
    t += [("head.dense.w", (Hd, Hd)), ("head.dense.b", (Hd,)),
          ("head.ln.w", (Hd,)), ("head.ln.b", (Hd,)), ("head.bias", (cfg.vocab,))]
    return t


def param_offsets(cfg: ModelConfig) -> tuple[dict[str, tuple[int, tuple[int, ...]]], int]:
    """name -> (element offset, shape); every entry starts on a 64-element (256 B) boundary."""
    off = 0
    out: dict[str, tuple[int, tuple[int, ...]]] = {}
    for name, shape in param_table(cfg):
        out[name] = (off, shape)
        off += _align(int(np.prod(shape)))
    return out, off


def num_parameters(cfg: ModelConfig) -> int:
    """Un-padded parameter count (98,238,412 for ViT-B + reference decoder)."""
    return sum(int(np.prod(s)) for _, s in param_table(cfg))


def hf_views(cfg: ModelConfig) -> list[tuple[str, str, int, tuple[int, ...]]]:
    """(hf_name, engine_name, row_offset_in_elements, shape) for every reference state_dict key.

    ViT-layer names use the transformers-4.57 spelling the reference pins (pyproject.toml:23);
    ``HF_VIT_ALIASES`` maps the 5.x spelling onto them for loading.
    """
    He, Hd = cfg.enc_hidden, cfg.dec_hidden
    v: list[tuple[str, str, int, tuple[int, ...]]] = []
    v.append(("encoder.patch_embeddings.projection.weight", "enc.patch.w", 0,
              (He, cfg.channels, cfg.patch_h, cfg.patch_w)))
    v.append(("encoder.patch_embeddings.projection.bias", "enc.patch.b", 0, (He,)))
    v.append(("encoder.cls_token", "enc.cls", 0, (1, 1, He)))
    v.append(("encoder.position_embeddings", "enc.pos", 0, (1, cfg.enc_seq, He)))
    for i in range(cfg.enc_layers):
        h = f"encoder.encoder.layer.{i}."
        e = f"enc.{i}."
        for j, n in enumerate(("query", "key", "value")):
            v.append((h + f"attention.attention.{n}.weight", e + "qkv.w", j * He * He, (He, He)))
            v.append((h + f"attention.attention.{n}.bias", e + "qkv.b", j * He, (He,)))
        v += [(h + "attention.output.dense.weight", e + "o.w", 0, (He, He)),
              (h + "attention.output.dense.bias", e + "o.b", 0, (He,)),
              (h + "intermediate.dense.weight", e + "fc1.w", 0, (cfg.enc_ffn, He)),
              (h + "intermediate.dense.bias", e + "fc1.b", 0, (cfg.enc_ffn,)),
              (h + "output.dense.weight", e + "fc2.w", 0, (He, cfg.enc_ffn)),
              (h + "output.dense.bias", e + "fc2.b", 0, (He,)),
              (h + "layernorm_before.weight", e + "ln1.w", 0, (He,)),
              (h + "layernorm_before.bias", e + "ln1.b", 0, (He,)),
              (h + "layernorm_after.weight", e + "ln2.w", 0, (He,)),
              (h + "layernorm_after.bias", e + "ln2.b", 0, (He,))]
    v += [("encoder.layernorm.weight", "enc.lnf.w", 0, (He,)), ("encoder.layernorm.bias", "enc.lnf.b", 0, (He,))]
    if cfg.has_proj:
        v += [("encoder_decoder_proj.weight", "proj.w", 0, (Hd, He)),
              ("encoder_decoder_proj.bias", "proj.b", 0, (Hd,))]
    r = "decoder.roberta."
    v += [(r + "embeddings.word_embeddings.weight", "dec.word", 0, (cfg.vocab, Hd)),
          (r + "embeddings.position_embeddings.weight", "dec.pos", 0, (cfg.max_pos, Hd)),
          (r + "embeddings.token_type_embeddings.weight", "dec.type", 0, (cfg.type_vocab, Hd)),
          (r + "embeddings.LayerNorm.weight", "dec.emb_ln.w", 0, (Hd,)),
          (r + "embeddings.LayerNorm.bias", "dec.emb_ln.b", 0, (Hd,))]
    for i in range(cfg.dec_layers):
        h = r + f"encoder.layer.{i}."
        e = f"dec.{i}."
        for j, n in enumerate(("query", "key", "value")):
            v.append((h + f"attention.self.{n}.weight", e + "sa_qkv.w", j * Hd * Hd, (Hd, Hd)))
            v.append((h + f"attention.self.{n}.bias", e + "sa_qkv.b", j * Hd, (Hd,)))
        v += [(h + "attention.output.dense.weight", e + "sa_o.w", 0, (Hd, Hd)),
              (h + "attention.output.dense.bias", e + "sa_o.b", 0, (Hd,)),
              (h + "attention.output.LayerNorm.weight", e + "sa_ln.w", 0, (Hd,)),
              (h + "attention.output.LayerNorm.bias", e + "sa_ln.b", 0, (Hd,)),
              (h + "crossattention.self.query.weight", e + "ca_q.w", 0, (Hd, Hd)),
              (h + "crossattention.self.query.bias", e + "ca_q.b", 0, (Hd,))]
        for j, n in enumerate(("key", "value")):
            v.append((h + f"crossattention.self.{n}.weight", "dec.cross_kv.w", (2 * i + j) * Hd * Hd, (Hd, Hd)))
            v.append((h + f"crossattention.self.{n}.bias", "dec.cross_kv.b", (2 * i + j) * Hd, (Hd,)))
        v += [(h + "crossattention.output.dense.weight", e + "ca_o.w", 0, (Hd, Hd)),
              (h + "crossattention.output.dense.bias", e + "ca_o.b", 0, (Hd,)),
              (h + "crossattention.output.LayerNorm.weight", e + "ca_ln.w", 0, (Hd,)),
              (h + "crossattention.output.LayerNorm.bias", e + "ca_ln.b", 0, (Hd,)),
              (h + "intermediate.dense.weight", e + "fc1.w", 0, (cfg.dec_ffn, Hd)),
              (h + "intermediate.dense.bias", e + "fc1.b", 0, (cfg.dec_ffn,)),
              (h + "output.dense.weight", e + "fc2.w", 0, (Hd, cfg.dec_ffn)),
              (h + "output.dense.bias", e + "fc2.b", 0, (Hd,)),
              (h + "output.LayerNorm.weight", e + "out_ln.w", 0, (Hd,)),
              (h + "output.LayerNorm.bias", e + "out_ln.b", 0, (Hd,))]
    v += [("decoder.lm_head.dense.weight", "head.dense.w", 0, (Hd, Hd)),
          ("decoder.lm_head.dense.bias", "head.dense.b", 0, (Hd,)),
          ("decoder.lm_head.layer_norm.weight", "head.ln.w", 0, (Hd,)),
          ("decoder.lm_head.layer_norm.bias", "head.ln.b", 0, (Hd,)),
          ("decoder.lm_head.bias", "head.bias", 0, (cfg.vocab,)),
          # tied aliases (same storage): modeling_roberta.py:684-687
          ("decoder.lm_head.decoder.weight", "dec.word", 0, (cfg.vocab, Hd)),
          ("decoder.lm_head.decoder.bias", "head.bias", 0, (cfg.vocab,))]
    return v


TIED_ALIASES = ("decoder.lm_head.decoder.weight", "decoder.lm_head.decoder.bias")

# transformers 5.x ViTLayer spelling -> 4.57 spelling (HF modeling_vit.py:202-205,246-247)
HF_VIT_ALIASES = {
    "attention.q_proj": "attention.attention.query",
    "attention.k_proj": "attention.attention.key",
    "attention.v_proj": "attention.attention.value",
    "attention.o_proj": "attention.output.dense",
    "mlp.fc1": "intermediate.dense",
    "mlp.fc2": "output.dense",
}


def canonical_hf_name(name: str) -> str:
    """Accept either transformers spelling of a ViT layer key; also the 5.x 'encoder.encoder.layers.'"""
    if name.startswith("encoder.encoder.layer"):
        name = name.replace("encoder.encoder.layers.", "encoder.encoder.layer.")
        for new, old in HF_VIT_ALIASES.items():
            name = name.replace("." + new + ".", "." + old + ".")
    return name


def to_hf5_name(name: str) -> str:
    """4.57 spelling -> 5.x spelling (used by tools/gen_golden.py to load the in-container HF)."""
    if name.startswith("encoder.encoder.layer."):
        for new, old in HF_VIT_ALIASES.items():
            name = name.replace("." + old + ".", "." + new + ".")
    return name


# ---- sin/cos table: src/models/trocr_model.py:11-58,154-167 -----------------
def sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    """[sin(pos*w), cos(pos*w)], w_k = 1/10000^(k/(dim/2))  (trocr_model.py:40-58), float32 math."""
    omega = np.arange(dim // 2, dtype=np.float32)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(dim: int, grid_h: int, grid_w: int) -> np.ndarray:
    """[grid_h*grid_w, dim]; first half encodes the W index, second half the H index
    (meshgrid(w, h) with w first: trocr_model.py:18-24, 33-36)."""
    gh = np.arange(grid_h, dtype=np.float32)
    gw = np.arange(grid_w, dtype=np.float32)
    gx, gy = np.meshgrid(gw, gh)  # gx[i, j] = j (w), gy[i, j] = i (h)
    a = sincos_1d(dim // 2, gx)
    b = sincos_1d(dim // 2, gy)
    return np.concatenate([a, b], axis=1)


def position_table(cfg: ModelConfig) -> np.ndarray:
    """[1 + num_patches, He] float32 with a zero CLS row (trocr_model.py:163-167)."""
    t = np.zeros((cfg.enc_seq, cfg.enc_hidden), dtype=np.float64)
    t[1:] = sincos_2d(cfg.enc_hidden, cfg.grid_h, cfg.grid_w)
    return t.astype(np.float32)


# ---- deterministic weight recipe ----------------------------------------------
def recipe_flat(cfg: ModelConfig, seed: int) -> np.ndarray:
    """Seeded random weights for the flat buffer (fixtures regenerate these, so only outputs are stored).

    Entry ``i`` of the table draws from ``default_rng([seed, i])``: matrices/embeddings/biases
    ~ N(0, 0.02) (biases non-zero on purpose, so bias paths are exercised), LayerNorm gains
    1 + N(0, 0.02), enc.pos = the sin/cos table.
    """
    offs, total = param_offsets(cfg)
    flat = np.zeros(total, dtype=np.float32)
    for i, (name, shape) in enumerate(param_table(cfg)):
        n = int(np.prod(shape))
        off = offs[name][0]
        rng = np.random.default_rng([seed, i])
        if name == "enc.pos":
            val = position_table(cfg).reshape(-1)
        else:
            val = (rng.standard_normal(n) * 0.02).astype(np.float32)
            if name.endswith("ln.w") or name.endswith("ln1.w") or name.endswith("ln2.w") or name.endswith("lnf.w"):
                val = val + 1.0
        flat[off:off + n] = val
    return flat


def state_dict_from_flat(cfg: ModelConfig, flat):
    """HF-named views (numpy or torch, same slicing) into a flat buffer."""
    offs, _ = param_offsets(cfg)
    out = {}
    for hf, eng, rel, shape in hf_views(cfg):
        base = offs[eng][0] + rel
        n = int(np.prod(shape))
        out[hf] = flat[base:base + n].reshape(shape)
    return out
