"""Model geometry for the TrOCR line-OCR training path.

Mirrors the two config sources the reference reads:
  * ``encoder_config`` dict keys/defaults -- src/models/trocr_model.py:234-244,
    scripts/train_trocr.py:111-121 (intermediate_size = 4*hidden at :118)
  * the decoder's HF ``config.json`` (RobertaConfig) -- train_language_model_scratch.py:406-428
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Any


@dataclasses.dataclass
class ModelConfig:
    # encoder (ViT) -- defaults = src/models/trocr_model.py:98-108
    image_h: int = 1024
    image_w: int = 64
    patch_h: int = 16
    patch_w: int = 16
    channels: int = 3
    enc_hidden: int = 768
    enc_layers: int = 12
    enc_heads: int = 12
    enc_ffn: int = 3072
    enc_hidden_dropout: float = 0.1
    enc_attn_dropout: float = 0.1
    # decoder (RoBERTa causal LM with cross attention) -- train_language_model_scratch.py:406-428
    dec_hidden: int = 256
    dec_layers: int = 12
    dec_heads: int = 4
    dec_ffn: int = 768
    vocab: int = 4300
    max_pos: int = 128
    type_vocab: int = 2
    pad_id: int = 1
    bos_id: int = 2
    eos_id: int = 3
    dec_hidden_dropout: float = 0.1
    dec_attn_dropout: float = 0.1
    ln_eps: float = 1e-12

    # ---- derived -------------------------------------------------------
    @property
    def grid_h(self) -> int:
        return self.image_h // self.patch_h

    @property
    def grid_w(self) -> int:
        return self.image_w // self.patch_w

    @property
    def num_patches(self) -> int:
        return self.grid_h * self.grid_w

    @property
    def enc_seq(self) -> int:  # patches + CLS
        return self.num_patches + 1

    @property
    def patch_dim(self) -> int:
        return self.channels * self.patch_h * self.patch_w

    @property
    def has_proj(self) -> bool:  # nn.Identity when equal (src/models/trocr_model.py:250-253)
        return self.enc_hidden != self.dec_hidden

    def validate(self) -> None:
        if self.enc_hidden % self.enc_heads or self.dec_hidden % self.dec_heads:
            raise ValueError("hidden size must be a multiple of the number of heads")
        if self.dec_hidden // self.dec_heads != 64:
            raise ValueError("the decoder's attention / generation kernels are built for head_dim == 64")
        eh = self.enc_hidden // self.enc_heads
        if eh % 8 or eh > 128:      # 64: MFMA kernels; others (the reference CLI's default 768 / 8 = 96): the plain fp32 kernel
            raise ValueError("the encoder's head_dim must be a multiple of 8 up to 128")
        for k in (self.enc_hidden, self.enc_ffn, self.dec_hidden, self.dec_ffn, self.patch_dim):
            if k % 64:
                raise ValueError(f"GEMM reduction dims must be multiples of 64 (got {k})")

    # ---- constructors --------------------------------------------------
    @classmethod
    def from_reference(cls, encoder_config: dict[str, Any], decoder_cfg: dict[str, Any]) -> "ModelConfig":
        """encoder_config dict (reference keys) + decoder HF config.json dict -> ModelConfig."""
        ih, iw = encoder_config.get("image_size", (1024, 64))
        ph, pw = encoder_config.get("patch_size", (16, 16))
        return cls(
            image_h=int(ih), image_w=int(iw), patch_h=int(ph), patch_w=int(pw),
            channels=int(encoder_config.get("num_channels", 3)),
            enc_hidden=int(encoder_config.get("hidden_size", 768)),
            enc_layers=int(encoder_config.get("num_hidden_layers", 12)),
            enc_heads=int(encoder_config.get("num_attention_heads", 12)),
            enc_ffn=int(encoder_config.get("intermediate_size", 3072)),
            enc_hidden_dropout=float(encoder_config.get("hidden_dropout_prob", 0.1)),
            enc_attn_dropout=float(encoder_config.get("attention_probs_dropout_prob", 0.1)),
            dec_hidden=int(decoder_cfg["hidden_size"]),
            dec_layers=int(decoder_cfg["num_hidden_layers"]),
            dec_heads=int(decoder_cfg["num_attention_heads"]),
            dec_ffn=int(decoder_cfg["intermediate_size"]),
            vocab=int(decoder_cfg["vocab_size"]),
            max_pos=int(decoder_cfg.get("max_position_embeddings", 128)),
            type_vocab=int(decoder_cfg.get("type_vocab_size", 2)),
            pad_id=int(decoder_cfg.get("pad_token_id", 1)),
            bos_id=int(decoder_cfg.get("bos_token_id", 2) if decoder_cfg.get("bos_token_id") is not None else 2),
            eos_id=int(decoder_cfg.get("eos_token_id", 3) if decoder_cfg.get("eos_token_id") is not None else 3),
            dec_hidden_dropout=float(decoder_cfg.get("hidden_dropout_prob", 0.1)),
            dec_attn_dropout=float(decoder_cfg.get("attention_probs_dropout_prob", 0.1)),
            ln_eps=float(decoder_cfg.get("layer_norm_eps", 1e-12)),
        )

    def encoder_config_dict(self) -> dict[str, Any]:
        return {
            "image_size": (self.image_h, self.image_w),
            "patch_size": (self.patch_h, self.patch_w),
            "num_channels": self.channels,
            "hidden_size": self.enc_hidden,
            "num_hidden_layers": self.enc_layers,
            "num_attention_heads": self.enc_heads,
            "intermediate_size": self.enc_ffn,
            "hidden_dropout_prob": self.enc_hidden_dropout,
            "attention_probs_dropout_prob": self.enc_attn_dropout,
        }

    def decoder_config_dict(self) -> dict[str, Any]:
        """HF RobertaConfig fields as written by train_language_model_scratch.py:406-428."""
        return {
            "architectures": ["RobertaForMaskedLM"],
            "model_type": "roberta",
            "vocab_size": self.vocab,
            "max_position_embeddings": self.max_pos,
            "num_hidden_layers": self.dec_layers,
            "num_attention_heads": self.dec_heads,
            "hidden_size": self.dec_hidden,
            "type_vocab_size": self.type_vocab,
            "intermediate_size": self.dec_ffn,
            "mask_token_id": 4,
            "bos_token_id": self.bos_id,
            "eos_token_id": self.eos_id,
            "pad_token_id": self.pad_id,
            "attention_probs_dropout_prob": self.dec_attn_dropout,
            "classifier_dropout": None,
            "hidden_act": "gelu",
            "hidden_dropout_prob": self.dec_hidden_dropout,
            "initializer_range": 0.02,
            "layer_norm_eps": self.ln_eps,
            "position_embedding_type": "absolute",
            "use_cache": True,
        }


def load_decoder_config(decoder_path: str) -> dict[str, Any]:
    """Read ``config.json`` from a local HF directory (reference: AutoConfig.from_pretrained,
    src/models/trocr_model.py:225).  Raises FileNotFoundError like scripts/train_trocr.py:88-89."""
    p = os.path.join(decoder_path, "config.json")
    if not os.path.isdir(decoder_path) or not os.path.exists(p):
        raise FileNotFoundError(f"Decoder path not found: {decoder_path}")
    with open(p, encoding="utf-8") as f:
        return json.load(f)


# ---- named workloads (BASELINE.json:configs) ------------------------------
def vit_b_config(dec_layers: int = 12) -> ModelConfig:
    """configs[1]/[2]: ViT-B/16 on 64x640 crops + the reference RoBERTa decoder geometry."""
    return ModelConfig(image_h=64, image_w=640, enc_hidden=768, enc_layers=12, enc_heads=12, enc_ffn=3072,
                       dec_hidden=256, dec_layers=dec_layers, dec_heads=4, dec_ffn=768, vocab=4300, max_pos=128)


def vit_l_config(enc_layers: int = 24, dec_layers: int = 12, dec_hidden: int = 256, dec_heads: int = 4, dec_ffn: int = 768) -> ModelConfig:
    """configs[3]: TrOCR-large geometry -- ViT-L/16 encoder (1024 / 16 heads / FFN 4096) on 64x640 crops.  Two decoder variants:
    the reference decoder's widths (256 / 4 heads / FFN 768, the default) and, with ``dec_hidden=1024, dec_heads=16,
    dec_ffn=4096``, the TrOCR-large decoder SURVEY.md section 8(d) prices the config with (468 GFLOP per image; encoder and
    decoder widths are then equal, so ``encoder_decoder_proj`` is nn.Identity: src/models/trocr_model.py:250-253).
    `enc_layers` / `dec_layers` let tests keep the oracle fast."""
    return ModelConfig(image_h=64, image_w=640, enc_hidden=1024, enc_layers=enc_layers, enc_heads=16, enc_ffn=4096,
                       dec_hidden=dec_hidden, dec_layers=dec_layers, dec_heads=dec_heads, dec_ffn=dec_ffn, vocab=4300, max_pos=128)


def vit_l_wide_config(enc_layers: int = 24, dec_layers: int = 12) -> ModelConfig:
    """configs[3] as SURVEY.md section 8(d) prices it: ViT-L/16 + a 12-layer decoder at 1024 hidden / 16 heads / FFN 4096."""
    return vit_l_config(enc_layers, dec_layers, dec_hidden=1024, dec_heads=16, dec_ffn=4096)


def small_config() -> ModelConfig:
    """configs[0]: 'TrOCR-small' plumbing case (6 L / 384 / 6 heads encoder + reference decoder)."""
    return ModelConfig(image_h=64, image_w=640, enc_hidden=384, enc_layers=6, enc_heads=6, enc_ffn=1536,
                       dec_hidden=256, dec_layers=12, dec_heads=4, dec_ffn=768, vocab=4300, max_pos=128)


def tiny_config() -> ModelConfig:
    """Test geometry: every op exercised, seconds on a CPU. head_dim stays 64."""
    return ModelConfig(image_h=32, image_w=64, enc_hidden=128, enc_layers=2, enc_heads=2, enc_ffn=256,
                       dec_hidden=64, dec_layers=2, dec_heads=1, dec_ffn=128, vocab=157, max_pos=40)


def micro_config() -> ModelConfig:
    """Smallest geometry the engine accepts (checkpoint fixtures): 16x32 image, 8x8 patches, one layer each."""
    return ModelConfig(image_h=16, image_w=32, patch_h=8, patch_w=8, enc_hidden=64, enc_layers=1, enc_heads=1, enc_ffn=64,
                       dec_hidden=64, dec_layers=1, dec_heads=1, dec_ffn=64, vocab=37, max_pos=16)
