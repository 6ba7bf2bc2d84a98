"""CPU-side checks of the C ABI: the library loads, exports every symbol include/kzv.h declares, and its
parameter table equals kzv/params.py (no compute calls -- there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from kzv import _lib as L
from kzv import params as P
from kzv.config import tiny_config, vit_b_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.load()


def test_header_symbols_are_exported_and_bound(lib):
    hdr = open(os.path.join(ROOT, "include", "kzv.h"), encoding="utf-8").read()
    declared = set(re.findall(r"\b(kzv_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"kzv_model"}
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"libkzv.so lacks {name}"
        assert name in L.SYMBOLS, f"kzv/_lib.py does not bind {name}"
    assert lib.kzv_version() >= 1


def test_kernels_whose_inline_asm_clobbers_m0_hold_no_compiler_generated_m0_use():
    """The LDS-DMA helpers of the GEMM / attention-backward kernels write M0 from inline asm and leave it clobbered; hipcc cannot be
    told (m0 is reserved: a clobber entry is ignored), so tools/check_m0.py compiles those sources and scans the gfx950 assembly:
    no kernel mixes an asm M0 write with a compiler-generated M0 use."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_m0.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("cfg", [tiny_config(), vit_b_config(), vit_b_config(dec_layers=6)])
def test_param_table_matches_python(lib, cfg):
    from kzv.model import TrOCRModel  # noqa: F401  (import must not need a GPU)
    c = L.kzv_config(image_h=cfg.image_h, image_w=cfg.image_w, patch_h=cfg.patch_h, patch_w=cfg.patch_w, channels=cfg.channels,
                     enc_hidden=cfg.enc_hidden, enc_layers=cfg.enc_layers, enc_heads=cfg.enc_heads, enc_ffn=cfg.enc_ffn,
                     dec_hidden=cfg.dec_hidden, dec_layers=cfg.dec_layers, dec_heads=cfg.dec_heads, dec_ffn=cfg.dec_ffn,
                     vocab=cfg.vocab, max_pos=cfg.max_pos, type_vocab=cfg.type_vocab, pad_id=cfg.pad_id,
                     enc_hidden_dropout=0.1, enc_attn_dropout=0.1, dec_hidden_dropout=0.1, dec_attn_dropout=0.1, ln_eps=1e-12)
    h = C.c_void_p()
    L.check(lib.kzv_model_create(C.byref(c), C.byref(h)), "create")
    offs, total = P.param_offsets(cfg)
    table = P.param_table(cfg)
    assert lib.kzv_param_count(h) == len(table)
    assert lib.kzv_param_total(h) == total
    for i, (name, shape) in enumerate(table):
        nm, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int64(), C.c_int64()
        L.check(lib.kzv_param_info(h, i, C.byref(nm), C.byref(off), C.byref(rows), C.byref(cols)), "info")
        assert nm.value.decode() == name
        assert off.value == offs[name][0]
        assert rows.value * cols.value == int(__import__("numpy").prod(shape))
    # backward segments tile the whole gradient buffer exactly once
    n = lib.kzv_backward_segments(h)
    spans = []
    for s in range(n):
        lo, hi = C.c_int64(), C.c_int64()
        L.check(lib.kzv_backward_segment_range(h, s, C.byref(lo), C.byref(hi)), "range")
        spans.append((lo.value, hi.value))
    spans.sort()
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert lib.kzv_workspace_bytes(h, 4, 16) > 0
    lib.kzv_model_destroy(h)


def test_bad_geometry_is_rejected(lib):
    cfg = tiny_config()
    c = L.kzv_config(image_h=30, image_w=64, patch_h=16, patch_w=16, channels=3, enc_hidden=128, enc_layers=1, enc_heads=2,
                     enc_ffn=256, dec_hidden=64, dec_layers=1, dec_heads=1, dec_ffn=128, vocab=cfg.vocab, max_pos=40,
                     type_vocab=2, pad_id=1, ln_eps=1e-12)
    h = C.c_void_p()
    assert lib.kzv_model_create(C.byref(c), C.byref(h)) == -1
    assert b"not divisible" in lib.kzv_last_error()
    c.image_h = 32
    c.enc_heads = 32  # encoder head_dim 4: not a multiple of 8
    assert lib.kzv_model_create(C.byref(c), C.byref(h)) == -1
    assert b"encoder's head_dim" in lib.kzv_last_error()
    c.enc_heads = 4   # encoder head_dim 32: allowed (plain fp32 attention kernel); the decoder's must stay 64
    c.dec_heads = 2
    assert lib.kzv_model_create(C.byref(c), C.byref(h)) == -1
    assert b"decoder's head_dim" in lib.kzv_last_error()
    c.dec_heads = 1
    assert lib.kzv_model_create(C.byref(c), C.byref(h)) == 0
    lib.kzv_model_destroy(h)
