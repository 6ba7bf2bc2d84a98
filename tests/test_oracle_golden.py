"""Pin the CPU oracle (oracle/trocr_oracle.py) against fixtures produced by the REFERENCE itself
(tools/gen_golden.py imported /root/reference/src/models/trocr_model.py in the build container)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from kzv import params as P
from kzv.config import tiny_config, vit_b_config
from kzv.data import synthetic_batch
from oracle import trocr_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_param_count_matches_reference():
    # enc 85,771,008 + proj 196,864 + dec 12,270,540
    assert P.num_parameters(vit_b_config()) == 98_238_412


def test_sincos_kat():
    # get_2d_sincos_pos_embed(8,(2,3))[4]
    got = P.sincos_2d(8, 2, 3)[4]
    want = np.array([0.841471, 0.00999983, 0.5403023, 0.99995, 0.841471, 0.00999983, 0.5403023, 0.99995])
    np.testing.assert_allclose(got, want, atol=1e-6)


def test_tiny_forward_backward_matches_reference(golden_dir):
    g = _load(golden_dir, "tiny_fwd_bwd.npz")
    cfg = tiny_config()
    flat = P.recipe_flat(cfg, int(g["seed"]))
    assert hashlib.sha256(flat.tobytes()).hexdigest() == str(g["weights_sha256"])
    sd = P.state_dict_from_flat(cfg, flat)
    r = O.forward_backward(cfg, sd, g["pixel_values"], g["labels"], want_stages=True)
    np.testing.assert_allclose(r["logits"], g["logits"], atol=2e-5, rtol=0)
    assert abs(r["loss"] - float(g["loss"])) < 1e-5
    assert np.array_equal(r["logits"].argmax(-1), g["logits"].argmax(-1))
    for k in g.files:
        if k.startswith("stage/"):
            np.testing.assert_allclose(r["stages"][k[6:]], g[k], atol=2e-5, rtol=0, err_msg=k)
    n = 0
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            if name.startswith("decoder.lm_head.decoder."):
                continue
            got = r["grads"][name]
            assert got is not None, name
            np.testing.assert_allclose(got, g[k], atol=2e-6, rtol=1e-4, err_msg=name)
            n += 1
    assert n > 50


def test_vitb_summary_matches_reference(golden_dir):
    g = _load(golden_dir, "vitb_b2_summary.npz")
    cfg = vit_b_config()
    flat = P.recipe_flat(cfg, int(g["seed"]))
    assert hashlib.sha256(flat.tobytes()).hexdigest() == str(g["weights_sha256"])
    sd = P.state_dict_from_flat(cfg, flat)
    px, labels = synthetic_batch(cfg, int(g["batch"]), int(g["label_len"]), seed=int(g["data_seed"]),
                                 min_chars=3, max_chars=int(g["label_len"]))
    r = O.forward_backward(cfg, sd, px, labels, want_stages=True)
    assert abs(r["loss"] - float(g["loss"])) < 2e-5
    idx = g["logit_idx"]
    np.testing.assert_allclose(r["logits"][idx[:, 0], idx[:, 1], idx[:, 2]], g["logit_val"], atol=5e-5)
    assert np.array_equal(r["logits"].argmax(-1), g["argmax"])
    np.testing.assert_allclose(r["stages"]["enc_out"][:, ::16, ::32], g["enc_out_sample"], atol=5e-5)
    norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
    for name, gr in r["grads"].items():
        if gr is None:
            continue
        want = norms[name]
        got = float(np.sqrt((gr.astype(np.float64) ** 2).sum()))
        assert abs(got - want) <= 1e-3 * want + 1e-7, (name, got, want)


def test_cer_kat(golden_dir):
    g = _load(golden_dir, "cer_kat.npz")
    for p, t, c in zip(g["preds"], g["targets"], g["cer"]):
        assert O.calculate_cer(str(p), str(t)) == pytest.approx(float(c))
    # tests/test_ocr_model.py:129-147 of the reference: corpus CER 2/9 for these pairs
    preds, tgts = ["ac", "cot", "test"], ["ab", "cat", "test"]
    assert sum(O.levenshtein(p, t) for p, t in zip(preds, tgts)) / sum(map(len, tgts)) == pytest.approx(2 / 9)
