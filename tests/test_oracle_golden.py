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


def test_cer_independent_known_answers(golden_dir):
    """Edit distances a reader can check by hand, and the CER values the REFERENCE's calculate_cer returned for them
    with a Wagner-Fischer matrix bound as ``editdistance`` (tools/gen_golden_trained.py: a different algorithm from the
    two-row scans of the oracle and of kzv/model.py, itself checked against the recursive definition)."""
    from kzv.model import _levenshtein
    hand = {("kitten", "sitting"): 3, ("flaw", "lawn"): 2, ("intention", "execution"): 5, ("sunday", "saturday"): 3,
            ("ac", "ab"): 1, ("cot", "cat"): 1, ("test", "test"): 0, ("", "abc"): 3, ("abc", ""): 3, ("", ""): 0}
    for (a, b), d in hand.items():
        assert O.levenshtein(a, b) == d and _levenshtein(a, b) == d and O.levenshtein(b, a) == d
    g = _load(golden_dir, "tiny_trained.npz")
    for p, t, c in zip(g["cer/preds"], g["cer/targets"], g["cer/values"]):
        assert O.calculate_cer(str(p), str(t)) == float(c)
    rng = np.random.default_rng(3)
    import functools

    def rec(a, b):
        @functools.lru_cache(maxsize=None)
        def f(i, j):
            if i == 0 or j == 0:
                return i + j
            return min(f(i - 1, j) + 1, f(i, j - 1) + 1, f(i - 1, j - 1) + (a[i - 1] != b[j - 1]))
        return f(len(a), len(b))
    for _ in range(200):
        a = "".join(rng.choice(list("abc"), rng.integers(0, 8)))
        b = "".join(rng.choice(list("abc"), rng.integers(0, 8)))
        assert O.levenshtein(a, b) == rec(a, b) == _levenshtein(a, b)


def test_oracle_reproduces_the_trained_reference(golden_dir):
    """tiny_trained.npz: peaked logits.  The oracle must give the reference's logits, EVERY argmax, the same step-wise
    greedy tokens and (through the repo's tokenizer directory) the same strings and CER."""
    from _trained import load, pad_to
    g, cfg, sd, data = load()
    tsd = O.leaf_state_dict(sd, requires_grad=False)
    for tag, (px, lab) in data.items():
        logits, loss = O.forward(cfg, tsd, torch.from_numpy(px), torch.from_numpy(lab))
        np.testing.assert_allclose(logits.numpy(), g[f"{tag}/logits"], atol=5e-5, rtol=0)
        assert abs(float(loss) - float(g[f"{tag}/loss"])) < 2e-5
        assert np.array_equal(logits.numpy().argmax(-1), g[f"{tag}/argmax"])
        ids, gaps = O.greedy_stepwise(cfg, tsd, px, int(g["label_len"]))
        assert np.array_equal(ids, g[f"{tag}/greedy_ids"])
    assert float(g["fit/top2_gap"][data["fit"][1][:, 1:] != cfg.pad_id].min()) > 5.0      # peaked where it was fitted


def test_oracle_explicit_dropout_masks():
    """oracle masks plumbing: all-ones masks == eval mode; a mask drawn like nn.Dropout reproduces F.dropout's arithmetic."""
    cfg = tiny_config()
    px, lab = synthetic_batch(cfg, 2, 10, seed=3, min_chars=2, max_chars=9)
    sd = O.leaf_state_dict(P.state_dict_from_flat(cfg, P.recipe_flat(cfg, 1)), requires_grad=False)
    base, _ = O.forward(cfg, sd, torch.from_numpy(px), torch.from_numpy(lab))
    B, T, Se = 2, 9, cfg.enc_seq
    shapes = {"enc_emb": (B, Se, cfg.enc_hidden), "dec_emb": (B, T, cfg.dec_hidden)}
    for i in range(cfg.enc_layers):
        shapes[f"enc{i}_attn"] = (B, cfg.enc_heads, Se, Se)
        shapes[f"enc{i}_o"] = shapes[f"enc{i}_mlp"] = (B, Se, cfg.enc_hidden)
    for i in range(cfg.dec_layers):
        shapes[f"dec{i}_sa"] = (B, cfg.dec_heads, T, T)
        shapes[f"dec{i}_ca"] = (B, cfg.dec_heads, T, cfg.num_patches)
        shapes[f"dec{i}_sa_o"] = shapes[f"dec{i}_ca_o"] = shapes[f"dec{i}_ffn"] = (B, T, cfg.dec_hidden)
    ones, _ = O.forward(cfg, sd, torch.from_numpy(px), torch.from_numpy(lab), masks={k: torch.ones(s) for k, s in shapes.items()})
    assert torch.equal(ones, base)
    gen = torch.Generator().manual_seed(0)
    for name, shp in shapes.items():
        mk = (torch.rand(shp, generator=gen) >= 0.5).float() * 2.0
        out, _ = O.forward(cfg, sd, torch.from_numpy(px), torch.from_numpy(lab), masks={name: mk})
        assert float((out - base).abs().max()) > 1e-6, name      # every site is wired


def test_fp8_quantiser_restatement_matches_an_independent_cast():
    """oracle quant_e4m3 (the fp8 weight path's reference arithmetic, restated from the OCP e4m3fn format) against torch's own
    float8_e4m3fn conversion on 200k values over 10 decades plus the format's corner cases."""
    import torch
    torch.manual_seed(0)
    x = torch.randn(200000) * 10 ** torch.empty(200000).uniform_(-5, 3)
    corners = torch.tensor([0.0, -0.0, 448.0, 447.9, 464.0, 500.0, -1e9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9,
                            2.0 ** -6, 0.0156, 0.9375, 1.0625, 15.5, 17.0, 240.0, 416.0, 432.0])
    x = torch.cat([x, corners, -corners])
    want = torch.clamp(x, -448.0, 448.0).to(torch.float8_e4m3fn).float()
    assert torch.equal(O.quant_e4m3(x), want)
    q, s = O.quant_rows_e4m3(torch.tensor([[0.0, 0.0, 0.0, 0.0], [1.0, -2.0, 0.5, 448.0], [3.0, 1.0, -7.0, 0.1]]))
    assert s.reshape(-1).tolist() == [1.0, 1.0, 7.0 / 448.0]
    assert q[0].abs().max() == 0 and q[2].abs().max() == 448.0
    assert O.next_act_qscale(0.0, 8.0) == 8.0 and O.next_act_qscale(3.0) == 64.0 and O.next_act_qscale(448.0) == 0.5


def test_attention_dropout_generator_is_statistically_bernoulli():
    """The 4 x 4-block generator the attention kernels draw their probability dropout from (oracle/attn_dropout.py restates
    csrc/kzv_common.h): marginal rate, joint drop rates of neighbours inside and across blocks, across heads and across site
    keys, and per-row / per-column rates -- all within sampling error of independent Bernoulli(0.1) draws."""
    from oracle import attn_dropout as AD
    p, pairs, S = 0.1, 96, 161
    keys = [0x1234567, 0x9abcdef1]
    d = [~AD.keep_mask(k, p, pairs, S, S) for k in keys]
    t = AD.thr16_of(p)
    assert t == 6554
    pd = t / 65536.0
    n = d[0].size
    for dm in d:
        assert abs(dm.mean() - pd) < 4 * np.sqrt(pd * (1 - pd) / n)
        for dq, dk in [(0, 1), (1, 0), (1, 1), (0, 2), (2, 0), (0, 3), (3, 0), (2, 2), (0, 4), (4, 0), (4, 4), (1, 3), (3, 1), (0, 8), (8, 0)]:
            a, b = dm[:, :S - dq, :S - dk], dm[:, dq:, dk:]
            pj = (a & b).mean()
            assert abs(pj - pd * pd) < 5 * np.sqrt(pd * pd * (1 - pd * pd) / a.size), (dq, dk, pj)
        pj = (dm[:-1] & dm[1:]).mean()                                       # same (q, k) of the next head
        assert abs(pj - pd * pd) < 5 * np.sqrt(pd * pd / dm[1:].size)
        rows, cols = dm.mean(axis=2).ravel(), dm.mean(axis=1).ravel()
        assert abs(rows.std() / np.sqrt(pd * (1 - pd) / S) - 1) < 0.03 and abs(cols.std() / np.sqrt(pd * (1 - pd) / S) - 1) < 0.03
    pj = (d[0] & d[1]).mean()                                                # two sites / steps
    assert abs(pj - pd * pd) < 5 * np.sqrt(pd * pd / n)
    # the multiplier form the debug entry reports, ragged sizes, and p = 0
    m = AD.multiplier(7, 0.1, 3, 5, 7)
    assert m.shape == (15, 7) and set(np.unique(m)).issubset({np.float32(0.0), np.float32(65536.0) / np.float32(65536 - 6554)})
    assert AD.keep_mask(7, 0.0, 2, 3, 3).all()
