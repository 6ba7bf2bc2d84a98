"""BASELINE.json configs[4], first half: variable-width line crops in width buckets (384-1024 px at H = 64 -> S_e = 97..257
patch tokens) on ONE model.  Beyond the reference, whose model instance has one image size
(src/models/trocr_model.py:83-86,113-115): the engine is created for the widest bucket and a narrower batch takes the position
rows of the same (h, w) grid cells, which for the sin/cos table (:154-167) is exactly the closed-form table of the narrow grid.
Parity per bucket: the oracle at that bucket's geometry with the gathered position rows."""
import dataclasses

import numpy as np
import pytest
import torch

from kzv import params as P
from kzv.config import tiny_config, vit_b_config
from kzv.data import BucketBatchSampler, SyntheticLineDataset, build_decoder_dir, make_loader, synthetic_batch
from kzv.model import TrOCRModel
from oracle import trocr_oracle as O

gpu = pytest.mark.gpu


def _no_dropout(cfg):
    return dataclasses.replace(cfg, enc_hidden_dropout=0.0, enc_attn_dropout=0.0, dec_hidden_dropout=0.0, dec_attn_dropout=0.0)


def _rows(cfg, w):
    gh, gw, gmax = cfg.grid_h, w // cfg.patch_w, cfg.grid_w
    return np.array([0] + [1 + h * gmax + x for h in range(gh) for x in range(gw)])


def _bucket_case(cfg, tmp_path, widths, B, Lh, seed):
    d = build_decoder_dir(str(tmp_path / f"dec{cfg.enc_hidden}"), cfg)
    m = TrOCRModel(cfg.encoder_config_dict(), d, init_seed=seed, load_tokenizer=False, width_buckets=widths)
    sd = P.state_dict_from_flat(cfg, P.recipe_flat(cfg, seed))
    m.train()
    for w in widths:
        cw = dataclasses.replace(cfg, image_w=w)
        rows = _rows(cfg, w)
        # the gathered rows ARE the closed-form sin/cos table of the narrow grid (trocr_model.py:154-167)
        assert np.array_equal(sd["encoder.position_embeddings"][0, rows], P.position_table(cw))
        sdw = dict(sd)
        sdw["encoder.position_embeddings"] = sd["encoder.position_embeddings"][:, rows]
        px, lab = synthetic_batch(cw, B, Lh, seed=w, min_chars=3, max_chars=Lh - 1)
        out = m(torch.from_numpy(px), torch.from_numpy(lab))
        m.backward()
        torch.cuda.synchronize()
        r = O.forward_backward(cw, sdw, px, lab)
        err = float(np.abs(out["logits"].cpu().numpy() - r["logits"]).max())
        print(f"width {w}: S_e = {cw.enc_seq}, max|dlogit| = {err:.4g}")
        assert err < 3e-2 and abs(float(out["loss"]) - r["loss"]) < 5e-3
        g = m.grad_dict()
        for k, v in r["grads"].items():
            if v is None or k.endswith("key.bias"):
                continue
            got = g[k].cpu().numpy()
            if k == "encoder.position_embeddings":
                full = got.reshape(1, cfg.enc_seq, cfg.enc_hidden)
                rest = np.delete(full, rows, axis=1)
                assert rest.size == 0 or np.abs(rest).max() == 0.0            # rows of cells outside the narrow grid get no gradient
                got = full[:, rows]
            got = got.reshape(v.shape)
            assert np.abs(got - v).max() < 0.05 * np.abs(v).max() + 1e-7, (w, k)
    with pytest.raises(ValueError, match="doesn't match model"):
        m(torch.zeros(1, 3, cfg.image_h, cfg.image_w - cfg.patch_w), torch.ones(1, 8, dtype=torch.int64))      # not a bucket
    return m


@gpu
def test_tiny_width_buckets_match_oracle(tmp_path):
    cfg = _no_dropout(dataclasses.replace(tiny_config(), image_w=96))       # grid 2 x 6; buckets 2 x {2, 4, 6}
    _bucket_case(cfg, tmp_path, (32, 64, 96), B=3, Lh=14, seed=5)


@gpu
def test_config4_geometry_buckets_384_to_1024(tmp_path):
    """64 x {384, 640, 1024} crops on a ViT-B-wide encoder (768 / 12 heads; 2 of its 12 layers so that the CPU oracle stays in
    seconds) + a 2-layer decoder: 97, 161 and 257 tokens through the <= 192- and <= 288-token attention kernels."""
    cfg = _no_dropout(dataclasses.replace(vit_b_config(dec_layers=2), image_w=1024, enc_layers=2))
    m = _bucket_case(cfg, tmp_path, (384, 640, 1024), B=2, Lh=20, seed=9)
    # greedy decoding on a narrow batch (cross-attention over 96 keys) agrees between the cached and the recompute forms
    m.eval()
    px, _ = synthetic_batch(dataclasses.replace(cfg, image_w=384), 2, 20, seed=3)
    g1 = m.generate(torch.from_numpy(px), max_length=8, use_cache=True).cpu()
    g0 = m.generate(torch.from_numpy(px), max_length=8, use_cache=False).cpu()
    w = min(g0.shape[1], g1.shape[1])
    assert float((g0[:, :w] == g1[:, :w]).float().mean()) > 0.8


@gpu
def test_bucketed_training_loop(tmp_path):
    """SyntheticLineDataset(width_buckets) + BucketBatchSampler + fit(): every batch holds one width, the loss goes down."""
    from kzv.trainer import fit
    cfg = dataclasses.replace(tiny_config(), image_w=96)
    d = build_decoder_dir(str(tmp_path / "dec"), cfg)
    m = TrOCRModel(cfg.encoder_config_dict(), d, learning_rate=5e-3, beta2=0.99, init_seed=1, load_tokenizer=False, width_buckets=(32, 64, 96))
    ds = SyntheticLineDataset(cfg, 24, 12, seed=3, width_buckets=(32, 64, 96))
    loader = make_loader(ds, 4, True, seed=1)
    seen = set()
    for b in loader:
        seen.add(b["pixel_values"].shape[-1])
        assert b["pixel_values"].shape[0] <= 4
    assert seen == {32, 64, 96}
    hist = fit(m, loader, None, max_epochs=12, log_every=1, log=lambda *_: None)
    losses = [v for _, v in hist]
    assert np.isfinite(losses).all() and np.mean(losses[-6:]) < np.mean(losses[:6]) - 0.3


def test_bucket_sampler_properties():
    widths = [384] * 5 + [640] * 7 + [1024] * 3
    for world in (1, 2):
        per_rank = [list(BucketBatchSampler(widths, 4, True, 7, r, world)) for r in range(world)]
        assert len({len(b) for b in per_rank}) == 1                      # equal number of batches on every rank
        flat = [i for bl in per_rank for b in bl for i in b]
        assert set(flat) == set(range(15))
        for bl in per_rank:
            for b in bl:
                assert len({widths[i] for i in b}) == 1 and 1 <= len(b) <= 4
    s = BucketBatchSampler(widths, 4, True, 7)
    a = list(s); s.set_epoch(1); b = list(s)
    assert a != b
