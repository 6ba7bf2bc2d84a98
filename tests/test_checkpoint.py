"""Checkpoint interchange: tests/golden/micro_lightning.ckpt was built from the REFERENCE model's own
state_dict / parameter order in Lightning's checkpoint layout with schedulefree's optimizer-state layout
(tools/gen_ckpt_fixture.py, which also checks that engine-written checkpoints load into the reference strictly)."""
import os

import numpy as np
import pytest
import torch

from kzv import checkpoint as CK
from kzv import params as P
from kzv.config import micro_config, tiny_config, vit_b_config

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref_ckpt():
    return torch.load(os.path.join(HERE, "golden", "micro_lightning.ckpt"), map_location="cpu", weights_only=False)


def test_restated_orders_equal_the_reference_checkpoint(ref_ckpt):
    cfg = micro_config()
    keys = [P.canonical_hf_name(k) for k in ref_ckpt["state_dict"]]
    assert keys == CK.reference_state_dict_order(cfg, "hf5")
    assert [P.canonical_hf_name(k) for k in CK.parameter_names_of(ref_ckpt["state_dict"])] == CK.reference_parameter_order(cfg, "hf5")
    assert len(ref_ckpt["optimizer_states"][0]["state"]) == len(CK.reference_parameter_order(cfg, "hf5")) == 58
    for c in (tiny_config(), vit_b_config()):
        for sp in ("hf4", "hf5"):
            order = CK.reference_parameter_order(c, sp)
            assert len(order) == len(set(order)) and set(order) == {h for h, *_ in P.hf_views(c)} - set(P.TIED_ALIASES)


def test_reference_checkpoint_reads_into_the_flat_layout(ref_ckpt):
    cfg = micro_config()
    sd, opt = CK.read_checkpoint(ref_ckpt, cfg)
    assert set(sd) == {h for h, *_ in P.hf_views(cfg)}
    _, total = P.param_offsets(cfg)
    flat = torch.zeros(total)
    views = P.state_dict_from_flat(cfg, flat)
    for k, v in sd.items():
        views[k].copy_(v.reshape(views[k].shape))
    used = torch.zeros(total, dtype=torch.bool)
    for k, v in P.state_dict_from_flat(cfg, used).items():
        v[...] = True
    # the generator stored z = p + 0.01 and exp_avg_sq = p**2 per reference parameter index
    assert torch.allclose(opt["z"][used], flat[used] + 0.01) and torch.allclose(opt["v"][used], flat[used] ** 2)
    assert (opt["k"], opt["train_mode"], opt["betas"], opt["lr_max"]) == (17, True, (0.9, 0.999), 9e-5)
    assert ref_ckpt["hyper_parameters"]["encoder_config"]["hidden_size"] == cfg.enc_hidden


def test_hf4_spelling_and_round_trip(ref_ckpt):
    cfg = micro_config()
    sd, opt = CK.read_checkpoint(ref_ckpt, cfg)
    _, total = P.param_offsets(cfg)
    flat = torch.zeros(total)
    for k, v in P.state_dict_from_flat(cfg, flat).items():
        v.copy_(sd[k].reshape(v.shape))
    for sp in ("hf4", "hf5"):
        ck = CK.build_checkpoint(cfg, flat, ref_ckpt["hyper_parameters"], 3, 17, optimizer=opt, spelling=sp)
        assert set(ck) >= {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "loops", "callbacks", "optimizer_states",
                           "lr_schedulers", "hparams_name", "hyper_parameters"}
        keys = list(ck["state_dict"])
        if sp == "hf5":
            assert keys == list(ref_ckpt["state_dict"])                       # byte-for-byte the reference's key list
            for k in keys:
                assert torch.equal(ck["state_dict"][k], ref_ckpt["state_dict"][k]), k
            for i, st in ref_ckpt["optimizer_states"][0]["state"].items():
                assert torch.equal(ck["optimizer_states"][0]["state"][i]["z"], st["z"])
                assert torch.equal(ck["optimizer_states"][0]["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
            g0, g1 = ck["optimizer_states"][0]["param_groups"][0], ref_ckpt["optimizer_states"][0]["param_groups"][0]
            assert set(g0) == set(g1) and all(g0[k] == g1[k] for k in g1)
        else:
            assert "encoder.encoder.layer.0.attention.attention.query.weight" in keys and "encoder.encoder.layer.0.intermediate.dense.bias" in keys
        w = ck["state_dict"]
        assert w["decoder.lm_head.decoder.weight"].data_ptr() == w["decoder.roberta.embeddings.word_embeddings.weight"].data_ptr()
        assert w["decoder.lm_head.decoder.bias"].data_ptr() == w["decoder.lm_head.bias"].data_ptr()
        # through torch.save / torch.load (storage sharing survives, so the parameter order is recoverable)
        import io
        buf = io.BytesIO()
        torch.save(ck, buf)
        buf.seek(0)
        sd2, opt2 = CK.read_checkpoint(torch.load(buf, map_location="cpu", weights_only=False), cfg)
        assert all(torch.equal(sd2[k], sd[k]) for k in sd)
        assert torch.equal(opt2["z"], opt["z"]) and torch.equal(opt2["v"], opt["v"]) and opt2["k"] == 17


@pytest.mark.gpu
def test_engine_loads_the_reference_checkpoint_and_round_trips(ref_ckpt, tmp_path):
    from kzv.data import build_decoder_dir, synthetic_batch
    from kzv.ema import EMACallback
    from kzv.model import TrOCRModel
    from kzv.trainer import load_checkpoint, save_checkpoint
    cfg = micro_config()
    d = build_decoder_dir(str(tmp_path / "dec"), cfg)
    path = tmp_path / "ref.ckpt"
    torch.save(ref_ckpt, path)
    m = TrOCRModel(ref_ckpt["hyper_parameters"]["encoder_config"], d, init_seed=1, load_tokenizer=False)
    opt = m.configure_optimizers()
    ema = EMACallback(0.99)
    load_checkpoint(m, opt, str(path), callbacks=[ema])
    px, _ = synthetic_batch(cfg, 3, 10, seed=int(ref_ckpt["reference/pixel_seed"]), min_chars=2, max_chars=9)
    lab = ref_ckpt["reference/labels"]
    m.eval()
    out = m(torch.from_numpy(px), lab)
    assert (out["logits"].cpu() - ref_ckpt["reference/logits"]).abs().max().item() < 3e-2
    assert abs(float(out["loss"]) - ref_ckpt["reference/loss"]) < 5e-3
    assert opt.k == 17 and abs(opt.lr_max - 9e-5) < 1e-12
    assert torch.allclose(opt.z[:64], m.flat_params[:64] + 0.01)
    assert torch.allclose(ema.shadow[:64], m.flat_params[:64] * 0.5)
    # save -> fresh model -> identical logits, parameters and optimizer state
    p2 = tmp_path / "mine.ckpt"
    save_checkpoint(m, opt, str(p2), 3, 17, callbacks=[ema])
    m2 = TrOCRModel(ref_ckpt["hyper_parameters"]["encoder_config"], d, init_seed=2, load_tokenizer=False)
    opt2 = m2.configure_optimizers()
    ck2 = load_checkpoint(m2, opt2, str(p2))
    assert torch.equal(m2.flat_params, m.flat_params) and torch.equal(opt2.z, opt.z) and torch.equal(opt2.v, opt.v)
    assert (opt2.k, opt2.lr_max, opt2.weight_sum) == (opt.k, opt.lr_max, opt.weight_sum)
    m2.eval()
    assert torch.equal(m2(torch.from_numpy(px), lab)["logits"], out["logits"])
    assert "ema_shadow" in ck2 and ck2["pytorch-lightning_version"] == CK.PL_VERSION
