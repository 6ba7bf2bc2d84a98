"""Parity of the parts of the TIMED configuration that round 1 left unchecked:

  (a) the clip + RAdamScheduleFree kernels (csrc/optim.hip) against the oracle's fp64 restatement
      (scripts/train_trocr.py:175, src/models/trocr_model.py:412-451);
  (b) dropout ON: the masks every fused epilogue draws are fetched from the library (kzv_debug_dropout_mask) and replayed
      through the oracle -- logits, loss and EVERY gradient must agree, which is the only thing that verifies the masks
      regenerated in backward (LayerNorm backward, column sums, embedding backward, attention backward);
  (c) a 10-step trajectory of the whole step (forward, backward, clip 1.0, optimizer) against the oracle.
"""
import dataclasses
import types

import numpy as np
import pytest
import torch

from kzv import _lib as L
from kzv import params as P
from kzv.config import tiny_config, vit_b_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
from kzv.optim import RAdamScheduleFree
from oracle import trocr_oracle as O

from _replay import step_masks

pytestmark = pytest.mark.gpu
LOGIT_TOL = 3e-2


def _make(cfg, tmp_path, seed=42):
    d = build_decoder_dir(str(tmp_path / f"dec{cfg.vocab}"), cfg)
    return TrOCRModel(cfg.encoder_config_dict(), d, init_seed=seed, load_tokenizer=False)


# ------------------------------------------------------------------------------------------------ (a) optimizer
class _FlatModel:
    """What kzv.optim.RAdamScheduleFree needs of a model: the flat buffers."""

    def __init__(self, p):
        self.flat_params = p
        self.flat_grads = torch.zeros_like(p)

    def sync_weights(self):
        pass

    def zero_grad(self):
        self.flat_grads.zero_()


@pytest.mark.parametrize("max_norm,grad_scale,wd,gmag", [
    (1.0, 1.0, 0.0, 3e-3),        # the benchmark's setting; ||g|| ~ 3 -> clip ACTIVE
    (1.0, 1.0, 0.0, 1e-5),        # ||g|| ~ 0.01 -> clip inactive
    (1.0, 0.125, 0.01, 3e-2),     # DDP mean over 8 ranks (grad_scale = 1/world), weight decay, clip active after scaling
    (0.0, 0.125, 0.01, 1e-3),     # clipping disabled
])
def test_clip_and_radam_schedulefree_kernels_match_oracle(max_norm, grad_scale, wd, gmag):
    n = 1_000_448                 # multiple of 64 like the engine's flat buffer
    rng = np.random.default_rng(5)
    p0 = (rng.standard_normal(n) * 0.02).astype(np.float32)
    fm = _FlatModel(torch.from_numpy(p0).cuda())
    opt = RAdamScheduleFree(fm, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    st = O.RAdamScheduleFreeState(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=wd)
    y = p0.astype(np.float64); z = y.copy(); v = np.zeros(n)
    seen = set()
    for step in range(12):        # steps 1-5 are the silent phase (rho_t <= 4, lr = 0), 6-12 adaptive
        g = (rng.standard_normal(n) * gmag * (1 + step % 3)).astype(np.float32)
        fm.flat_grads.copy_(torch.from_numpy(g))
        opt.step(max_grad_norm=max_norm, grad_scale=grad_scale)
        seen.add(opt.scheduled_lr > 0)
        # oracle: DDP mean (scale), then clip_grad_norm_ on the averaged gradient, then the optimizer (fp64)
        gs = g.astype(np.float64) * grad_scale
        if max_norm > 0:
            total, coef = O.clip_grad_norm([gs], max_norm)
            assert abs(opt.grad_norm() - total) <= 1e-5 * total
            gs = gs * coef
            seen.add(("clip", coef < 1.0))
        O.radam_schedulefree_step(st, y, z, v, gs)
        got = {"p": fm.flat_params, "z": opt.z, "v": opt.v}
        for name, want in (("p", y), ("z", z), ("v", v)):
            d = np.abs(got[name].double().cpu().numpy() - want)
            tol = 1e-5 * np.abs(want) + (1e-8 if name != "v" else 1e-14)
            assert (d <= tol).all(), (step, name, float((d / (np.abs(want) + 1e-30)).max()))
    assert seen >= {True, False}   # both the silent and the adaptive branch ran
    # optimizer.eval() / .train() swap (trocr_model.py:423-451)
    opt.eval()
    x = O.to_eval(y, z, 0.9)
    assert np.abs(fm.flat_params.double().cpu().numpy() - x).max() <= 1e-5 * np.abs(x).max()
    with pytest.raises(RuntimeError):
        opt.step()
    opt.train()
    assert np.abs(fm.flat_params.double().cpu().numpy() - O.to_train(x, z, 0.9)).max() <= 1e-5 * np.abs(y).max()


# ------------------------------------------------------------------------------------------------ (b) mask replay
def test_debug_mask_statistics_and_determinism():
    lib = L.load()
    a = torch.empty(512, 768, device="cuda"); b = torch.empty(512, 768, device="cuda")
    key = lib.kzv_drop_key(7, 18)
    L.check(lib.kzv_debug_dropout_mask(key, 0.1, 512, 768, 768, a.data_ptr(), L.stream_handle()), "mask")
    L.check(lib.kzv_debug_dropout_mask(key, 0.1, 512, 768, 768, b.data_ptr(), L.stream_handle()), "mask")
    assert torch.equal(a, b)
    vals = torch.unique(a).cpu().numpy()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 65536.0 / (65536 - 6554)) < 1e-6
    assert abs(float((a == 0).float().mean()) - 0.1) < 3e-3
    L.check(lib.kzv_debug_dropout_mask(lib.kzv_drop_key(8, 18), 0.1, 512, 768, 768, b.data_ptr(), L.stream_handle()), "mask")
    assert not torch.equal(a, b)       # another step seed, another mask


def _replay_case(cfg, tmp_path, B, Lh, seed, model_seed, grad_tol):
    m = _make(cfg, tmp_path, model_seed)
    px, lab = synthetic_batch(cfg, B, Lh, seed=3, min_chars=3, max_chars=Lh - 1)
    m.train()
    loss, logits = m.forward_loss(torch.from_numpy(px), torch.from_numpy(lab), want_logits=True, seed=seed)
    m.backward()
    torch.cuda.synchronize()
    masks = step_masks(cfg, seed, B, Lh - 1)
    assert len(masks) == 1 + 3 * cfg.enc_layers + 1 + 5 * cfg.dec_layers
    sd = P.state_dict_from_flat(cfg, P.recipe_flat(cfg, model_seed))
    r = O.forward_backward(cfg, sd, px, lab, masks=masks)
    r_eval = O.forward(cfg, O.leaf_state_dict(sd, requires_grad=False), torch.from_numpy(px), torch.from_numpy(lab))
    got = logits.cpu().numpy()
    err = float(np.abs(got - r["logits"]).max())
    off = float(np.abs(r["logits"] - r_eval[0].numpy()).max())
    print(f"replay: max|dlogit|={err:.4g} (dropout moves the logits by {off:.3g}); dloss={abs(float(loss.item()) - r['loss']):.3g}")
    assert off > 10 * LOGIT_TOL / 3        # the masks matter: without them the comparison below could not pass
    assert err < LOGIT_TOL
    assert abs(float(loss.item()) - r["loss"]) < 5e-3
    g = m.grad_dict()
    worst = {}
    for k, want in r["grads"].items():
        if want is None or k.endswith("key.bias"):      # exactly-zero true gradient (softmax shift invariance)
            continue
        have = g[k].cpu().numpy().reshape(want.shape)
        worst[k] = float(np.abs(have - want).max() / (np.abs(want).max() + 1e-9))
    bad = {k: e for k, e in worst.items() if e > grad_tol}
    print(f"replay: worst relative gradient error {max(worst.values()):.4g}, median {np.median(list(worst.values())):.4g}")
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    return worst


def test_dropout_mask_replay_tiny(tmp_path):
    """tiny geometry, p = 0.1 at every site, B = 3: forward AND backward with the HIP step's own masks."""
    worst = _replay_case(tiny_config(), tmp_path, B=3, Lh=24, seed=12345, model_seed=42, grad_tol=0.05)
    assert np.median(list(worst.values())) < 0.02


def test_dropout_mask_replay_vit_b(tmp_path):
    """BASELINE.json configs[1] geometry (ViT-B/16 + 6-layer decoder, dropout 0.1 as timed), B = 2."""
    _replay_case(vit_b_config(dec_layers=6), tmp_path, B=2, Lh=40, seed=777, model_seed=42, grad_tol=0.06)


def test_dropout_replay_with_unequal_probabilities(tmp_path):
    """hidden and attention dropout use different thresholds per site family; a wrong site/probability pairing in any
    backward kernel shows up here."""
    cfg = dataclasses.replace(tiny_config(), enc_hidden_dropout=0.2, enc_attn_dropout=0.05, dec_hidden_dropout=0.15,
                              dec_attn_dropout=0.3)
    _replay_case(cfg, tmp_path, B=4, Lh=17, seed=99, model_seed=7, grad_tol=0.05)


# ------------------------------------------------------------------------------------------------ (c) trajectory
def test_twenty_step_trajectory_matches_oracle(tmp_path):
    """20 full steps (dropout off, gradient_clip_val 1.0, RAdamScheduleFree) on two alternating batches: the HIP parameters
    against the oracle's (fp32 forward/backward, fp64 optimizer).  scripts/train_trocr.py:165-176, trocr_model.py:412-421.
    The first steps are the optimizer's silent phase (parameters stay put, v accumulates), the rest move them."""
    from kzv.trainer import Stepper
    cfg = dataclasses.replace(tiny_config(), enc_hidden_dropout=0.0, enc_attn_dropout=0.0, dec_hidden_dropout=0.0,
                              dec_attn_dropout=0.0)
    lr, betas = 5e-3, (0.9, 0.99)       # beta2 0.99: RAdam's rectification leaves the silent phase at step 6 and grows quickly
    m = _make(cfg, tmp_path, 4)
    m.hparams.learning_rate, m.hparams.beta2 = lr, betas[1]
    opt = m.configure_optimizers()
    stepper = Stepper(m, opt, world=1, max_grad_norm=1.0)
    sd0 = {k: v.copy() for k, v in P.state_dict_from_flat(cfg, P.recipe_flat(cfg, 4)).items()
           if not k.startswith("decoder.lm_head.decoder.")}
    y = {k: v.astype(np.float64) for k, v in sd0.items()}
    z = {k: v.copy() for k, v in y.items()}
    vv = {k: np.zeros_like(v) for k, v in y.items()}
    states = {k: O.RAdamScheduleFreeState(lr=lr, beta1=betas[0], beta2=betas[1]) for k in y}
    m.train()
    hip_losses, ref_losses = [], []
    for step in range(20):
        px, lab = synthetic_batch(cfg, 6, 20, seed=100 + step % 2, min_chars=3, max_chars=19)      # two alternating batches
        loss = stepper.step({"pixel_values": torch.from_numpy(px), "labels": torch.from_numpy(lab)}, step)
        hip_losses.append(float(loss.item()))
        r = O.forward_backward(cfg, {k: v.astype(np.float32) for k, v in y.items()}, px, lab)
        ref_losses.append(r["loss"])
        grads = {k: (r["grads"][k].astype(np.float64) if r["grads"][k] is not None else np.zeros_like(y[k])) for k in y}
        total, coef = O.clip_grad_norm(list(grads.values()), 1.0)
        assert abs(opt.grad_norm() - total) < 0.03 * total
        for k in y:
            O.radam_schedulefree_step(states[k], y[k], z[k], vv[k], grads[k] * coef)
    print("trajectory losses hip", np.round(hip_losses, 4), "oracle", np.round(ref_losses, 4))
    assert np.abs(np.array(hip_losses) - np.array(ref_losses)).max() < 5e-3
    assert ref_losses[-1] < ref_losses[0] - 0.05          # the steps did move the model
    got = m.state_dict()
    num = den = 0.0
    cos = {}
    for k in y:
        if k.endswith("key.bias"):       # true gradient exactly 0: Adam normalises pure rounding noise there
            continue
        d_ref = (y[k] - sd0[k]).ravel()
        d_hip = (got[k].double().cpu().numpy().reshape(y[k].shape) - sd0[k]).ravel()
        num += float(((d_hip - d_ref) ** 2).sum()); den += float((d_ref ** 2).sum())
        cos[k] = float(d_hip @ d_ref / (np.linalg.norm(d_hip) * np.linalg.norm(d_ref) + 1e-30))
    rel = (num / den) ** 0.5
    print(f"trajectory: relative L2 error of the parameter update {rel:.4f}; min cosine {min(cos.values()):.4f}")
    # Adam divides by sqrt(v): elements whose gradient is below the bf16 noise floor take +-lr steps of arbitrary sign, so
    # the update is compared as a whole (L2) and per tensor by direction
    # observed on MI355X: 0.017 / 0.994 -- the bounds are 3x that, not an order of magnitude
    assert rel < 0.05
    assert min(cos.values()) > 0.98, sorted(cos.items(), key=lambda kv: kv[1])[:5]
    # second-moment state (no normalisation): tight
    vh = P.state_dict_from_flat(cfg, opt.v.cpu().numpy())
    for k in ("encoder.encoder.layer.0.intermediate.dense.weight", "decoder.roberta.encoder.layer.1.output.dense.weight",
              "encoder.patch_embeddings.projection.weight"):
        want = vv[k]
        assert np.abs(vh[k].reshape(want.shape) - want).max() < 0.08 * want.max(), k
