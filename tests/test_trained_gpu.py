"""north_star: "token argmax and CER bit-identical on fixed seeds".  Random-init logits are flat, so this is tested on
tests/golden/tiny_trained.npz -- the REFERENCE fitted on 8 crops until its logits are peaked (tools/gen_golden_trained.py)
-- through the product's own tokenizer load, ``calculate_cer`` and ``generate``:

  * teacher-forced: every argmax on the fitted batch identical, identical decoded strings, identical CER values
    (src/models/trocr_model.py:292, 350-357, 400-410);
  * greedy ``generate()`` (KV-cached and prefix-recompute) token-for-token against the reference's step-wise forward
    (SURVEY.md H13), on the fitted AND on unseen crops (non-zero CER), and against the oracle's step-wise forward on
    fresh crops;
  * ``validation_step`` / ``test_step`` / ``decode_predictions`` decode through ``forward(labels=None)`` = beam 4,
    max_length 128, early stopping (trocr_model.py:306-316, 346, 373, 457)."""
import numpy as np
import pytest
import torch

from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
from oracle import trocr_oracle as O

from _trained import load, pad_to

pytestmark = pytest.mark.gpu
LOGIT_TOL = 3e-2


@pytest.fixture(scope="module")
def trained(tmp_path_factory):
    g, cfg, sd, data = load()
    d = build_decoder_dir(str(tmp_path_factory.mktemp("dec")), cfg)
    m = TrOCRModel(cfg.encoder_config_dict(), d, load_tokenizer=True)
    m.load_state_dict(sd, strict=True)
    m.eval()
    return g, cfg, sd, data, m


def _tf_strings(m, cfg, argmax, labels):
    tgt = labels[:, 1:]
    ids = [[int(t) for t, gg in zip(argmax[b], tgt[b]) if gg != cfg.pad_id] for b in range(len(labels))]
    return m.tokenizer.batch_decode(ids, skip_special_tokens=True), m.tokenizer.batch_decode(labels, skip_special_tokens=True)


def test_teacher_forced_argmax_strings_and_cer_identical_to_reference(trained):
    g, cfg, sd, data, m = trained
    for tag, (px, lab) in data.items():
        out = m(torch.from_numpy(px), torch.from_numpy(lab))
        logits = out["logits"].cpu().numpy()
        err = float(np.abs(logits - g[f"{tag}/logits"]).max())
        print(f"{tag}: max|dlogit|={err:.4g} (logit range {np.abs(g[f'{tag}/logits']).max():.1f}), dloss={abs(float(out['loss']) - float(g[f'{tag}/loss'])):.3g}")
        # bf16 operands (8 mantissa bits): the absolute tolerance of flat logits, or 4e-3 of the logit range once peaked
        assert err < max(LOGIT_TOL, 4e-3 * float(np.abs(g[f"{tag}/logits"]).max()))
        assert abs(float(out["loss"]) - float(g[f"{tag}/loss"])) < 5e-3 * max(1.0, float(g[f"{tag}/loss"]))
        am = logits.argmax(-1)
        live = lab[:, 1:] != cfg.pad_id
        if tag == "fit":
            assert np.array_equal(am[live], g["fit/argmax"][live])           # 100 % identity where the loss looks
            assert np.array_equal(am, g["fit/argmax"])                       # ... and in fact everywhere
        else:
            decided = g[f"{tag}/top2_gap"] > 2 * err
            assert decided.mean() > 0.97 and np.array_equal(am[decided], g[f"{tag}/argmax"][decided])
        if np.array_equal(am, g[f"{tag}/argmax"]):
            txt, tgt = _tf_strings(m, cfg, am, lab)
            assert txt == [str(s) for s in g[f"{tag}/tf_text"]]
            assert tgt == [str(s) for s in g[f"{tag}/target_text"]]
            assert [m.calculate_cer(p, t) for p, t in zip(txt, tgt)] == [float(c) for c in g[f"{tag}/tf_cer"]]
        else:
            assert tag != "fit"


@pytest.mark.parametrize("use_cache", [True, False])
def test_greedy_generate_equals_reference_stepwise_decoding(trained, use_cache):
    g, cfg, sd, data, m = trained
    Lh = int(g["label_len"])
    for tag, (px, lab) in data.items():
        gen = m.generate(torch.from_numpy(px), max_length=Lh, num_beams=1, use_cache=use_cache).cpu().numpy()
        want = g[f"{tag}/greedy_ids"]
        assert float(g[f"{tag}/greedy_gap"].min()) > 0.1          # every decision of the reference is decided
        assert np.array_equal(pad_to(gen, Lh, cfg.pad_id), want), tag
        txt = m.tokenizer.batch_decode(gen, skip_special_tokens=True)
        tgt = m.tokenizer.batch_decode(lab, skip_special_tokens=True)
        assert txt == [str(s) for s in g[f"{tag}/greedy_text"]]
        assert [m.calculate_cer(p, t) for p, t in zip(txt, tgt)] == [float(c) for c in g[f"{tag}/greedy_cer"]]
    assert max(float(c) for c in g["unseen/greedy_cer"]) > 0     # the unseen crops exercise non-zero CER


def test_greedy_generate_equals_oracle_stepwise_on_fresh_crops(trained):
    g, cfg, sd, data, m = trained
    px, _ = synthetic_batch(cfg, 16, 20, seed=4242, min_chars=3, max_chars=17)
    px[:8] = 0.7 * data["fit"][0] + 0.3 * px[:8]                # perturbed fitted crops: mostly the fitted strings
    want, gaps = O.greedy_stepwise(cfg, O.leaf_state_dict(sd, requires_grad=False), px, 24)
    gen = pad_to(m.generate(torch.from_numpy(px), max_length=24, num_beams=1).cpu().numpy(), 24, cfg.pad_id)
    # rows whose every decision had a top-2 gap above the logit tolerance must match token for token
    sure = gaps.min(axis=1) > 2 * LOGIT_TOL
    print(f"fresh crops: {int(sure.sum())}/16 rows fully decided; smallest gap {gaps.min():.3g}")
    assert sure.sum() >= 8
    assert np.array_equal(gen[sure], want[sure])
    # the others: identical up to the first undecided step
    for b in np.where(~sure)[0]:
        first = int(np.argmax(gaps[b] <= 2 * LOGIT_TOL))
        assert np.array_equal(gen[b, :first + 1], want[b, :first + 1])


def _oracle_beam(cfg, sd, px, max_length=128, early_stopping=True):
    """beam-4 decode with ORACLE step logits through the same bookkeeping (kzv/beam.py, pinned against HF generate in
    tests/test_host_cpu.py)."""
    from kzv import beam as BM
    Lh = min(max_length, cfg.max_pos - cfg.pad_id - 1)
    step = O.stepwise_logits(cfg, O.leaf_state_dict(sd, requires_grad=False), px, repeat=4)
    return BM.beam_search(step, lambda rows, t: None, px.shape[0], 4, Lh, cfg.vocab, cfg.pad_id, cfg.bos_id, cfg.eos_id, "cpu",
                          early_stopping=early_stopping).numpy()


def test_validation_and_test_steps_decode_with_beam_search_like_the_reference(trained):
    """trocr_model.py:346, 373, 457 call self(pixel_values) -> generate(num_beams=4, max_length=128, early_stopping=True).
    Expected ids: the oracle's step logits through the same (HF-pinned) beam bookkeeping."""
    g, cfg, sd, data, m = trained
    px, lab = data["fit"]
    calls = []
    orig = m.generate

    def spy(pixel_values, **kw):
        calls.append(kw)
        return orig(pixel_values, **kw)
    m.generate = spy
    try:
        batch = {"pixel_values": torch.from_numpy(px), "labels": torch.from_numpy(lab)}
        m.logged.clear()
        m.validation_step(batch, 0)
        m.test_step(batch, 0)
        texts = m.decode_predictions(torch.from_numpy(px))
    finally:
        m.generate = orig
    assert len(calls) == 3 and all(kw.get("num_beams") == 4 and kw.get("max_length") == 128 and kw.get("early_stopping") is True for kw in calls)
    want = _oracle_beam(cfg, sd, px)
    want_txt = m.tokenizer.batch_decode(want, skip_special_tokens=True)
    tgt = m.tokenizer.batch_decode(lab, skip_special_tokens=True)
    assert texts == want_txt
    cers = [O.calculate_cer(p, t) for p, t in zip(want_txt, tgt)]
    assert m.logged["val_cer"] == [cers[0]] and m.logged["test_cer"] == [sum(cers) / len(cers)]
    assert abs(m.logged["val_loss"][0] - float(g["fit/loss"])) < 5e-3
    # ids, for both cache modes and for the exhaustive variant (which recovers the fitted labels: CER 0)
    for es in (True, False):
        want = _oracle_beam(cfg, sd, px, max_length=20, early_stopping=es)
        for uc in (True, False):
            got = m.generate(torch.from_numpy(px), max_length=20, num_beams=4, early_stopping=es, use_cache=uc).cpu().numpy()
            assert got.shape == want.shape and np.array_equal(got, want), (es, uc)
    full = m.tokenizer.batch_decode(want, skip_special_tokens=True)
    assert [O.calculate_cer(p, t) for p, t in zip(full, tgt)] == [0.0] * len(tgt)


def test_graph_replayed_decode_step_equals_the_eager_step(trained, monkeypatch):
    """kzv_decode_step_graph (device-side step index, one hipGraph per cache copy) against the eager kzv_decode_step: same
    tokens for greedy and beam-4, on the fitted crops and on flat-logit crops, twice in a row (graph reuse across calls)."""
    g, cfg, sd, data, m = trained
    px = torch.from_numpy(np.concatenate([data["fit"][0], data["unseen"][0]]))
    for beams in (1, 4):
        outs = {}
        for mode in ("0", "1", "1"):
            monkeypatch.setenv("KZV_DECODE_GRAPH", mode)
            outs.setdefault(mode, []).append(m.generate(px, max_length=20, num_beams=beams, early_stopping=False).cpu().numpy())
        assert np.array_equal(outs["1"][0], outs["1"][1])
        assert outs["0"][0].shape == outs["1"][0].shape and np.array_equal(outs["0"][0], outs["1"][0]), beams


def test_decode_graph_does_not_outlive_a_rebind(trained, monkeypatch):
    """A captured decode step bakes in workspace-internal pointers.  generate at (B, L), then a LARGER bind (the host mirror
    reallocates its workspace on growth), then generate at the first shape again: the replayed step must be re-captured, not
    replayed against the freed workspace (kzv_model_bind drops the graphs).  Checked against the eager step."""
    g, cfg, sd, data, m = trained
    px = torch.from_numpy(data["fit"][0])
    monkeypatch.setenv("KZV_DECODE_GRAPH", "1")
    first = m.generate(px, max_length=20, num_beams=1).cpu().numpy()
    ws_before = m._ws.data_ptr()
    big = torch.from_numpy(np.concatenate([data["fit"][0]] * 4 + [data["unseen"][0]]))
    m.generate(big, max_length=24, num_beams=4, early_stopping=False)           # grows the workspace: new allocation
    junk = torch.full((m._ws.numel() // 4,), float("nan"), device=m.device)      # whatever the allocator recycles is poisoned
    del junk
    again = m.generate(px, max_length=20, num_beams=1).cpu().numpy()
    monkeypatch.setenv("KZV_DECODE_GRAPH", "0")
    eager = m.generate(px, max_length=20, num_beams=1).cpu().numpy()
    assert m._ws.data_ptr() != ws_before or m._ws.numel() > 0
    assert np.array_equal(first, again) and np.array_equal(again, eager)
