#!/usr/bin/env python3
"""Generate tests/golden/tiny_trained.npz by TRAINING AND RUNNING THE REFERENCE (build container only).

Random-init logits are nearly flat, so "token argmax and CER bit-identical" (BASELINE.json north_star) cannot be tested
on them.  This script imports the reference ``TrOCRModel`` (same three import shims as tools/gen_golden.py), fits it with
AdamW on ONE fixed batch of 8 synthetic crops (tiny geometry, labels wrapped in BOS ... EOS) until its logits are peaked,
rounds the trained weights to bf16-representable values (so the fixture is half the size and the engine's bf16 weight
copies are exact), reloads them into the reference and records what the REFERENCE then computes:

  * teacher-forced logits / argmax / loss on the fitted batch and on 4 unseen crops,
  * the strings its own tokenizer decodes from the argmax ids and ``TrOCRModel.calculate_cer`` of them,
  * greedy decoding from BOS by step-wise ``forward`` (token t+1 = argmax of position t; SURVEY.md H13: HF ``generate``
    under transformers 5.x is not a usable golden), decoded strings and CER.

``editdistance`` is absent from the container; ``calculate_cer`` is run with a full-matrix Wagner-Fischer bound under
that module name, itself checked here against the recursive definition of the edit distance -- a different algorithm
from the two-row scans in oracle/trocr_oracle.py and kzv/model.py, so the stored CER values pin both independently.

Run:  python -B tools/gen_golden_trained.py        (from the repo root)
"""
from __future__ import annotations

import functools
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import torch  # noqa: E402

from gen_golden import build_reference, install_shims  # noqa: E402
from kzv import params as P  # noqa: E402
from kzv.config import tiny_config  # noqa: E402
from kzv.data import synthetic_batch  # noqa: E402


def wagner_fischer(a, b) -> int:
    d = np.zeros((len(a) + 1, len(b) + 1), dtype=np.int64)
    d[:, 0] = np.arange(len(a) + 1)
    d[0, :] = np.arange(len(b) + 1)
    for i in range(1, len(a) + 1):
        for j in range(1, len(b) + 1):
            d[i, j] = min(d[i - 1, j] + 1, d[i, j - 1] + 1, d[i - 1, j - 1] + (a[i - 1] != b[j - 1]))
    return int(d[len(a), len(b)])


def recursive_distance(a, b) -> int:
    @functools.lru_cache(maxsize=None)
    def f(i, j):
        if i == 0 or j == 0:
            return i + j
        return min(f(i - 1, j) + 1, f(i, j - 1) + 1, f(i - 1, j - 1) + (a[i - 1] != b[j - 1]))
    return f(len(a), len(b))


def wrap_bos_eos(cfg, lab0):
    lab = np.full_like(lab0, cfg.pad_id)
    for b in range(lab0.shape[0]):
        n = int((lab0[b] != cfg.pad_id).sum())
        lab[b, 0] = cfg.bos_id
        lab[b, 1:1 + n] = lab0[b, :n]
        lab[b, 1 + n] = cfg.eos_id
    return lab


def greedy_stepwise(model, cfg, px, Lh):
    """token t+1 = argmax of the teacher-forced logits of position t over the prefix so far (src/models/trocr_model.py:
    258-297 called step by step; under the causal + key-padding mask position t only sees ids[:, :t+1])."""
    B = px.shape[0]
    ids = torch.full((B, Lh), cfg.pad_id, dtype=torch.int64)
    ids[:, 0] = cfg.bos_id
    done = torch.zeros(B, dtype=torch.bool)
    gaps = []
    with torch.no_grad():
        for t in range(Lh - 1):
            lg = model(px, ids)["logits"][:, t]
            top2 = lg.topk(2, dim=-1).values
            gaps.append((top2[:, 0] - top2[:, 1]).masked_fill(done, float("inf")))
            nxt = lg.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, cfg.pad_id), nxt)
            ids[:, t + 1] = nxt
            done |= nxt == cfg.eos_id
            if bool(done.all()):
                break
    return ids.numpy(), torch.stack(gaps, 1).numpy()


def main():
    install_shims()
    rng = np.random.default_rng(0)
    for _ in range(300):          # the CER arithmetic below rests on this implementation: check it against the definition
        a = "".join(rng.choice(list("abcd"), rng.integers(0, 7)))
        b = "".join(rng.choice(list("abcd"), rng.integers(0, 7)))
        assert wagner_fischer(a, b) == recursive_distance(a, b)
    ed = types.ModuleType("editdistance")
    ed.eval = wagner_fischer
    sys.modules["editdistance"] = ed

    cfg = tiny_config()
    B, Lh = 8, 20
    with tempfile.TemporaryDirectory() as tmp:
        model, _ = build_reference(cfg, tmp, seed=42)
        px, lab0 = synthetic_batch(cfg, B, Lh, seed=21, min_chars=3, max_chars=Lh - 3)
        lab = wrap_bos_eos(cfg, lab0)
        px_u, lab0_u = synthetic_batch(cfg, 4, Lh, seed=22, min_chars=3, max_chars=Lh - 3)
        lab_u = wrap_bos_eos(cfg, lab0_u)
        pxt, labt = torch.from_numpy(px), torch.from_numpy(lab)
        model.eval()              # dropout off: a deterministic fit (only the final weights matter)
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.0)
        torch.manual_seed(0)
        for step in range(600):
            opt.zero_grad()
            loss = model(pxt, labt)["loss"]
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            if step % 100 == 0:
                print("fit step", step, float(loss))
        print("fit loss", float(loss))
        # bf16-representable weights, reloaded into the reference
        sd = {k: v.detach().to(torch.bfloat16).float() for k, v in model.state_dict().items()}
        model.load_state_dict(sd, strict=True)
        tok = model.tokenizer
        save = {"fit_seed": 21, "unseen_seed": 22, "label_len": Lh, "labels": lab, "labels_unseen": lab_u}
        for k5, v in sd.items():
            name = P.canonical_hf_name(k5)
            save["w/" + name] = v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
        for tag, x, y in (("fit", px, lab), ("unseen", px_u, lab_u)):
            xt, yt = torch.from_numpy(x), torch.from_numpy(y)
            with torch.no_grad():
                out = model(xt, yt)
            lg = out["logits"].numpy()
            am = lg.argmax(-1)
            srt = np.sort(lg, -1)
            save[f"{tag}/logits"] = lg
            save[f"{tag}/loss"] = np.float64(float(out["loss"]))
            save[f"{tag}/argmax"] = am.astype(np.int32)
            save[f"{tag}/top2_gap"] = srt[..., -1] - srt[..., -2]
            # teacher-forced strings: argmax ids at the positions whose target is not padding
            tgt = y[:, 1:]
            tf_ids = [[int(t) for t, g in zip(am[b], tgt[b]) if g != cfg.pad_id] for b in range(len(y))]
            tf_txt = tok.batch_decode(tf_ids, skip_special_tokens=True)
            tgt_txt = tok.batch_decode(y, skip_special_tokens=True)
            save[f"{tag}/tf_text"] = np.array(tf_txt)
            save[f"{tag}/target_text"] = np.array(tgt_txt)
            save[f"{tag}/tf_cer"] = np.array([model.calculate_cer(p, t) for p, t in zip(tf_txt, tgt_txt)])
            gen, gaps = greedy_stepwise(model, cfg, xt, Lh)
            gen_txt = tok.batch_decode(gen, skip_special_tokens=True)
            save[f"{tag}/greedy_ids"] = gen
            save[f"{tag}/greedy_gap"] = gaps
            save[f"{tag}/greedy_text"] = np.array(gen_txt)
            save[f"{tag}/greedy_cer"] = np.array([model.calculate_cer(p, t) for p, t in zip(gen_txt, tgt_txt)])
            print(tag, "loss", float(out["loss"]), "min top-2 gap (non-pad targets)", float(save[f"{tag}/top2_gap"][tgt != cfg.pad_id].min()),
                  "tf cer", save[f"{tag}/tf_cer"], "greedy cer", save[f"{tag}/greedy_cer"], "min greedy gap", float(gaps.min()))
        # CER known answers through the reference's method with hand-checkable pairs (textbook distances in the comment)
        pairs = [("kitten", "sitting"),      # 3
                 ("flaw", "lawn"),           # 2
                 ("intention", "execution"),  # 5
                 ("sunday", "saturday"),     # 3
                 ("ac", "ab"), ("cot", "cat"), ("test", "test"),   # tests/test_ocr_model.py:129-147 samples: 1, 1, 0
                 ("", "abc"), ("abc", ""), ("", "")]
        save["cer/preds"] = np.array([p for p, _ in pairs])
        save["cer/targets"] = np.array([t for _, t in pairs])
        save["cer/values"] = np.array([model.calculate_cer(p, t) for p, t in pairs])
        print("cer", save["cer/values"])
        out_path = os.path.join(ROOT, "tests", "golden", "tiny_trained.npz")
        np.savez_compressed(out_path, **save)
        print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
