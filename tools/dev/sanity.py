"""dev: odd shapes through the whole engine vs the oracle (GPU)."""
import os, sys, tempfile, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import numpy as np, torch
from kzv import params as P
from kzv.config import tiny_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
from oracle import trocr_oracle as O
base = dataclasses.replace(tiny_config(), enc_hidden_dropout=0, enc_attn_dropout=0, dec_hidden_dropout=0, dec_attn_dropout=0)
for (B, L, kw) in [(1, 2, {}), (7, 5, {}), (3, 39, {}), (5, 17, dict(enc_hidden=64, enc_heads=1, dec_hidden=64)),
                   (2, 12, dict(image_h=16, image_w=16)), (2, 12, dict(enc_layers=1, dec_layers=1)), (300, 9, {})]:
    cfg = dataclasses.replace(base, **kw)
    with tempfile.TemporaryDirectory() as tmp:
        m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), init_seed=3, load_tokenizer=False)
    px, lab = synthetic_batch(cfg, B, L, seed=B, min_chars=1, max_chars=L)
    m.train()
    out = m(torch.from_numpy(px), torch.from_numpy(lab)); m.backward(); torch.cuda.synchronize()
    r = O.forward_backward(cfg, P.state_dict_from_flat(cfg, P.recipe_flat(cfg, 3)), px, lab)
    e = np.abs(out["logits"].cpu().numpy() - r["logits"]).max()
    g = m.grad_dict(); worst = 0.0
    for k, v in r["grads"].items():
        if v is None or k.endswith("key.bias"): continue
        worst = max(worst, float(np.abs(g[k].cpu().numpy().reshape(v.shape) - v).max() / (np.abs(v).max() + 1e-9)))
    print(f"B={B} L={L} {kw}: logit err {e:.2e} loss err {abs(float(out['loss'])-r['loss']):.1e} worst rel grad err {worst:.3f} has_proj={cfg.has_proj}")
