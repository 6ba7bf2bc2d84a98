"""dev: per-tensor gradient differences between kzv_set_dec_chain(1) (launches) and (2) (backward segments), small decoder."""
import dataclasses, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv import _lib as L
from kzv.config import tiny_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
lib = L.load()
cfg = dataclasses.replace(tiny_config(), dec_hidden=256, dec_heads=4, dec_ffn=768, dec_layers=int(os.environ.get("R5_LD", "3")))
if os.environ.get("R5_SMALL"):
    from kzv.config import small_config
    cfg = small_config()
with tempfile.TemporaryDirectory() as tmp:
    m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), init_seed=9, load_tokenizer=False)
B, Lh = int(os.environ.get("R5_B", "5")), int(os.environ.get("R5_L", "30"))
px, lab = synthetic_batch(cfg, B, Lh, seed=9, min_chars=1, max_chars=Lh - 2)
pxt, ids = torch.from_numpy(px).cuda(), torch.from_numpy(lab).cuda()
m.train()
gs = []
for mode in (1, 2, 2):
    L.check(lib.kzv_set_dec_chain(mode), "mode"); m.zero_grad(); m.forward_loss(pxt, ids, want_logits=True, seed=11); m.backward(); torch.cuda.synchronize()
    gs.append({k: v.clone() for k, v in m.grad_dict().items()})
for a, b, name in ((0, 1, "launches vs segments"), (1, 2, "segments vs segments")):
    diffs = sorted(((float((gs[a][k] - gs[b][k]).abs().max()), float(gs[a][k].abs().max()), k) for k in gs[a]), reverse=True)[:8]
    print(name, [(f"{d:.2e} of {s:.2e}", k.replace("decoder.roberta.encoder.", "")) for d, s, k in diffs])
