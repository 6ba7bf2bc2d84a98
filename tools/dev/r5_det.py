"""dev: is the training forward bit-reproducible run to run (benchmark decoder geometry, ragged rows)?"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv.config import small_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
cfg = small_config()
with tempfile.TemporaryDirectory() as tmp:
    m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), init_seed=2, load_tokenizer=False)
px, lab = synthetic_batch(cfg, 15, cfg.max_pos - cfg.pad_id - 1, seed=2, min_chars=1, max_chars=cfg.max_pos - cfg.pad_id - 3)
pxt, ids = torch.from_numpy(px).cuda(), torch.from_numpy(lab).cuda()
m.train()
zs = []
for i in range(6):
    m.zero_grad(); loss, z = m.forward_loss(pxt, ids, want_logits=True, seed=11); m.backward(); torch.cuda.synchronize(); zs.append(z.clone())
print("runs differing from run 0:", [int(not torch.equal(zs[0], z)) for z in zs], "max diff", max(float((zs[0] - z).abs().max()) for z in zs))
