"""dev: two same-seed training steps at the bench shape: which gradient tensors differ, by how much (KZV_DEC_CHAIN = 0 / 1 / 2)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv.config import vit_b_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
cfg = vit_b_config(dec_layers=6)
with tempfile.TemporaryDirectory() as tmp:
    m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), init_seed=42, load_tokenizer=False)
px, lab = synthetic_batch(cfg, 256, 128, seed=1)
pxt, ids = torch.from_numpy(px).cuda(), torch.from_numpy(lab).cuda()
m.train()
gs = []
for i in range(0 if os.environ.get("R5_LOGITS_ONLY") else 3):
    loss, _ = m.forward_loss(pxt, ids, seed=11); m.backward(); torch.cuda.synchronize()
    gs.append(({k: v.clone() for k, v in m.grad_dict().items()}, float(loss)))
if gs: gmax = max(float(v.abs().max()) for v in gs[0][0].values())
if gs: print("mode", os.environ.get("KZV_DEC_CHAIN"), "losses", [g[1] for g in gs], "gmax", gmax)
for j in ((1, 2) if gs else ()):
    diffs = sorted(((float((gs[0][0][k] - gs[j][0][k]).abs().max()), k) for k in gs[0][0]), reverse=True)[:6]
    print(f"run {j} vs 0:", [(f"{d:.2e}", k) for d, k in diffs])
zs = []
for i in range(6):
    loss, z = m.forward_loss(pxt, ids, want_logits=True, seed=11); torch.cuda.synchronize(); zs.append(z.clone())
print("logits (bench shape, GEMM + ce path) differing from run 0:", [int(not torch.equal(zs[0], z)) for z in zs], "max", max(float((zs[0] - z).abs().max()) for z in zs))
