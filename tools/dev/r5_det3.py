"""dev: chain-mode logits against the launch-per-operation path (bit-identical by construction) at the bench shape, with the INPUT changing between
steps: a read of a tensor before it is written returns the previous step's value, which repeated inputs hide."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv import _lib as L
from kzv.config import vit_b_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
lib = L.load()
cfg = vit_b_config(dec_layers=int(os.environ.get("R5_LD", "6")))
B = int(os.environ.get("R5_B", "256"))
with tempfile.TemporaryDirectory() as tmp:
    m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), init_seed=42, load_tokenizer=False)
ins = []
for seed in (1, 2):
    px, lab = synthetic_batch(cfg, B, 128, seed=seed)
    ins.append((torch.from_numpy(px).cuda(), torch.from_numpy(lab).cuda()))
m.train()
def logits(i, mode):
    L.check(lib.kzv_set_dec_chain(mode), "mode")
    _, z = m.forward_loss(ins[i][0], ins[i][1], want_logits=True, seed=11); torch.cuda.synchronize(); return z.clone()
ref = [logits(0, 0), logits(1, 0)]
assert torch.equal(ref[0], logits(0, 0))
for mode in (1, 2):
    out = []
    for i in (0, 1, 0, 1, 1, 0, 0, 1):
        z = logits(i, mode); d = (z - ref[i]).abs()
        bad = (d.amax(-1) > 0).reshape(-1).nonzero().flatten()            # row = b * T + t, 64 rows per workgroup
        wgs = torch.unique(bad // 64)
        out.append(f"{bad.numel()} rows / {wgs.numel()} wgs")
    print(f"mode {mode}:", " | ".join(out))
