"""dev: phase timeline of the one-launch decoder step (variant library built with -DKZV_STAMPS; KZV_LIB points at it)."""
import ctypes, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv.config import vit_b_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
from kzv import _lib
cfg = vit_b_config(dec_layers=6)
with tempfile.TemporaryDirectory() as tmp:
    m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), load_tokenizer=False)
px = torch.from_numpy(synthetic_batch(cfg, 256, 128, seed=1)[0]).cuda()
m.eval()
names = ["qkv", "self", "o", "ln1", "cq", "cross", "co", "ln2", "fc1", "fc2", "ln3"]
lib = ctypes.CDLL(_lib.LIB_PATH)
for beams in (1, 4):
    out = m.generate(px, max_length=int(os.environ.get("DEC_LEN", "128")), num_beams=beams, early_stopping=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 64)()
    assert lib.kzv_debug_decode_stamps(buf, 64) == 0
    st = list(buf)
    print(f"beams={beams}: tokens {out.shape[1]}; whole kernel {st[41] - st[0]} cycles; embed {st[1] - st[0]}; head {st[41] - st[40]}")
    for base, tag in ((2, "layer 0"), (20, "last layer")):
        d = [st[base + k + 1] - st[base + k] for k in range(11)]
        print(f"  {tag}: " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f"  (sum = {sum(d)})")
    a = st[44:52]
    print("  attention wave, layer 1: B1->self done", a[1] - a[0], "rest of sequences", a[2] - a[1], "cross issue", a[3] - a[2], "| B5->cross done", a[5] - a[4],
          "params", a[6] - a[5], "self issue", a[7] - a[6], "| B1 -> B5", a[4] - a[0])
