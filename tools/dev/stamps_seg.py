"""dev: phase timeline of dec_bwd_seg_kernel<KZV_STAMP_SEG_KS1, KZV_STAMP_SEG_NP2> (variant library built with -DKZV_STAMPS; KZV_LIB points at it):
workgroup 0 of its last launch in one benchmark-size step."""
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
px, lab = synthetic_batch(cfg, 256, 128, seed=1, min_chars=8, max_chars=60)
pxt, ids = torch.from_numpy(px).cuda(), torch.from_numpy(lab).cuda()
m.train()
for _ in range(3):
    m.zero_grad(); m.forward_loss(pxt, ids, want_logits=False, seed=1); m.backward()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 32)()
assert lib.kzv_debug_chain_stamps(buf, 32) == 0
st = list(buf)
names = ["window + load rows", "gemm 1", "d tile", "LayerNorm backward", "partials", "gemm 2 (+ gelu')", "rows out"]
print("dec_bwd_seg, workgroup 0 (cycles of the 100 MHz... s_memtime): " + ", ".join(f"{n} {st[17 + k] - st[16 + k]}" for k, n in enumerate(names)) + f" | total {st[23] - st[16]}")
