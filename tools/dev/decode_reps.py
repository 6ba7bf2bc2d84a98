"""dev: repeated generate() timings, every repetition printed (greedy / beam-4, graph replay)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv.config import vit_b_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
cfg = vit_b_config(dec_layers=6)
with tempfile.TemporaryDirectory() as tmp:
    m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), load_tokenizer=False)
px = torch.from_numpy(synthetic_batch(cfg, 256, 128, seed=1)[0]).cuda()
m.eval()
os.environ["KZV_DECODE_GRAPH"] = "1"
for beams in (1, 4, 1, 4):
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.generate(px, max_length=128, num_beams=beams, early_stopping=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"beams={beams}: " + " ".join(f"{t:.1f}" for t in ts) + " ms", flush=True)
