import sys, os, time, tempfile
sys.path.insert(0, "kuzushiji-vision_amd")
import torch, numpy as np
from kzv.config import vit_b_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
cfg = vit_b_config(6)
d = build_decoder_dir(tempfile.mkdtemp(), cfg)
m = TrOCRModel(cfg.encoder_config_dict(), d, init_seed=1, load_tokenizer=False)
m.eval()
for B in (64, 256):
    px, lab = synthetic_batch(cfg, B, 128, seed=3)
    x = torch.from_numpy(px).cuda()
    for beams, ml in ((1, 32), (4, 32), (1, 128), (4, 128)):
        ids = m.generate(x, max_length=ml, num_beams=beams, early_stopping=False)     # first call (re)binds the workspace
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids = m.generate(x, max_length=ml, num_beams=beams, early_stopping=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"B={B} beams={beams} max_length={ml}: {dt*1e3:.0f} ms -> {B/dt:.0f} img/s, out {tuple(ids.shape)}")
