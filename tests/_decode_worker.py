"""Worker for tests/test_model_gpu.py: greedy-decodes a fixed batch on a 256-wide decoder and saves the step logits it saw.
Run twice by the test, with KZV_DECODE_FUSE_LN=1 and =0 (the switch is read once per process)."""
import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
from kzv import _lib as L                      # noqa: E402
from kzv.config import tiny_config             # noqa: E402
from kzv.data import build_decoder_dir, synthetic_batch  # noqa: E402
from kzv.model import TrOCRModel               # noqa: E402

out, tmp = sys.argv[1], sys.argv[2]
cfg = dataclasses.replace(tiny_config(), dec_hidden=256, dec_heads=4, dec_ffn=768, enc_hidden_dropout=0.0, enc_attn_dropout=0.0,
                          dec_hidden_dropout=0.0, dec_attn_dropout=0.0)
m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "dec"), cfg), init_seed=21, load_tokenizer=False)
m.eval()
B, Lh = 5, 12
px, lab = synthetic_batch(cfg, B, Lh, seed=8, min_chars=3, max_chars=10)
ids = torch.from_numpy(lab).cuda()
ids[:, 0] = cfg.bos_id
pxt = torch.from_numpy(px).cuda()
lib = L.load()
m.forward_loss(pxt, ids, want_logits=False, seed=0)
logits = torch.empty(B, cfg.vocab, device="cuda")
valid = torch.zeros(B, Lh, dtype=torch.uint8, device="cuda")
posids = torch.empty(B, dtype=torch.int32, device="cuda")
tok = torch.empty(B, dtype=torch.int64, device="cuda")
seen = []
for t in range(Lh - 1):
    L.check(lib.kzv_decode_prep(ids.data_ptr(), ids.stride(0), t, cfg.pad_id, B, tok.data_ptr(), valid.data_ptr(), Lh, posids.data_ptr(), L.stream_handle()), "prep")
    L.check(lib.kzv_decode_step(m._h, tok.data_ptr(), posids.data_ptr(), t, valid.data_ptr(), Lh, logits.data_ptr(), L.stream_handle()), "step")
    torch.cuda.synchronize()
    seen.append(logits.cpu().numpy().copy())
np.save(out, np.stack(seen))
