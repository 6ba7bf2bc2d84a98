import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from kzv.ocr_model import OCRModel
from oracle.ocr_oracle import OCROracle
from test_ocr_gpu import _vocab, _batch
c2i, i2c = _vocab(); mb = 4
m = OCRModel(c2i, i2c, learning_rate=1e-3, max_boxes=mb, blocks=(1, 1), widths=(64, 128), init_seed=3)
sd0 = {k: v.cpu() for k, v in m.state_dict().items()}
batch = _batch(6, 64, 96, mb, seed=9)
m.train(); m.zero_grad(); got = m.training_step(batch, 0); masks = m.relu_masks_of_last_step(); m.backward(); torch.cuda.synchronize()
o2 = OCROracle(len(c2i), 0, max_boxes=mb, blocks=(1, 1), widths=(64, 128)); o2.load_state_dict(sd0, strict=True); o2.train()
o2.relu_masks.extend(masks)
t2, _, _ = o2.shared_step(batch, c2i); t2.backward()
for name, p in o2.named_parameters():
    if "weight_hh" in name or "rnn" in name: continue
    g, want = m.grad(name).cpu(), p.grad
    print(f"{name:45s} maxerr {(g - want).abs().max().item() / max(want.abs().max().item(), 1e-12):7.4f} relL2 {float((g-want).norm()/want.norm()):7.4f}")
