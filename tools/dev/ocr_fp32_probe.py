"""dev: per-parameter gradient error of kzv.OCRModel(precision="fp32") against the float64 oracle, next to torch fp32's own error."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from kzv.ocr_model import OCRModel
from oracle.ocr_oracle import OCROracle
from test_ocr_gpu import _vocab, _batch
blocks = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,4,6,3").split(","))
widths = (64, 128, 256, 512)[:len(blocks)]
c2i, i2c = _vocab(); mb = 6
m = OCRModel(c2i, i2c, learning_rate=1e-3, max_boxes=mb, init_seed=2, precision="fp32", blocks=blocks, widths=widths)
sd0 = {k: v.cpu() for k, v in m.state_dict().items()}
batch = _batch(6, 64, 160, mb, seed=4)
def oracle(dtype):
    o = OCROracle(len(c2i), 0, max_boxes=mb, blocks=blocks, widths=widths); o.load_state_dict(sd0, strict=True); o = o.to(dtype).train()
    b = dict(batch, images=batch["images"].to(dtype), bounding_boxes_batch=batch["bounding_boxes_batch"].to(dtype))
    t, l, r = o.shared_step(b, c2i); t.backward(); return o, float(t.detach())
o, t32 = oracle(torch.float32); o64, t64 = oracle(torch.float64)
m.train(); m.zero_grad(); got = m.training_step(batch, 0); m.backward(); torch.cuda.synchronize()
print("loss", got, t32, t64)
g64 = dict(o64.named_parameters())
for name, p in o.named_parameters():
    if "weight_hh" in name: continue
    g = m.grad(name).cpu().double(); ex = g64[name].grad
    print(f"{name:55s} engine {float((g-ex).norm()/(ex.norm()+1e-300)):.2e}  torch32 {float((p.grad.double()-ex).norm()/(ex.norm()+1e-300)):.2e}")
