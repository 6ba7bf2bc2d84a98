"""N3 side measurement: one optimisation step of kzv.OCRModel at full ResNet34 depth (ocr_lightning/train.py's defaults: batch 16,
max_boxes 50) on crops of 3 x 64 x 512, against the torch restatement (oracle/ocr_oracle.py) on the host cores.
   python tools/dev/ocr_bench.py [--batch 16] [--height 64] [--width 512] [--steps 20] [--cpu-steps 2]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--height", type=int, default=64); ap.add_argument("--width", type=int, default=512)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=3); ap.add_argument("--cpu-steps", type=int, default=2)
ap.add_argument("--max-boxes", type=int, default=50)
a = ap.parse_args()
from kzv.ocr_model import OCRModel
v = "_" + "abcdefghijklmnopqrstuvwxyz0123456789"
c2i, i2c = {ch: i for i, ch in enumerate(v)}, {i: ch for i, ch in enumerate(v)}
g = torch.Generator().manual_seed(0)
B = a.batch
counts = [int(x) for x in torch.randint(1, a.max_boxes + 1, (B,), generator=g)]
gt = torch.full((B, max(counts), 4), -1.0)
for i, n in enumerate(counts):
    gt[i, :n] = torch.rand(n, 4, generator=g)
texts = ["".join(v[1 + int(k)] for k in torch.randint(0, len(v) - 1, (1,), generator=g)) for _ in range(B)]
batch = {"images": torch.rand(B, 3, a.height, a.width, generator=g), "label_texts": texts, "bounding_boxes_batch": gt,
         "target_lengths": [len(t) for t in texts], "bbox_counts": counts, "image_paths": [""] * B}
m = OCRModel(c2i, i2c, learning_rate=1e-4, max_boxes=a.max_boxes, init_seed=1)
dev = {k: (x.cuda() if torch.is_tensor(x) else x) for k, x in batch.items()}
for _ in range(a.warmup):
    m.fit_step(dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    loss = m.fit_step(dev)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(f"kzv.OCRModel (ResNet34 + BiLSTM + CTC + box head), batch {B} of 3 x {a.height} x {a.width}: {dt * 1e3:.2f} ms per optimisation step = {B / dt:.0f} img/s "
      f"(loss {float(loss):.4f})", flush=True)
if a.cpu_steps > 0:
    from oracle.ocr_oracle import OCROracle
    o = OCROracle(len(c2i), 0, max_boxes=a.max_boxes); o.train()
    opt = torch.optim.Adam(o.parameters(), lr=1e-4)
    ts = []
    for _ in range(a.cpu_steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad(); tot, _, _ = o.shared_step(batch, c2i); tot.backward(); opt.step()
        ts.append(time.perf_counter() - t0)
    cpu = min(ts[1:])
    print(f"torch restatement on the host ({torch.get_num_threads()} threads, fp32): {cpu * 1e3:.0f} ms per step = {B / cpu:.1f} img/s -> x{cpu / dt:.0f}")
