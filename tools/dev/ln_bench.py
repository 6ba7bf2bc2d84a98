"""dev: time layernorm backward at the ViT / decoder shapes of one training step."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load()
def st(): return torch.cuda.current_stream().cuda_stream
def bench(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for rows, H in [(41216, 768), (15360, 256)]:
    x = torch.randn(rows, H, device="cuda"); g = torch.randn(H, device="cuda"); stats = torch.stack([x.mean(1), 1 / x.std(1)], 1).contiguous()
    dx = torch.zeros(rows, H, device="cuda"); dg = torch.zeros(H, device="cuda"); db = torch.zeros(H, device="cuda")
    for f32 in (0, 1):
        dy = torch.randn(rows, H, device="cuda") if f32 else torch.randn(rows, H, device="cuda").bfloat16()
        for acc in (0, 1):
            us = bench(lambda: L.check(lib.kzv_layernorm_bwd(dy.data_ptr(), f32, x.data_ptr(), stats.data_ptr(), g.data_ptr(), dx.data_ptr(), acc,
                                                             dg.data_ptr(), db.data_ptr(), rows, H, st())))
            byts = rows * H * (4 + (4 if f32 else 2) + 4 + (4 if acc else 0))
            print(f"ln_bwd rows {rows} H {H} dy_f32 {f32} acc {acc}: {us:7.1f} us  {byts / us / 1e6:6.2f} TB/s (incl. reduce kernel)")
