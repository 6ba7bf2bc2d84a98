"""dev: N2 throughput -- the reference's PIL transform on one host core vs the device pipeline, 256 crops of ~100x1000."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import numpy as np, torch
from PIL import Image
from kzv.data import image_to_tensor, resize_with_padding
from kzv.preprocess import DevicePreprocessor, plan_line
rng = np.random.default_rng(0)
crops = [rng.integers(0, 256, (int(rng.integers(70, 140)), int(rng.integers(700, 1400)), 3), dtype=np.uint8) for _ in range(256)]
t0 = time.perf_counter()
for c in crops[:64]:
    image_to_tensor(resize_with_padding(Image.fromarray(c), (64, 640)))
host = (time.perf_counter() - t0) / 64
print(f"host PIL transform: {host * 1e3:.2f} ms per crop on one core -> {1 / host:.0f} img/s/core")
t0 = time.perf_counter()
plans = [plan_line(c.shape[0], c.shape[1], 64, 640) for c in crops]
print(f"host plans (geometry + Lanczos tables, first sight of each size): {(time.perf_counter() - t0) / 256 * 1e3:.3f} ms per crop")
pre = DevicePreprocessor((64, 640))
pre(crops, plans); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): out = pre(crops, plans)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"device pipeline (pack + H2D + 2 kernels): {dt * 1e3:.1f} ms per batch of 256 -> {256 / dt:.0f} img/s")
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
from kzv import _lib as L
import ctypes as C
# kernels only: reuse the staged buffers of the last call
d_rgb, d_coef, d_desc, d_tmp = pre._keep
mt = max(p["in_h"] * p["new_w"] for p in plans)
e0.record()
for _ in range(20):
    L.check(L.load().kzv_preprocess_lines(d_rgb.data_ptr(), d_desc.data_ptr(), d_coef.data_ptr(), 256, 64, 640, mt, pre.lut.data_ptr(),
                                           d_tmp.data_ptr(), out.data_ptr(), L.stream_handle()))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
byts = d_rgb.numel() + 2 * d_tmp.numel() + out.numel() * 4
print(f"kernels only: {us:.0f} us per batch ({byts / 1e6:.0f} MB touched -> {byts / us / 1e6:.2f} TB/s)")
