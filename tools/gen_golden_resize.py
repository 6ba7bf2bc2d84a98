"""Golden vectors for the N2 image transform, produced by the reference's own dependency stack: Pillow's
Image.resize(LANCZOS) + paste on white (ResizeWithPadding, src/data/trocr_dataset.py:24-53) and torch's ToTensor /
Normalize(0.5, 0.5) arithmetic (:100-102; torchvision itself is not installed in this image, its two ops are
`uint8 -> float32 / 255` and `(x - mean) / std`).  Run here (Pillow 12.2.0):  python tools/gen_golden_resize.py
Writes tests/golden/resize_kat.npz: inputs (seeded random and structured crops of assorted sizes), the uint8 canvases for two
target sizes, the fp32 tensors for the small one and the 256-entry uint8 -> fp32 table of ToTensor + Normalize."""
import os

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_transform(img: np.ndarray, target_h: int, target_w: int):
    image = Image.fromarray(img, "RGB")
    orig_w, orig_h = image.size
    scale = min(target_w / orig_w, target_h / orig_h)
    new_w, new_h = int(orig_w * scale), int(orig_h * scale)
    image = image.resize((new_w, new_h), Image.Resampling.LANCZOS)
    canvas = Image.new("RGB", (target_w, target_h), (255, 255, 255))
    canvas.paste(image, ((target_w - new_w) // 2, (target_h - new_h) // 2))
    arr = np.asarray(canvas).copy()
    t = torch.from_numpy(arr).permute(2, 0, 1).to(torch.float32).div(255).sub(0.5).div(0.5)
    return arr, t.numpy()


def main():
    rng = np.random.default_rng(20240611)
    sizes = [(100, 800), (47, 1013), (64, 640), (300, 90), (9, 35), (63, 641), (31, 200), (70, 640)]
    out = {}
    for i, (h, w) in enumerate(sizes):
        if h * w > 12000:       # large crops: ink-like strokes on a smooth background (compresses well, still exercises
            yy, xx = np.mgrid[0:h, 0:w]                                  # every filter tap and the clipping overshoot at edges)
            img = np.stack([200 + (xx * 55 // w), 190 + (yy * 60 // h), 180 + ((xx + yy) * 70 // (w + h))], -1)
            for k in range(12):
                y0, x0 = int(rng.integers(0, h - 3)), int(rng.integers(0, w - 3))
                img[y0:y0 + int(rng.integers(2, max(3, h // 3))), x0:x0 + int(rng.integers(1, 9))] = int(rng.integers(0, 60))
            img = img.astype(np.uint8)
        else:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out[f"in{i}"] = img
        for th, tw in ((64, 640), (32, 64)):
            canvas, t = reference_transform(img, th, tw)
            out[f"canvas{i}_{th}x{tw}"] = canvas
            if th == 32:            # the fp32 tensor is lut[canvas] (256 possible values): stored for the small target only
                out[f"tensor{i}_{th}x{tw}"] = t
    u = torch.arange(256, dtype=torch.uint8)
    out["lut256"] = u.to(torch.float32).div(255).sub(0.5).div(0.5).numpy()
    out["n"] = np.array(len(sizes))
    import PIL
    out["pillow_version"] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resize_kat.npz"), **out)
    print("wrote", len(sizes), "crops; Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
