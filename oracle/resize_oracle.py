"""CPU restatement of the reference's image transform (test infrastructure only: imported by tests/, smoke and tools).

Path restated: `ResizeWithPadding.__call__` (src/data/trocr_dataset.py:24-53) -> `transforms.ToTensor()` ->
`transforms.Normalize(0.5, 0.5)` (src/data/trocr_dataset.py:97-104).  The resize itself is Pillow's
`Image.resize(..., Image.Resampling.LANCZOS)` -- a third-party dependency that is not part of /root/reference (Pillow
12.2.0 in this image; the algorithm below is `ImagingResample` of Pillow's src/libImaging/Resample.c as published:
separable two-pass resampling, horizontal pass first, per-output-pixel windows [xmin, xmin+xmax) with Lanczos-3 weights
evaluated in double precision at (x + xmin - center + 0.5) / filterscale, normalised, rounded to 22-bit fixed point;
every pass accumulates in int32 from 1 << 21 and clips (ss >> 22) to uint8).  Pinned by tests/golden/resize_kat.npz,
produced by running Pillow itself (tools/gen_golden_resize.py), and by direct comparison with Pillow wherever it is
importable.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
LANCZOS_SUPPORT = 3.0


def target_geometry(orig_w: int, orig_h: int, target_h: int, target_w: int):
    """ResizeWithPadding: aspect-preserving size and centred paste position (trocr_dataset.py:30-47)."""
    scale = min(target_w / orig_w, target_h / orig_h)
    new_w, new_h = int(orig_w * scale), int(orig_h * scale)
    return new_w, new_h, (target_w - new_w) // 2, (target_h - new_h) // 2


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def coefficients(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the whole-image box: (bounds [out,2] int32 = (xmin, count),
    kk [out, ksize] int32 fixed-point weights)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w) if False else 0.0
        for v in w:                                  # the C loop accumulates in this order
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(v * (1 << PRECISION_BITS) + (-0.5 if v < 0 else 0.5))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """one 8-bit resampling pass along `axis` of an [H, W, C] uint8 image"""
    src = np.moveaxis(img, axis, 0).astype(np.int64)                     # [n_in, other, C]
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for o in range(bounds.shape[0]):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[o, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_lanczos(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """Image.resize((new_w, new_h), LANCZOS) on an [H, W, 3] uint8 array (Pillow skips a pass whose size is unchanged)."""
    h, w = img.shape[:2]
    out = img
    if new_w != w:
        out = _pass(out, *coefficients(w, new_w), axis=1)
    if new_h != h:
        out = _pass(out, *coefficients(h, new_h), axis=0)
    return out


def normalise_lut() -> np.ndarray:
    """ToTensor (uint8 / 255 in fp32) then Normalize(0.5, 0.5) ((x - 0.5) / 0.5 in fp32): 256 possible values."""
    x = np.arange(256, dtype=np.float32) / np.float32(255)
    return ((x - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


def transform(img: np.ndarray, target_h: int, target_w: int) -> np.ndarray:
    """[H, W, 3] uint8 -> [3, target_h, target_w] fp32 in [-1, 1]; white (= +1) padding."""
    h, w = img.shape[:2]
    new_w, new_h, px, py = target_geometry(w, h, target_h, target_w)
    canvas = np.full((target_h, target_w, 3), 255, np.uint8)
    canvas[py:py + new_h, px:px + new_w] = resize_lanczos(img, new_w, new_h)
    return normalise_lut()[canvas].transpose(2, 0, 1).copy()
