"""N2: the reference's image transform on the device.

`ResizeWithPadding` + `ToTensor` + `Normalize(0.5, 0.5)` (src/data/trocr_dataset.py:12-53, 97-104) for a batch of decoded
uint8 RGB crops of different sizes.  The host side computes what is per-crop and cheap (geometry with the reference's own
Python float arithmetic, Pillow's fixed-point Lanczos weights through `kzv_lanczos_coeffs`); the two resampling passes,
the paste and the normalisation run in HIP (`csrc/preprocess.hip`).  Byte-exact with Pillow.

`plan_line` needs no GPU (DataLoader workers can call it); `DevicePreprocessor` owns the device buffers.
"""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np

from . import _lib as L


def target_geometry(orig_w: int, orig_h: int, target_h: int, target_w: int):
    """ResizeWithPadding.__call__ (trocr_dataset.py:30-47): (new_w, new_h, paste_x, paste_y)."""
    scale = min(target_w / orig_w, target_h / orig_h)
    new_w, new_h = int(orig_w * scale), int(orig_h * scale)
    if new_w < 1 or new_h < 1:
        raise ValueError(f"crop {orig_w}x{orig_h} collapses to {new_w}x{new_h} at target {target_w}x{target_h}")   # Pillow raises too
    return new_w, new_h, (target_w - new_w) // 2, (target_h - new_h) // 2


@functools.lru_cache(maxsize=4096)
def lanczos_coeffs(in_size: int, out_size: int):
    """(bounds [out, 2] int32, kk [out, ksize] int32): Pillow's tables for one axis (cached: sizes repeat in a dataset)."""
    lib = L.load()
    ks = C.c_int(0)
    L.check(lib.kzv_lanczos_coeffs(in_size, out_size, None, None, C.byref(ks)), "lanczos_coeffs")
    bounds = np.empty((out_size, 2), np.int32)
    kk = np.empty((out_size, ks.value), np.int32)
    L.check(lib.kzv_lanczos_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data, C.byref(ks)), "lanczos_coeffs")
    bounds.setflags(write=False); kk.setflags(write=False)
    return bounds, kk


@functools.lru_cache(maxsize=4096)
def _lanczos_coeffs_t(in_size: int, out_size: int):
    """horizontal-pass form: weights transposed to [ksize, out] (adjacent output pixels read adjacent words on the GPU)"""
    b, k = lanczos_coeffs(in_size, out_size)
    kt = np.ascontiguousarray(k.T)
    kt.setflags(write=False)
    return b, kt


def plan_line(h: int, w: int, target_h: int, target_w: int) -> dict:
    """Everything `DevicePreprocessor` needs for one crop besides its pixels."""
    new_w, new_h, px, py = target_geometry(w, h, target_h, target_w)
    plan = {"in_h": h, "in_w": w, "new_h": new_h, "new_w": new_w, "paste_x": px, "paste_y": py, "h": None, "v": None}
    if new_w != w:
        plan["h"] = _lanczos_coeffs_t(w, new_w)
    if new_h != h:
        plan["v"] = lanczos_coeffs(h, new_h)
    return plan


def normalise_lut() -> np.ndarray:
    """ToTensor (uint8 -> fp32 / 255) then Normalize ((x - 0.5) / 0.5), for the 256 possible byte values."""
    x = np.arange(256, dtype=np.float32) / np.float32(255)
    return ((x - np.float32(0.5)) / np.float32(0.5)).astype(np.float32)


class DevicePreprocessor:
    """images: list of [H, W, 3] uint8 arrays (decoded RGB crops) -> torch.float32 [n, 3, target_h, target_w] on the GPU."""

    def __init__(self, image_size, device="cuda"):
        import torch
        self.target_h, self.target_w = int(image_size[0]), int(image_size[1])
        self.device = torch.device(device)
        self.lut = torch.from_numpy(normalise_lut()).to(self.device)

    def __call__(self, images, plans=None, out=None, target_w=None):
        """`target_w` overrides the canvas width for this batch (width buckets: every crop of a batch shares one bucket)."""
        import torch
        n = len(images)
        if n == 0:
            raise ValueError("empty batch")
        tw = int(target_w) if target_w else self.target_w
        plans = plans or [plan_line(im.shape[0], im.shape[1], self.target_h, tw) for im in images]
        desc = (L.kzv_line_desc * n)()
        coef_parts, coef_len, src_off, tmp_off, max_tmp = [], 0, 0, 0, 1
        for i, (im, p) in enumerate(zip(images, plans)):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3 or im.shape[:2] != (p["in_h"], p["in_w"]):
                raise ValueError("images must be [H, W, 3] uint8 arrays matching their plans")
            d = desc[i]
            d.src_off, d.tmp_off = src_off, tmp_off
            d.in_h, d.in_w, d.new_h, d.new_w, d.paste_x, d.paste_y = p["in_h"], p["in_w"], p["new_h"], p["new_w"], p["paste_x"], p["paste_y"]
            for key, boff, koff, ksz in (("h", "hb_off", "hk_off", "hk_size"), ("v", "vb_off", "vk_off", "vk_size")):
                if p[key] is None:
                    setattr(d, boff, 0); setattr(d, koff, 0); setattr(d, ksz, 0)
                    continue
                b, k = p[key]
                setattr(d, boff, coef_len); coef_parts.append(b.ravel()); coef_len += b.size
                setattr(d, koff, coef_len); coef_parts.append(k.ravel()); coef_len += k.size
                setattr(d, ksz, k.shape[0] if key == "h" else k.shape[1])
            src_off += im.shape[0] * im.shape[1] * 3
            tmp_off += p["in_h"] * p["new_w"] * 3
            max_tmp = max(max_tmp, p["in_h"] * p["new_w"])
        rgb = np.zeros(src_off + 4, np.uint8)             # + slack: the kernels fetch a pixel as one unaligned 32-bit word
        o = 0
        for im in images:
            sz = im.size
            rgb[o:o + sz] = np.ascontiguousarray(im).ravel(); o += sz
        coef = np.concatenate(coef_parts) if coef_parts else np.zeros(1, np.int32)
        d_rgb = torch.from_numpy(rgb).to(self.device, non_blocking=True)
        d_coef = torch.from_numpy(coef).to(self.device, non_blocking=True)
        d_desc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(self.device, non_blocking=True)
        d_tmp = torch.empty(tmp_off + 4, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty(n, 3, self.target_h, tw, dtype=torch.float32, device=self.device)
        L.check(L.load().kzv_preprocess_lines(d_rgb.data_ptr(), d_desc.data_ptr(), d_coef.data_ptr(), n, self.target_h, tw,
                                               max_tmp, self.lut.data_ptr(), d_tmp.data_ptr(), out.data_ptr(), L.stream_handle()),
                "preprocess_lines")
        # the staging tensors must outlive the asynchronous kernels: keep them until the next call
        self._keep = (d_rgb, d_coef, d_desc, d_tmp)
        return out
