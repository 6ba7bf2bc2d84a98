"""TEST INFRASTRUCTURE (oracle side; never imported by the product path).

numpy statement of the attention-probability dropout generator of the HIP kernels (kuzushiji-vision_amd/csrc/kzv_common.h,
"attention-probability dropout"; drawn where the reference calls nn.functional.dropout on the softmax output: HF
modeling_vit.py:184, modeling_roberta.py:178).  The reference draws its masks from torch's Philox stream, which no other
implementation can reproduce bit for bit; what parity needs is (i) that the masks the kernels draw are Bernoulli(1 - p) with no
structure a model could see -- checked statistically in tests/test_oracle_golden.py -- and (ii) that a test can replay the
kernels' exact masks through the oracle's explicit-mask forward -- kzv_debug_attn_dropout_mask, checked bit for bit against
this file in tests/test_ops_gpu.py.

One 32-bit hash per 4 x 4 block of a (batch, head)'s [Sq, Sk] matrix, one 16-bit multiply per element:
    x = mix(block * 0x9E3779B9 + key),  block = ((b * heads + h) * ceil(Sq / 4) + (q >> 2)) * ceil(Sk / 4) + (k >> 2)
    u = ((half ^ C[q & 3][k & 3]) * M[q & 3][k & 3]) mod 2^16,  half = low 16 bits of x if (q + k) is even, else the high ones
    kept  iff  int16(u) >= thr16 - 32768,   thr16 = round(p * 65536)            (so P(drop) = thr16 / 65536)
"""
import numpy as np

C = np.array([[0xba79, 0x0e76, 0x9b89, 0x53b0], [0x431d, 0x0cc3, 0xa452, 0x4805],
              [0xd3bc, 0xd36a, 0x9c49, 0x9be5], [0xd12f, 0x8ff4, 0x38f5, 0x7f7a]], dtype=np.uint32)
M = np.array([[0x5195, 0x0735, 0xf067, 0x26fb], [0xbafb, 0xee95, 0xe455, 0x0e9d],
              [0x1f77, 0xc189, 0x6fa9, 0x4599], [0x31b9, 0x473d, 0xf055, 0xa1a9]], dtype=np.uint32)


def thr16_of(p: float) -> int:
    """kzv_drop_params (csrc/host.cpp): the 16-bit threshold actually used for drop probability p."""
    if not p > 0:
        return 0
    return int(min(65535, max(1, np.rint(np.float32(p) * np.float32(65536.0)))))


def u16(key: int, pairs: int, Sq: int, Sk: int) -> np.ndarray:
    """The 16-bit value of every element: uint32 array [pairs, Sq, Sk] holding values < 65536."""
    u32 = np.uint32
    nQ4, nK4 = (Sq + 3) >> 2, (Sk + 3) >> 2
    pr = np.arange(pairs, dtype=np.uint32)[:, None, None]
    q = np.arange(Sq, dtype=np.uint32)[None, :, None]
    k = np.arange(Sk, dtype=np.uint32)[None, None, :]
    with np.errstate(over="ignore"):
        x = ((pr * u32(nQ4) + (q >> 2)) * u32(nK4) + (k >> 2)) * u32(0x9E3779B9) + u32(key & 0xFFFFFFFF)
        x ^= x >> 16
        x = x * u32(0x7feb352d)
        x ^= x >> 15
        r, c = q & 3, k & 3
        half = np.where(((r + c) & 1) == 1, x >> 16, x & u32(0xffff))
        return ((half ^ C[r, c]) * M[r, c]) & u32(0xffff)


def keep_mask(key: int, p: float, pairs: int, Sq: int, Sk: int) -> np.ndarray:
    """bool [pairs, Sq, Sk]: True where the probability is kept."""
    t = thr16_of(p)
    if t == 0:
        return np.ones((pairs, Sq, Sk), dtype=bool)
    u = u16(key, pairs, Sq, Sk)
    return (u ^ 0x8000) >= t           # int16(u) >= t - 32768


def multiplier(key: int, p: float, pairs: int, Sq: int, Sk: int) -> np.ndarray:
    """What kzv_debug_attn_dropout_mask writes: 0 or 1 / P(keep) of the threshold actually used, fp32 [pairs * Sq, Sk]."""
    t = thr16_of(p)
    inv = np.float32(65536.0) / np.float32(65536 - t) if t else np.float32(1.0)
    return (keep_mask(key, p, pairs, Sq, Sk).astype(np.float32) * inv).reshape(pairs * Sq, Sk)
