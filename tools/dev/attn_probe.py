"""dev: is attention backward bound by memory?  Same launch with (a) real strides, (b) Q/K/V rows aliased (row stride 0: every
read hits cache), (c) a quarter-filled chip."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
def st(): return torch.cuda.current_stream().cuda_stream
def bench(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
heads, Sq, Sk = 12, 161, 161
H = heads * 64
for B, alias in [(256, False), (256, True), (64, False), (16, False)]:
    qkv = torch.randn(B * Sq, 3 * H, device=dev).bfloat16(); q, k, v = qkv, qkv[:, H:], qkv[:, 2 * H:]
    dqkv = torch.zeros_like(qkv); dq, dk, dv = dqkv, dqkv[:, H:], dqkv[:, 2 * H:]
    o = torch.empty(B * Sq, H, dtype=torch.bfloat16, device=dev); do = torch.randn(B * Sq, H, device=dev).bfloat16()
    lse = torch.empty(B, heads, Sq, device=dev)
    for p in (0.0, 0.1):
        a = L.kzv_attn_args(Q=q.data_ptr(), K=k.data_ptr(), V=v.data_ptr(), O=o.data_ptr(), LSE=lse.data_ptr(), dO=do.data_ptr(),
                            dQ=dq.data_ptr(), dK=dk.data_ptr(), dV=dv.data_ptr(), ldq=3 * H, ldk=3 * H, ldv=3 * H, ldo=H, ids=None,
                            ld_ids=0, pad_id=1, B=B, heads=heads, Sq=Sq, Sk=Sk, mode=0, drop_p=p, drop_key=7)
        L.check(lib.kzv_attn_fwd(C.byref(a), st()))
        if alias:
            a.ldq = 0; a.ldk = 0; a.ldv = 0
        f = bench(lambda: L.check(lib.kzv_attn_fwd(C.byref(a), st())))
        b = bench(lambda: L.check(lib.kzv_attn_bwd(C.byref(a), st())))
        print(f"B{B} alias={alias} p={p}: fwd {f:7.1f} us  bwd {b:7.1f} us   (per 256-batch-equivalent: fwd {f * 256 / B:6.1f} bwd {b * 256 / B:6.1f})")
