"""dev: time attention fwd/bwd at the bench shapes."""
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
for (B, heads, Sq, Sk, mode, packed) in [(256, 12, 161, 161, 0, True), (256, 4, 127, 127, 1, True), (256, 4, 127, 160, 0, False)]:
    H = heads * 64
    if packed:
        qkv = torch.randn(B * Sq, 3 * H, device=dev).bfloat16(); q, k, v = qkv, qkv[:, H:], qkv[:, 2 * H:]; ldq = ldk = 3 * H
        dqkv = torch.zeros_like(qkv); dq, dk, dv = dqkv, dqkv[:, H:], dqkv[:, 2 * H:]
    else:
        q = torch.randn(B * Sq, H, device=dev).bfloat16(); kv = torch.randn(B * Sk, 2 * H, device=dev).bfloat16(); k, v = kv, kv[:, H:]; ldq = H; ldk = 2 * H
        dq = torch.zeros_like(q); dkv = torch.zeros_like(kv); dk, dv = dkv, dkv[:, H:]
    o = torch.empty(B * Sq, H, dtype=torch.bfloat16, device=dev); do = torch.randn(B * Sq, H, device=dev).bfloat16()
    lse = torch.empty(B, heads, Sq, device=dev)
    ids = torch.randint(5, 100, (B, Sk + 1), device=dev, dtype=torch.int64); ids[:, 40:] = 1
    for p in (0.0, 0.1):
        a = L.kzv_attn_args(Q=q.data_ptr(), K=k.data_ptr(), V=v.data_ptr(), O=o.data_ptr(), LSE=lse.data_ptr(), dO=do.data_ptr(),
                            dQ=dq.data_ptr(), dK=dk.data_ptr(), dV=dv.data_ptr(), ldq=ldq, ldk=ldk, ldv=ldk, ldo=H, ids=ids.data_ptr(),
                            ld_ids=Sk + 1, pad_id=1, B=B, heads=heads, Sq=Sq, Sk=Sk, mode=mode, drop_p=p, drop_key=7)
        f = bench(lambda: L.check(lib.kzv_attn_fwd(C.byref(a), st())))
        b = bench(lambda: L.check(lib.kzv_attn_bwd(C.byref(a), st())))
        fl = 4.0 * B * heads * Sq * Sk * 64
        print(f"B{B} h{heads} {Sq}x{Sk} mode{mode} p={p}: fwd {f:7.1f} us ({fl/f/1e6:6.1f} TF/s)  bwd {b:7.1f} us ({2.5*fl/b/1e6:6.1f} TF/s)")
