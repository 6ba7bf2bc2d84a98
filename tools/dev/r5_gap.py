"""dev: is there a ~4.4 us boundary between certain kernel pairs (seen in the rocprofv3 trace) without the profiler?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
def st(): return torch.cuda.current_stream().cuda_stream
B, S, H, heads = 256, 161, 768, 12; M = B * S
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
A = bf(M, H); W = (torch.randn(H, H, device=dev) * 0.02).bfloat16(); out = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
ga = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=H, B=W.data_ptr(), ldb=H, C=out.data_ptr(), ldc=H, M=M, N=H, K=H, n_valid=H)
P = bf(M, H); Q = bf(M, H); O = torch.zeros(H, H, device=dev)
ta = L.kzv_gemm_tn_args(P=P.data_ptr(), ldp=H, Q=Q.data_ptr(), ldq=H, OUT=O.data_ptr(), ldo=H, Mtok=M, N=H, K=H, n_store=H)
qkv = bf(M, 3 * H); o = torch.empty(M, H, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, heads, S, device=dev)
at = L.kzv_attn_args(Q=qkv.data_ptr(), K=qkv[:, H:].data_ptr(), V=qkv[:, 2 * H:].data_ptr(), O=o.data_ptr(), LSE=lse.data_ptr(), ldq=3 * H, ldk=3 * H, ldv=3 * H, ldo=H,
                     B=B, heads=heads, Sq=S, Sk=S, mode=0, drop_p=0.0, drop_key=7)
x = torch.randn(M, H, device=dev); y = torch.empty(M, H, dtype=torch.bfloat16, device=dev); stt = torch.empty(M, 2, device=dev); gam = torch.ones(H, device=dev); bet = torch.zeros(H, device=dev)
ops = {"nt": lambda: L.check(lib.kzv_gemm_nt(C.byref(ga), 0, st())), "tn": lambda: L.check(lib.kzv_gemm_tn(C.byref(ta), st())),
       "attn": lambda: L.check(lib.kzv_attn_fwd(C.byref(at), st())),
       "ln": lambda: L.check(lib.kzv_layernorm_fwd(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), 0, stt.data_ptr(), M, H, 1e-12, st()))}
def bench(seq, it=100):
    def run():
        for k in seq: ops[k]()
    for _ in range(5): run()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
single = {k: bench([k]) for k in ops}
print("alone:", {k: round(v, 1) for k, v in single.items()})
for pair in (["nt", "tn"], ["nt", "attn"], ["nt", "ln"], ["ln", "tn"], ["ln", "attn"], ["attn", "tn"], ["nt", "nt"], ["tn", "tn"]):
    t = bench(pair); print(pair, f"{t:.1f} us vs sum {sum(single[k] for k in pair):.1f}  (+{t - sum(single[k] for k in pair):.1f})")
