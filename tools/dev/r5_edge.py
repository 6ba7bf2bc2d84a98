"""dev: gemm_nt time against M (edge row tile or not), qkv shape."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
def st(): return torch.cuda.current_stream().cuda_stream
def bench(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for (N, K, epi) in [(2304, 768, 0), (768, 768, 3), (768, 3072, 3), (3072, 768, 2), (3072, 768, 4), (768, 3072, 0)]:
    for M in (41216, 20608, 20480, 20736, 10304):
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=torch.float32 if epi in (1, 3) else torch.bfloat16, device=dev)
        res = torch.randn(M, N, device=dev) if epi == 3 else None
        aux = torch.randn(M, N, device=dev).bfloat16() if epi in (2, 4) else None
        a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr(),
                               resid=L.ptr(res), ldr=N, aux=L.ptr(aux), ldaux=N, M=M, N=N, K=K, n_valid=N, drop_p=0.1 if epi == 3 else 0.0, drop_key=5)
        us = bench(lambda: L.check(lib.kzv_gemm_nt(C.byref(a), epi, st())))
        print(f"nt epi{epi} {M}x{N}x{K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
