"""dev: does the NT epilogue time track output bytes?  Same GEMM (41216 x 3072 x 768), epilogues writing 253 MB (bf16),
506 MB (fp32 / GELU's two bf16 tensors), and DGELU (253 MB read + 253 MB write)."""
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
M, N, K = 41216, 3072, 768
A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
for epi, name in [(0, "bf16 out"), (1, "fp32 out"), (2, "gelu (2 x bf16)"), (4, "dgelu (bf16 in + out)"), (3, "resid (fp32 in + out)")]:
    out = torch.empty(M, N, dtype=torch.float32 if epi in (1, 3) else torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev) if epi == 3 else None
    aux = torch.randn(M, N, device=dev).bfloat16() if epi in (2, 4) else None
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr(),
                           resid=L.ptr(res), ldr=N, aux=L.ptr(aux), ldaux=N, M=M, N=N, K=K, n_valid=N, drop_p=0.0, drop_key=5)
    us = bench(lambda: L.check(lib.kzv_gemm_nt(C.byref(a), epi, st())))
    print(f"epi{epi} {name:24s}: {us:7.1f} us")
