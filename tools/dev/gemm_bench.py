"""dev: time the GEMM shapes of one training step (B=256).  KZV_NT_VARIANT selects the tile variant."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load()
dev = "cuda"
def st(): return torch.cuda.current_stream().cuda_stream
def bench(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
which = sys.argv[1] if len(sys.argv) > 1 else "nt"
tot = 0.0
if which in ("nt", "all"):
    shapes = [(41216, 2304, 768, 0, 1), (41216, 768, 768, 3, 1), (41216, 3072, 768, 2, 1), (41216, 768, 3072, 3, 1),
              (41216, 3072, 768, 4, 1), (41216, 768, 3072, 0, 1), (41216, 768, 768, 0, 1), (41216, 768, 2304, 0, 1),
              (32512, 768, 256, 0, 0.5), (32512, 256, 256, 3, 1.5), (32512, 256, 768, 3, 0.5), (32512, 4352, 256, 1, 1 / 12)]
    for (M, N, K, epi, weight) in shapes:
        A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, dtype=torch.float32 if epi in (1, 3) else torch.bfloat16, device=dev)
        res = torch.randn(M, N, device=dev) if epi == 3 else None
        aux = torch.randn(M, N, device=dev).bfloat16() if epi in (2, 4) else None
        a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr(),
                               resid=L.ptr(res), ldr=N, aux=L.ptr(aux), ldaux=N, M=M, N=N, K=K, n_valid=N, drop_p=0.1 if epi == 3 else 0.0, drop_key=5)
        us = bench(lambda: L.check(lib.kzv_gemm_nt(C.byref(a), epi, st())))
        tot += us * weight
        print(f"nt epi{epi} {M}x{N}x{K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s")
    print(f"weighted NT time per encoder layer (+dec share): {tot:.0f} us")
if which in ("tn", "all"):
    tot = 0.0
    shapes = [(41216, 2304, 768, 1), (41216, 768, 768, 1), (41216, 3072, 768, 1), (41216, 768, 3072, 1),
              (32512, 256, 256, 1.5), (32512, 768, 256, 1), (32512, 256, 768, 0.5), (32512, 4352, 256, 1 / 12), (40960, 3072, 256, 1 / 12)]
    for (Mt, N, K, weight) in shapes:
        P = torch.randn(Mt, N, device=dev).bfloat16(); Q = torch.randn(Mt, K, device=dev).bfloat16()
        O = torch.zeros(N, K, device=dev)
        a = L.kzv_gemm_tn_args(P=P.data_ptr(), ldp=N, Q=Q.data_ptr(), ldq=K, OUT=O.data_ptr(), ldo=K, Mtok=Mt, N=N, K=K, n_store=N)
        us = bench(lambda: L.check(lib.kzv_gemm_tn(C.byref(a), st())))
        tot += us * weight
        print(f"tn {Mt}: {N}x{K}: {us:8.1f} us  {2.0*Mt*N*K/us/1e6:7.1f} TF/s")
    print(f"weighted TN time per layer: {tot:.0f} us")
