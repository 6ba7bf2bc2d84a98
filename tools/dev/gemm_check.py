"""dev: GEMM correctness + timing on the GPU (python tools/dev/gemm_check.py)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = C.CDLL(L.LIB_PATH)
lib.kzv_last_error.restype = C.c_char_p
lib.kzv_gemm_nt.argtypes = [C.POINTER(L.kzv_gemm_nt_args), C.c_int, C.c_void_p]
lib.kzv_gemm_tn.argtypes = [C.POINTER(L.kzv_gemm_tn_args), C.c_void_p]
dev = "cuda"
torch.manual_seed(0)
def st(): return torch.cuda.current_stream().cuda_stream
def nt(A, B, bias=None, epi=0, n_store=None, resid=None, aux=None):
    M, K = A.shape; nv = B.shape[0]; N = n_store or nv
    out_dt = torch.float32 if epi in (1, 3) else torch.bfloat16
    Cc = torch.empty(M, N, dtype=out_dt, device=dev)
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=A.stride(0), B=B.data_ptr(), ldb=B.stride(0), C=Cc.data_ptr(), ldc=N,
        bias=0 if bias is None else bias.data_ptr(), resid=0 if resid is None else resid.data_ptr(), ldr=N,
        aux=0 if aux is None else aux.data_ptr(), ldaux=N, M=M, N=N, K=K, n_valid=nv, drop_p=0.0, drop_key=0)
    rc = lib.kzv_gemm_nt(C.byref(a), epi, st())
    assert rc == 0, lib.kzv_last_error()
    return Cc
def tn(P, Q):
    Mt, N = P.shape; K = Q.shape[1]
    O = torch.zeros(N, K, dtype=torch.float32, device=dev)
    a = L.kzv_gemm_tn_args(P=P.data_ptr(), ldp=P.stride(0), Q=Q.data_ptr(), ldq=Q.stride(0), OUT=O.data_ptr(), ldo=K, Mtok=Mt, N=N, K=K, n_store=N)
    rc = lib.kzv_gemm_tn(C.byref(a), st())
    assert rc == 0, lib.kzv_last_error()
    return O
ok = True
for (M, N, K) in [(128, 128, 64), (200, 132, 128), (483, 384, 192), (1000, 4300, 256)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); bias = torch.randn(N, device=dev)
    ref = A.float() @ B.float().t() + bias
    got = nt(A, B, bias, epi=1)
    err = (got - ref).abs().max().item()
    print("nt f32", M, N, K, "maxerr", err); ok &= err < 1e-2
    got = nt(A, B, bias, epi=0).float()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print("nt bf16 rel", err); ok &= err < 1e-2
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    got = nt(A, B, bias, epi=2, aux=aux).float()
    err = (got - torch.nn.functional.gelu(ref)).abs().max().item(); print("nt gelu", err); ok &= err < 0.1
    err = (aux.float() - ref).abs().max().item() / ref.abs().max().item(); print("   pre", err); ok &= err < 1e-2
    res = torch.randn(M, N, device=dev)
    got = nt(A, B, bias, epi=3, resid=res); err = (got - ref - res).abs().max().item(); print("nt resid", err); ok &= err < 1e-2
for (Mt, N, K) in [(64, 128, 128), (200, 136, 64), (1000, 256, 384), (3000, 4352, 256)]:
    P = torch.randn(Mt, N, device=dev).bfloat16(); Q = torch.randn(Mt, K, device=dev).bfloat16()
    ref = P.float().t() @ Q.float()
    got = tn(P, Q)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print("tn", Mt, N, K, "relerr", err); ok &= err < 1e-3
print("ALL OK" if ok else "FAILED")
# timing
def bench(fn, flops, name, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s")
M = 41216
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16(); bias = torch.randn(N, device=dev)
    bench(lambda: nt(A, B, bias, epi=0), 2.0*M*N*K, f"nt {M}x{N}x{K}")
    ref = lambda: torch.nn.functional.linear(A, B)
    bench(ref, 2.0*M*N*K, f"   torch(hipblaslt) same")
for (N, K) in [(2304, 768), (768, 768), (3072, 768)]:
    P = torch.randn(M, N, device=dev).bfloat16(); Q = torch.randn(M, K, device=dev).bfloat16()
    bench(lambda: tn(P, Q), 2.0*M*N*K, f"tn {M}: {N}x{K}")
