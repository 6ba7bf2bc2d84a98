"""dev: the fp32-operand GEMMs (gemm_f32.hip) at the ResNet34 shapes of tools/dev/ocr_bench.py (batch 16 of 64 x 512)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import numpy as np, torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
def bench(fn, it=30):
    for _ in range(5): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
# (name, M, Cout, Kp, count per step fwd)
shapes = [("stem 7x7", 131072, 64, 192, 1), ("layer1 3x3", 32768, 64, 576, 6), ("layer2 3x3", 8192, 128, 1152, 7), ("layer3 3x3", 2048, 256, 2304, 11), ("layer4 3x3", 512, 512, 4608, 5)]
tot = [0.0, 0.0, 0.0]
for name, M, N, K, cnt in shapes:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05; out = torch.empty(M, N, device=dev)
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=None, resid=None, ldr=N, aux=None, ldaux=0, M=M, N=N, K=K, n_valid=N, drop_p=0.0, drop_key=0)
    t_f = bench(lambda: lib.kzv_gemm_nt_f32(C.byref(a), 1, st))
    dY = torch.randn(M, N, device=dev); Bt = torch.randn(K, N, device=dev) * 0.05; dc = torch.empty(M, K, device=dev)
    b = L.kzv_gemm_nt_args(A=dY.data_ptr(), lda=N, B=Bt.data_ptr(), ldb=N, C=dc.data_ptr(), ldc=K, bias=None, resid=None, ldr=K, aux=None, ldaux=0, M=M, N=K, K=N, n_valid=K, drop_p=0.0, drop_key=0)
    t_d = bench(lambda: lib.kzv_gemm_nt_f32(C.byref(b), 1, st))
    gp = torch.zeros(N, K, device=dev)
    c = L.kzv_gemm_tn_args(P=dY.data_ptr(), ldp=N, Q=A.data_ptr(), ldq=K, OUT=gp.data_ptr(), ldo=K, Mtok=M, N=N, K=K, n_store=N, dbias=None)
    t_w = bench(lambda: lib.kzv_gemm_tn_f32(C.byref(c), st))
    fl = 2.0 * M * N * K
    print(f"{name:12s} M {M:6d} N {N:3d} K {K:4d}: fwd {t_f:6.1f} us ({fl/t_f/1e6:5.1f} TF)  dgrad {t_d:6.1f} us ({fl/t_d/1e6:5.1f} TF)  wgrad {t_w:6.1f} us ({fl/t_w/1e6:5.1f} TF)   x{cnt}")
    tot[0] += t_f * cnt; tot[1] += t_d * cnt; tot[2] += t_w * cnt
print(f"per step (3x3 / 7x7 convolutions only): fwd {tot[0]/1e3:.2f} ms, dgrad {tot[1]/1e3:.2f} ms, wgrad {tot[2]/1e3:.2f} ms")
