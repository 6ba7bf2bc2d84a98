import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
M, H = 20608, 768
g = lambda *s: torch.randn(*s, device=dev)
for wscale in (0.02, 0.05):
    A = g(M, H).bfloat16(); W = (g(3 * H, H) * wscale).bfloat16(); bias = g(3 * H); out = torch.empty(M, 3 * H, dtype=torch.bfloat16, device=dev)
    ga = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=H, B=W.data_ptr(), ldb=H, C=out.data_ptr(), ldc=3 * H, bias=bias.data_ptr(), M=M, N=3 * H, K=H, n_valid=3 * H)
    s1 = torch.cuda.Stream()
    for name, s in (("default", torch.cuda.current_stream()), ("side", s1)):
        for n in (5, 40, 200):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s):
                for _ in range(5): L.check(lib.kzv_gemm_nt(C.byref(ga), 0, s.cuda_stream))
                e0.record()
                for _ in range(n): L.check(lib.kzv_gemm_nt(C.byref(ga), 0, s.cuda_stream))
                e1.record()
            torch.cuda.synchronize()
            print(f"w {wscale} stream {name} x{n}: {e0.elapsed_time(e1) / n * 1e3:.1f} us each", flush=True)
