"""dev: do an MFMA-bound persistent GEMM on half the CUs and an HBM-bound kernel on the other half overlap (max, not sum)?
usage: KZV_NT_GRID=128 r5_overlap.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
M, H = 20608, 768
g = lambda *s: torch.randn(*s, device=dev)
A = g(M, H).bfloat16(); W = (g(3 * H, H) * 0.02).bfloat16(); bias = g(3 * H); out = torch.empty(M, 3 * H, dtype=torch.bfloat16, device=dev)
ga = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=H, B=W.data_ptr(), ldb=H, C=out.data_ptr(), ldc=3 * H, bias=bias.data_ptr(), M=M, N=3 * H, K=H, n_valid=3 * H)
x = g(2 * M, H); y = torch.empty(2 * M, H, dtype=torch.bfloat16, device=dev); st = torch.empty(2 * M, 2, device=dev); gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(n):
    for _ in range(n): L.check(lib.kzv_gemm_nt(C.byref(ga), 0, s1.cuda_stream))
def ln(n, rows):
    for _ in range(n): L.check(lib.kzv_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), 0, st.data_ptr(), rows, H, 1e-12, s2.cuda_stream))
def timed(f):
    for _ in range(2):
        f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream(); e0.record()
    s1.wait_stream(cur); s2.wait_stream(cur); f(); cur.wait_stream(s1); cur.wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
NG = 40
tg = timed(lambda: gemm(NG)); print(f"gemm qkv M={M} x{NG} alone (KZV_NT_GRID={os.environ.get('KZV_NT_GRID')}): {tg:.0f} us = {tg / NG:.1f} each")
for rows in (M, 2 * M):
    t1 = timed(lambda: ln(1, rows)) ; n = 100; tl = timed(lambda: ln(n, rows)); print(f"ln rows={rows} x{n} alone: {tl:.0f} us = {tl / n:.1f} each")
    nl = max(1, int(tg / (tl / n)))
    tb = timed(lambda: (gemm(NG), ln(nl, rows)))
    print(f"  gemm x{NG} || ln x{nl}: {tb:.0f} us  (sum of the two alone: {tg + tl / n * nl:.0f}, max: {max(tg, tl / n * nl):.0f})")
