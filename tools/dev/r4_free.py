"""dev: the free-running K loop (gemm_nt256f.hip) against the eight-phase ping-pong (gemm_nt256p.hip): bit-equality, then
interleaved timing in ONE process (kzv_set_nt_schedule 0 / 1), per epilogue and shape; K-slope of the main loop at N = 768."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import numpy as np, torch
from kzv import _lib as L
lib = L.load()
dev = "cuda"
def st(): return torch.cuda.current_stream().cuda_stream
def mk(M, N, K, epi, nv=None):
    nv = nv or N
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(nv, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(nv, device=dev)
    out = torch.empty(M, N, dtype=torch.float32 if epi in (1, 3, 5) else torch.bfloat16, device=dev)
    res = torch.randn(M, N, device=dev) if epi == 3 else None
    aux = (torch.rand(M, N, device=dev) * 1.2).bfloat16() if epi in (2, 4, 5) else None
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr() if epi != 4 else None,
                           resid=L.ptr(res), ldr=N, aux=L.ptr(aux), ldaux=N, M=M, N=N, K=K, n_valid=nv, drop_p=0.1 if epi == 3 else 0.0, drop_key=5)
    keep = (A, B, bias, out, res, aux)
    return a, out, keep
def run(a, epi): L.check(lib.kzv_gemm_nt(C.byref(a), epi, st()))
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("check", "all"):
    for (M, N, K, epi, nv) in [(41216, 768, 768, 0, None), (41216, 2304, 768, 0, None), (41216, 768, 3072, 3, None), (41216, 3072, 768, 4, None),
                               (24600, 1024, 128, 1, None), (24600, 1024, 128, 0, 1000), (24600, 1024, 256, 3, 1000), (41216, 768, 768, 1, None)]:
        a, out, keep = mk(M, N, K, epi, nv)
        lib.kzv_set_nt_schedule(0); run(a, epi); torch.cuda.synchronize(); ref = out.clone()
        lib.kzv_set_nt_schedule(1)
        bad = 0
        for _ in range(6):
            out.zero_(); run(a, epi); torch.cuda.synchronize()
            bad += int((out != ref).sum().item()) if not torch.equal(out, ref) else 0
        A, B = keep[0], keep[1]
        print(f"check epi{epi} {M}x{N}x{K} nv={nv}: mismatching elements over 6 runs = {bad}", flush=True)
        assert bad == 0
if mode in ("time", "all", "time2"):
    def bench(a, epi, it=40):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): run(a, epi)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    shapes = [(41216, 2304, 768, 0), (41216, 768, 768, 3), (41216, 768, 3072, 3), (41216, 3072, 768, 4), (41216, 768, 3072, 0),
              (41216, 768, 768, 0), (41216, 768, 2304, 0), (41216, 768, 1536, 0)]
    if mode == "time2": shapes = [(41216, 2304, 768, 0), (41216, 768, 3072, 0), (41216, 768, 768, 0)]
    res = {}
    for (M, N, K, epi) in shapes:
        a, out, keep = mk(M, N, K, epi)
        for s in (0, 1): lib.kzv_set_nt_schedule(s); bench(a, epi, 10)
        t = {0: [], 1: []}
        for r in range(5):
            for s in (0, 1):
                lib.kzv_set_nt_schedule(s); t[s].append(bench(a, epi))
        m0, m1 = np.median(t[0]), np.median(t[1])
        res[(N, K, epi)] = (m0, m1)
        fl = 2.0 * M * N * K
        print(f"time epi{epi} {M}x{N}x{K}: pingpong {m0:7.1f} us ({fl/m0/1e6:6.0f} TF)  free {m1:7.1f} us ({fl/m1/1e6:6.0f} TF)  {100*(m0/m1-1):+.1f} %", flush=True)
    # K-slope at N = 768, BF16 epilogue: main-loop rate = 2*M*N*dK / dt
    for s, nm in ((0, "pingpong"), (1, "free")):
        t768, t3072 = res[(768, 768, 0)][s], res[(768, 3072, 0)][s]
        print(f"K-slope {nm}: {2.0*41216*768*(3072-768)/(t3072-t768)/1e6:.0f} TF/s main loop")
lib.kzv_set_nt_schedule(-1)

if mode in ("tn", "all"):
    def mk_tn(Mt, N, K, bias=True):
        P = torch.randn(Mt, N, device=dev).bfloat16(); Q = torch.randn(Mt, K, device=dev).bfloat16()
        O = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        a = L.kzv_gemm_tn_args(P=P.data_ptr(), ldp=N, Q=Q.data_ptr(), ldq=K, OUT=O.data_ptr(), ldo=K, Mtok=Mt, N=N, K=K, n_store=N,
                               dbias=db.data_ptr() if bias else None)
        return a, O, db, (P, Q)
    def run_tn(a): L.check(lib.kzv_gemm_tn(C.byref(a), st()))
    for (Mt, N, K) in [(41216, 2304, 768), (41216, 768, 3072), (41216, 3072, 768), (41216, 768, 768), (24640, 1000, 776), (8192, 2304, 768)]:
        a, O, db, keep = mk_tn(Mt, N, K)
        lib.kzv_set_tn_schedule(0); O.zero_(); db.zero_(); run_tn(a); torch.cuda.synchronize(); ref = O.clone(); rb = db.clone()
        lib.kzv_set_tn_schedule(1)
        bad = 0
        for _ in range(4):
            O.zero_(); db.zero_(); run_tn(a); torch.cuda.synchronize()
            bad += int((O != ref).sum().item())
        berr = float((db - rb).abs().max() / (rb.abs().max() + 1e-9))
        want = keep[0].float().t() @ keep[1].float()
        rel = float((O - want).abs().max() / want.abs().max())
        print(f"tn check {Mt}: {N}x{K}: mismatching elements over 4 runs = {bad}; bias rel diff {berr:.2e}; vs fp32 torch rel {rel:.2e}", flush=True)
        assert bad == 0 and berr < 1e-5
    def bench_tn(a, it=30):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): run_tn(a)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    for (Mt, N, K) in [(41216, 2304, 768), (41216, 768, 768), (41216, 3072, 768), (41216, 768, 3072)]:
        a, O, db, keep = mk_tn(Mt, N, K)
        for s in (0, 1): lib.kzv_set_tn_schedule(s); bench_tn(a, 10)
        t = {0: [], 1: []}
        for r in range(5):
            for s in (0, 1):
                lib.kzv_set_tn_schedule(s); t[s].append(bench_tn(a))
        m0, m1 = np.median(t[0]), np.median(t[1])
        fl = 2.0 * Mt * N * K
        print(f"tn time {Mt}: {N}x{K}: pingpong {m0:7.1f} us ({fl/m0/1e6:6.0f} TF)  free {m1:7.1f} us ({fl/m1/1e6:6.0f} TF)  {100*(m0/m1-1):+.1f} %  (launch + fold)", flush=True)
    lib.kzv_set_tn_schedule(-1)
