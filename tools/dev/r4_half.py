"""dev: the four-wave 256x128 kernel, two workgroups per CU (gemm_nt256h.hip, kzv_set_nt_schedule(2)) against the library's default
choice per epilogue (ping-pong persistent / one tile per workgroup for GELU): bit-equality, occupancy, then interleaved timing in
ONE process, per epilogue and shape, over a few start-up staggers."""
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
                               (41216, 3072, 768, 2, None), (24600, 1024, 384, 1, None), (24600, 1024, 384, 0, 1000), (24600, 1024, 768, 3, 1000),
                               (24600, 1024, 384, 2, 1000), (41216, 768, 768, 1, None), (41216, 3072, 768, 5, None)]:
        a, out, keep = mk(M, N, K, epi, nv)
        aux = keep[5]
        lib.kzv_set_nt_schedule(0); run(a, epi); torch.cuda.synchronize(); ref = out.clone(); raux = aux.clone() if aux is not None else None
        lib.kzv_set_nt_schedule(2)
        bad = 0
        for _ in range(6):
            out.zero_()
            if epi in (2, 5): aux.zero_()
            run(a, epi); torch.cuda.synchronize()
            bad += int((out != ref).sum().item()) if not torch.equal(out, ref) else 0
            if epi in (2, 5): bad += int((aux != raux).sum().item())
        print(f"check epi{epi} {M}x{N}x{K} nv={nv}: mismatching elements over 6 runs = {bad}", flush=True)
        assert bad == 0
if mode in ("time", "all"):
    def bench(a, epi, it=40):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): run(a, epi)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    shapes = [(41216, 3072, 768, 2), (41216, 3072, 768, 4), (41216, 2304, 768, 0), (41216, 768, 768, 3), (41216, 768, 3072, 3), (41216, 768, 3072, 0),
              (41216, 768, 768, 0), (41216, 768, 2304, 0)]
    staggers = [int(x) for x in os.environ.get("STAGGERS", "0,6,12").split(",")]
    for (M, N, K, epi) in shapes:
        a, out, keep = mk(M, N, K, epi)
        variants = [("default", 0, 0)] + [(f"half/{sg}us", 2, sg) for sg in staggers]
        t = {v[0]: [] for v in variants}
        for nm, sch, sg in variants:
            lib.kzv_set_nt_schedule(sch); lib.kzv_set_nt_half_stagger(sg); bench(a, epi, 10)
        for r in range(5):
            for nm, sch, sg in variants:
                lib.kzv_set_nt_schedule(sch); lib.kzv_set_nt_half_stagger(sg); t[nm].append(bench(a, epi))
        fl = 2.0 * M * N * K
        m0 = np.median(t["default"])
        line = f"time epi{epi} {M}x{N}x{K}: default {m0:7.1f} us ({fl/m0/1e6:5.0f} TF)"
        for nm, _, _ in variants[1:]:
            m = np.median(t[nm]); line += f" | {nm} {m:7.1f} ({100*(m0/m-1):+.1f} %)"
        print(line, flush=True)
lib.kzv_set_nt_schedule(-1)
