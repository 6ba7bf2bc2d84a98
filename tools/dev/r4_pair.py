"""dev: the dgrad + wgrad pair launch (kzv_gemm_dgrad_wgrad) against the two separate launches: results, then interleaved timing."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import numpy as np, torch
from kzv import _lib as L
lib = L.load()
dev = "cuda"
def st(): return torch.cuda.current_stream().cuda_stream
M = 41216
# (name, Nout, Kin, epilogue): dX[M, Kin] = dY[M, Nout] . W[Nout, Kin];  dW[Nout, Kin] += dY^T X
cases = [("fc2", 768, 3072, 4), ("fc1", 3072, 768, 0), ("o", 768, 768, 0), ("qkv", 2304, 768, 0)]
def mk(Nout, Kin, epi):
    dY = torch.randn(M, Nout, device=dev).bfloat16(); X = torch.randn(M, Kin, device=dev).bfloat16()
    Wt = (torch.randn(Kin, Nout, device=dev) * 0.05).bfloat16()         # transposed weight copy [Kin, Nout]
    dX = torch.empty(M, Kin, dtype=torch.bfloat16, device=dev)
    aux = (torch.rand(M, Kin, device=dev) * 1.2).bfloat16() if epi == 4 else None
    dW = torch.zeros(Nout, Kin, device=dev); db = torch.zeros(Nout, device=dev)
    na = L.kzv_gemm_nt_args(A=dY.data_ptr(), lda=Nout, B=Wt.data_ptr(), ldb=Nout, C=dX.data_ptr(), ldc=Kin, bias=None, resid=None, ldr=Kin,
                            aux=L.ptr(aux), ldaux=Kin, M=M, N=Kin, K=Nout, n_valid=Kin, drop_p=0.0, drop_key=0)
    ta = L.kzv_gemm_tn_args(P=dY.data_ptr(), ldp=Nout, Q=X.data_ptr(), ldq=Kin, OUT=dW.data_ptr(), ldo=Kin, Mtok=M, N=Nout, K=Kin, n_store=Nout, dbias=db.data_ptr())
    return na, ta, dX, dW, db, (dY, X, Wt, aux)
def run(na, ta, epi): L.check(lib.kzv_gemm_dgrad_wgrad(C.byref(na), epi, C.byref(ta), st()))
for name, Nout, Kin, epi in (cases if 'time' not in sys.argv else []):
    na, ta, dX, dW, db, keep = mk(Nout, Kin, epi)
    lib.kzv_set_pair(0); dW.zero_(); db.zero_(); run(na, ta, epi); torch.cuda.synchronize(); rX, rW, rb = dX.clone(), dW.clone(), db.clone()
    lib.kzv_set_pair(1)
    for rep in range(3):
        dX.zero_(); dW.zero_(); db.zero_(); run(na, ta, epi); torch.cuda.synchronize()
        assert torch.equal(dX, rX), f"{name}: dgrad differs"
        want = keep[0].float().t() @ keep[1].float()
        e_pair = float((dW - want).abs().max() / want.abs().max()); e_sep = float((rW - want).abs().max() / want.abs().max())
        eb = float((db - rb).abs().max() / (rb.abs().max() + 1e-9))
        assert e_pair < 4e-3 and eb < 1e-5, (name, e_pair, eb)
    print(f"check {name}: dgrad bit-identical; wgrad rel err pair {e_pair:.2e} / separate {e_sep:.2e}; bias rel diff {eb:.1e}", flush=True)
def bench(na, ta, epi, it=30):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): run(na, ta, epi)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
tot = [0.0, 0.0]
for name, Nout, Kin, epi in cases:
    na, ta, dX, dW, db, keep = mk(Nout, Kin, epi)
    for s in (0, 1): lib.kzv_set_pair(s); bench(na, ta, epi, 8)
    t = {0: [], 1: []}
    for r in range(5):
        for s in (0, 1):
            lib.kzv_set_pair(s); t[s].append(bench(na, ta, epi))
    m0, m1 = np.median(t[0]), np.median(t[1]); tot[0] += m0; tot[1] += m1
    print(f"time {name}: separate {m0:7.1f} us  pair {m1:7.1f} us  {m0 - m1:+6.1f} us ({100*(m0/m1-1):+.1f} %)  [incl. the fold launch]", flush=True)
print(f"per encoder layer: separate {tot[0]:.0f} us, pair {tot[1]:.0f} us: {tot[0]-tot[1]:+.0f} us -> x12 layers = {(tot[0]-tot[1])*12/1e3:+.2f} ms per step")
lib.kzv_set_pair(-1)
