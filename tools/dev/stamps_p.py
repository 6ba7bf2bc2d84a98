"""dev: in-kernel timeline of the persistent gemm_nt (rebuild gemm_nt256p.hip with -DKZV_STAMPS on the GPU box, run with KZV_NT256P=1)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load()
dev = "cuda"
for (M, N, K) in [(41216, 2304, 768), (41216, 768, 768), (41216, 768, 3072)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(256 * 16, dtype=torch.int64, device=dev)
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr(),
                           resid=None, ldr=N, aux=st.data_ptr(), ldaux=N, M=M, N=N, K=K, n_valid=N, drop_p=0.0, drop_key=5)
    for _ in range(3):
        st.zero_()
        rc = lib.kzv_gemm_nt(C.byref(a), 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    import numpy as np
    def report(tag):
        t = st.cpu().view(256, 16)
        clk = t[:, 14:16].numpy().copy(); t = t.clone(); t[:, 14:] = 0
        t0 = int(t[:, 0][t[:, 0] > 0].min())
        starts = (t[:, 0].numpy() - t0) / 100.0
        ends = np.array([(max(r.tolist()) - t0) / 100.0 for r in t])
        dur = np.array([(max(r.tolist()) - r[0].item()) / 100.0 for r in t])
        ghz = (clk[:, 1] - clk[:, 0]) / np.maximum(dur, 1e-9) / 1e3
        print(f"== {tag} {M}x{N}x{K}: end min/med/max {ends.min():.1f}/{np.median(ends):.1f}/{ends.max():.1f} us; in-kernel clock med {np.median(ghz):.2f} GHz (min {ghz.min():.2f}, max {ghz.max():.2f})")
        r = t[0].tolist(); print("   block 0:", " ".join(f"{(x - t0) / 100:.1f}" for x in r if x))
    report("cold")
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): lib.kzv_gemm_nt(C.byref(a), 0, torch.cuda.current_stream().cuda_stream)
    e1.record()
    st.zero_()
    lib.kzv_gemm_nt(C.byref(a), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print(f"   event-timed: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch (200 back to back)")
    report("hot ")
