"""dev: who shares a CU, and when -- workgroup timeline of gemm_nt256h (variant library built with -DKZV_STAMPS, KZV_LIB points at it)."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import numpy as np, torch
from kzv import _lib as L
lib = L.load()
dev = "cuda"
lib.kzv_set_nt_schedule(2)
G = 512
for (M, N, K, sg) in [(41216, 2304, 768, 0), (41216, 2304, 768, 8), (41216, 768, 3072, 0)]:
    lib.kzv_set_nt_half_stagger(sg)
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(G * 16, dtype=torch.int64, device=dev)
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr(),
                           resid=None, ldr=N, aux=st.data_ptr(), ldaux=N, M=M, N=N, K=K, n_valid=N, drop_p=0.0, drop_key=5)
    for _ in range(20): lib.kzv_gemm_nt(C.byref(a), 0, torch.cuda.current_stream().cuda_stream)
    st.zero_()
    lib.kzv_gemm_nt(C.byref(a), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t = st.cpu().view(G, 16).numpy()
    hw = t[:, 15].copy(); ts = t[:, :15].astype(np.float64)
    t0 = ts[:, 0][ts[:, 0] > 0].min()
    ts = np.where(ts > 0, (ts - t0) / 100.0, np.nan)
    cu = {}
    for b in range(G):
        h = int(hw[b]) & 0xffffffff; x = int(hw[b]) >> 32
        key = (x & 0xf, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xf)
        cu.setdefault(key, []).append(b)
    sizes = collections.Counter(len(v) for v in cu.values())
    print(f"== {M}x{N}x{K} stagger {sg}: {len(cu)} distinct (xcc, se, sh, cu); workgroups per CU: {dict(sizes)}; kernel end {np.nanmax(ts):.1f} us")
    for key in list(sorted(cu))[:3] + list(sorted(cu))[-2:]:
        for b in cu[key]:
            print(f"   cu {key} block {b:3d} (xcd-local {b >> 3:2d}): " + " ".join(f"{x:6.1f}" for x in ts[b] if not np.isnan(x)))
    # K-loop and drain durations over all blocks, tiles 1.. (tile 0 includes the pipeline fill)
    kl = ts[:, 3:15:2] - ts[:, 2:14:2]; dr = ts[:, 4:15:2] - ts[:, 3:14:2]
    print(f"   K loop per tile: median {np.nanmedian(kl):.2f} us (p10 {np.nanpercentile(kl, 10):.2f}, p90 {np.nanpercentile(kl, 90):.2f}); drain: median {np.nanmedian(dr):.2f} us (p10 {np.nanpercentile(dr, 10):.2f}, p90 {np.nanpercentile(dr, 90):.2f})")
