"""dev: per-workgroup phase timeline of gemm_nt from a -DKZV_STAMPS build (see gemm.hip KZV_STAMP)."""
import ctypes as C, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
L.LIB_PATH = "/tmp/kzv_stamps/libkzv.so"
lib = L.load(); dev = "cuda"
for (M, N, K, epi) in [(41216, 2304, 768, 0), (41216, 3072, 768, 2), (41216, 768, 3072, 0)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=B.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr(), resid=0, ldr=N,
                           aux=aux.data_ptr(), ldaux=N, M=M, N=N, K=K, n_valid=N, drop_p=0.0, drop_key=0)
    for _ in range(3):
        L.check(lib.kzv_gemm_nt(C.byref(a), epi, torch.cuda.current_stream().cuda_stream))
    st = np.fromfile(os.path.join(ROOT, "gpurun_out", "stamps.bin"), dtype=np.uint64).reshape(-1, 8).astype(np.float64) / 100.0  # us
    d = {"iter5: vmcnt wait": st[:, 3] - st[:, 2], "barrier": st[:, 4] - st[:, 3], "issue next stage": st[:, 5] - st[:, 4],
         "16 lds reads + 32 mfma": st[:, 6] - st[:, 5], "whole iteration": st[:, 6] - st[:, 2], "whole wg": st[:, 7] - st[:, 0]}
    print(f"== M{M} N{N} K{K} epi{epi}")
    for k, v in d.items():
        print(f"  {k:24s} median {np.median(v):8.2f}  p10 {np.percentile(v, 10):8.2f}  p90 {np.percentile(v, 90):8.2f} us")
