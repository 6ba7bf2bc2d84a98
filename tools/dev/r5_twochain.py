"""dev: does splitting the batch into two half-batch chains on two streams (persistent GEMMs on half the CUs each) overlap one
chain's HBM-bound kernels with the other's MFMA-bound ones?  Encoder-layer FORWARD sequence through the per-op C ABI.
usage: r5_twochain.py one|two [layers] [offset_us]   (KZV_NT_GRID=128 in the environment for `two`)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
mode = sys.argv[1]; layers = int(sys.argv[2]) if len(sys.argv) > 2 else 12; offset_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
S, H, F, heads = 161, 768, 3072, 12


class Chain:
    def __init__(self, B):
        M = B * S; self.B = B; self.M = M
        g = lambda *s: torch.randn(*s, device=dev)
        self.x = g(M, H); self.x2 = torch.empty(M, H, device=dev)
        self.a = torch.empty(M, H, dtype=torch.bfloat16, device=dev); self.st = torch.empty(M, 2, device=dev)
        self.qkv = torch.empty(M, 3 * H, dtype=torch.bfloat16, device=dev); self.o = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
        self.lse = torch.empty(B, heads, S, device=dev)
        self.h = torch.empty(M, F, dtype=torch.bfloat16, device=dev); self.aux = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
        self.gamma = torch.ones(H, device=dev); self.beta = torch.zeros(H, device=dev)
        w = lambda n, k: (g(n, k) * 0.02).bfloat16()
        self.wqkv, self.wo, self.w1, self.w2 = w(3 * H, H), w(H, H), w(F, H), w(H, F)
        self.bqkv, self.bo, self.b1, self.b2 = g(3 * H) * 0.01, g(H) * 0.01, g(F) * 0.01, g(H) * 0.01

    def gemm(self, s, A, K, W, N, out, bias, epi, resid=None, aux=None, key=3):
        a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=W.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, bias=bias.data_ptr(), resid=L.ptr(resid), ldr=N,
                               aux=L.ptr(aux), ldaux=N, M=self.M, N=N, K=K, n_valid=N, drop_p=0.1 if epi == 3 else 0.0, drop_key=key)
        L.check(lib.kzv_gemm_nt(C.byref(a), epi, s))

    def ln(self, s, x, y):
        L.check(lib.kzv_layernorm_fwd(x.data_ptr(), self.gamma.data_ptr(), self.beta.data_ptr(), y.data_ptr(), 0, self.st.data_ptr(), self.M, H, 1e-12, s))

    def steps(self, s):
        """the seven launches of one ViTLayer forward, as thunks"""
        q = self.qkv
        at = L.kzv_attn_args(Q=q.data_ptr(), K=q[:, H:].data_ptr(), V=q[:, 2 * H:].data_ptr(), O=self.o.data_ptr(), LSE=self.lse.data_ptr(), ldq=3 * H, ldk=3 * H,
                             ldv=3 * H, ldo=H, B=self.B, heads=heads, Sq=S, Sk=S, mode=0, drop_p=0.0, drop_key=7)
        return [lambda: self.ln(s, self.x, self.a),
                lambda: self.gemm(s, self.a, H, self.wqkv, 3 * H, self.qkv, self.bqkv, 0),
                lambda: L.check(lib.kzv_attn_fwd(C.byref(at), s)),
                lambda: self.gemm(s, self.o, H, self.wo, H, self.x2, self.bo, 3, resid=self.x),
                lambda: self.ln(s, self.x2, self.a),
                lambda: self.gemm(s, self.a, H, self.w1, F, self.h, self.b1, 2, aux=self.aux),
                lambda: self.gemm(s, self.h, F, self.w2, H, self.x, self.b2, 3, resid=self.x2)]


def run(chains, streams, reps):
    thunks = [c.steps(s.cuda_stream) for c, s in zip(chains, streams)]
    def once():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        if len(streams) > 1 and offset_us > 0:
            with torch.cuda.stream(streams[1]): torch.cuda._sleep(int(offset_us * 2100))     # ~2.1 GHz
        for _ in range(layers):
            for i in range(7):
                for t in thunks: t[i]()
        for s in streams: cur.wait_stream(s)
    for _ in range(2): once()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if mode == "one":
    c = Chain(256); us = run([c], [torch.cuda.Stream()], 5)
    print(f"one chain B=256: {us:.0f} us for {layers} layers = {us / layers:.1f} us per layer")
elif mode == "half":       # one half-batch chain alone (what a chain costs without a partner)
    c = Chain(128); us = run([c], [torch.cuda.Stream()], 5)
    print(f"one chain B=128 alone (KZV_NT_GRID={os.environ.get('KZV_NT_GRID')}): {us:.0f} us = {us / layers:.1f} us per layer")
else:
    cs = [Chain(128), Chain(128)]; us = run(cs, [torch.cuda.Stream(), torch.cuda.Stream()], 5)
    print(f"two chains B=128 (KZV_NT_GRID={os.environ.get('KZV_NT_GRID')}, offset {offset_us} us): {us:.0f} us = {us / layers:.1f} us per layer (both)")
