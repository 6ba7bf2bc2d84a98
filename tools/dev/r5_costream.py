"""dev: an encoder layer's BACKWARD with the four weight gradients on a side stream, both sides on a limited number of CUs
(KZV_NT_GRID for the persistent gemm_nt, KZV_TN_CUS for gemm_tn256's token splits): does the chip do more per microsecond when
MFMA-bound and HBM-bound kernels run side by side?  Dependencies between the two sides are ignored (timing only)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
B, S, H, F, heads = 256, 161, 768, 3072, 12
M = B * S
g = lambda *s: torch.randn(*s, device=dev)
bf = lambda *s: g(*s).bfloat16()
w = lambda n, k: (g(n, k) * 0.02).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

# ---- main-stream operands
dy768 = bf(M, H); dh = torch.empty(M, F, dtype=torch.bfloat16, device=dev); aux = bf(M, F); dx768 = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
dqkv = bf(M, 3 * H); w2t, w1t, wot, wqkvt = w(F, H), w(H, F), w(H, H), w(H, 3 * H)
x = g(M, H); stats = torch.stack([x.mean(1), 1.0 / x.std(1)], 1).contiguous(); gamma = torch.ones(H, device=dev); dres = g(M, H); dgam = torch.zeros(H, device=dev); dbet = torch.zeros(H, device=dev)
qkv = bf(M, 3 * H); o = bf(M, H); lse = g(B, heads, S) + 5.0


def gemm(s, A, K, W, N, out, epi, aux_=None):
    a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=K, B=W.data_ptr(), ldb=K, C=out.data_ptr(), ldc=N, aux=L.ptr(aux_), ldaux=N, M=M, N=N, K=K, n_valid=N)
    L.check(lib.kzv_gemm_nt(C.byref(a), epi, s))
def lnb(s):
    L.check(lib.kzv_layernorm_bwd(dx768.data_ptr(), 0, x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dres.data_ptr(), 1, dgam.data_ptr(), dbet.data_ptr(), M, H, s))
at = L.kzv_attn_args(Q=qkv.data_ptr(), K=qkv[:, H:].data_ptr(), V=qkv[:, 2 * H:].data_ptr(), O=o.data_ptr(), LSE=lse.data_ptr(), dO=dy768.data_ptr(),
                     dQ=dqkv.data_ptr(), dK=dqkv[:, H:].data_ptr(), dV=dqkv[:, 2 * H:].data_ptr(), ldq=3 * H, ldk=3 * H, ldv=3 * H, ldo=H, B=B, heads=heads, Sq=S, Sk=S, mode=0,
                     drop_p=0.1, drop_key=7)
def attnb(s): L.check(lib.kzv_attn_bwd(C.byref(at), s))
def main_gemms(s):
    gemm(s, dy768, H, w2t, F, dh, 4, aux); gemm(s, dh, F, w1t, H, dx768, 0); gemm(s, dy768, H, wot, H, dx768, 0); gemm(s, dqkv, 3 * H, wqkvt, H, dx768, 0)
def main_layer(s):
    gemm(s, dy768, H, w2t, F, dh, 4, aux); gemm(s, dh, F, w1t, H, dx768, 0); lnb(s); gemm(s, dy768, H, wot, H, dx768, 0); attnb(s); gemm(s, dqkv, 3 * H, wqkvt, H, dx768, 0); lnb(s)
def main_hbm(s):
    lnb(s); attnb(s); lnb(s)

# ---- side-stream operands: the four weight gradients
xin768 = bf(M, H); hact = bf(M, F)
outs = [torch.zeros(3 * H, H, device=dev), torch.zeros(H, H, device=dev), torch.zeros(F, H, device=dev), torch.zeros(H, F, device=dev)]
def tn(s, P, N, Q, K, O):
    a = L.kzv_gemm_tn_args(P=P.data_ptr(), ldp=N, Q=Q.data_ptr(), ldq=K, OUT=O.data_ptr(), ldo=K, Mtok=M, N=N, K=K, n_store=N)
    L.check(lib.kzv_gemm_tn(C.byref(a), s))
def side_layer(s):
    tn(s, dy768, H, hact, F, outs[3]); tn(s, dh, F, xin768, H, outs[2]); tn(s, dy768, H, xin768, H, outs[1]); tn(s, dqkv, 3 * H, xin768, H, outs[0])


def wall(f_main, f_side, reps=12, it=3):
    best = 1e9
    for _ in range(it + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            if f_main: f_main(s1.cuda_stream)
            if f_side: f_side(s2.cuda_stream)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best

def setenv(nt, tncus):
    for k, v in (("KZV_NT_GRID", nt), ("KZV_TN_CUS", tncus)):
        if v: os.environ[k] = str(v)
        else: os.environ.pop(k, None)

for f in (main_layer, side_layer):          # warm both streams / queues
    f(s1.cuda_stream); f(s2.cuda_stream)
torch.cuda.synchronize()
setenv(0, 0)
tm, tg, th, ts = wall(main_layer, None), wall(main_gemms, None), wall(main_hbm, None), wall(None, side_layer)
print(f"full chip, alone, per layer: main chain {tm:.0f} us (its 4 gemm_nt {tg:.0f}, ln_bwd x2 + attn_bwd {th:.0f}), 4 weight gradients {ts:.0f}; serial sum {tm + ts:.0f}", flush=True)
for (nt, tc) in ((0, 0), (192, 64), (160, 96), (128, 128), (96, 160), (64, 192)):
    setenv(nt, tc)
    a, b = wall(main_layer, None), wall(None, side_layer)
    both = wall(main_layer, side_layer)
    hb = wall(main_hbm, side_layer)
    gb = wall(main_gemms, side_layer)
    print(f"nt grid {nt or 256:3d} / tn CUs {tc or 256:3d}: main alone {a:.0f}, side alone {b:.0f}, BOTH {both:.0f} ({both / (tm + ts):.3f} of serial) | hbm-part||side {hb:.0f} (alone {th:.0f} + {b:.0f}) | gemm-part||side {gb:.0f}", flush=True)
