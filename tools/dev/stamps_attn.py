"""dev: workgroup timeline of attention forward (rebuild attention.hip with -DKZV_STAMPS on the GPU box)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kuzushiji-vision_amd"))
import torch
from kzv import _lib as L
lib = L.load(); dev = "cuda"
B, heads, Sq, Sk = 256, 12, 161, 161
H = heads * 64
qkv = torch.randn(B * Sq, 3 * H, device=dev).bfloat16(); q, k, v = qkv, qkv[:, H:], qkv[:, 2 * H:]
o = torch.empty(B * Sq, H, dtype=torch.bfloat16, device=dev); lse = torch.empty(B, heads, Sq, device=dev)
st = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
a = L.kzv_attn_args(Q=q.data_ptr(), K=k.data_ptr(), V=v.data_ptr(), O=o.data_ptr(), LSE=lse.data_ptr(), dO=None, dQ=st.data_ptr(), dK=None, dV=None,
                    ldq=3 * H, ldk=3 * H, ldv=3 * H, ldo=H, ids=None, ld_ids=0, pad_id=1, B=B, heads=heads, Sq=Sq, Sk=Sk, mode=0, drop_p=0.1, drop_key=7)
for _ in range(3):
    st.zero_(); L.check(lib.kzv_attn_fwd(C.byref(a), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
t = st.cpu().view(16, 8)
t0 = int(t[:, 0][t[:, 0] > 0].min())
for i in range(12):
    r = [x for x in t[i].tolist() if x]
    print(f"block {i * 257:5d}: " + " ".join(f"{(x - t0) / 100:7.2f}" for x in r))

# ---- backward timeline of one workgroup (block 771, thread 0): start, DMA issued, delta done, staged | per q-block: after each
# owned key tile, phase A end, barrier, dQ MFMAs done, phase B end, barrier
do = torch.randn(B * Sq, H, device=dev).bfloat16(); dqkv = torch.zeros_like(qkv)
a.dO = do.data_ptr(); a.dQ = dqkv.data_ptr(); a.dK = dqkv[:, H:].data_ptr(); a.dV = dqkv[:, 2 * H:].data_ptr()
for _ in range(3): L.check(lib.kzv_attn_bwd(C.byref(a), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
raw = C.CDLL(L.__dict__.get("_path", None) or os.path.join(ROOT, "kuzushiji-vision_amd", "kzv", "libkzv.so"))
assert raw.kzv_debug_bwd_stamps(buf) == 0
r = [x for x in buf if x]
print("bwd block 771: " + " ".join(f"{(x - r[0]) / 100:.2f}" for x in r))
