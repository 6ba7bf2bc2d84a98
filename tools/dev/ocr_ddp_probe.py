import os, sys, subprocess, socket, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "kuzushiji-vision_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch
import _ocr_ddp_worker as W
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    from kzv.trainer import init_distributed
    rank, world, local = init_distributed()
    m = W.make()
    m.train(); m.zero_grad(); m.training_step(W.shard(0, rank, world), 0); m.backward()
    own = m.flat_grads.clone()
    m.allreduce_grads()
    torch.cuda.synchronize()
    torch.save({"own": own.cpu(), "red": m.flat_grads.cpu()}, os.path.join(sys.argv[2], f"g{rank}.pt"))
    torch.distributed.barrier(); torch.distributed.destroy_process_group(); sys.exit(0)
tmp = tempfile.mkdtemp()
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
procs = []
for rank in range(2):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KZV_DIST_BACKEND="gloo", KZV_FORCE_DEVICE="0")
    procs.append(subprocess.Popen([sys.executable, __file__, "worker", tmp], env=env))
for p in procs: p.wait()
g0, g1 = torch.load(os.path.join(tmp, "g0.pt")), torch.load(os.path.join(tmp, "g1.pt"))
m = W.make()
def grads(rank):
    m.train(); m.zero_grad(); m.training_step(W.shard(0, rank, 2), 0); m.backward(); torch.cuda.synchronize(); return m.flat_grads.cpu().clone()
c_first = grads(1)
a = grads(0)
c = grads(1)
m2 = W.make()
def grads2(rank):
    m2.train(); m2.zero_grad(); m2.training_step(W.shard(0, rank, 2), 0); m2.backward(); torch.cuda.synchronize(); return m2.flat_grads.cpu().clone()
c2 = grads2(1)
print("own1 vs first-call emu1", float((g1["own"] - c_first).norm() / c_first.norm()), " vs after-shard0 emu1", float((g1["own"] - c).norm() / c.norm()), " vs fresh-model emu1", float((g1["own"] - c2).norm() / c2.norm()))
for name in m.offsets:
    x, y = m._view(g1["own"], name), m._view(c, name)
    r = float((x - y).norm() / (y.norm() + 1e-30))
    if r > 1e-4: print(f"   {name:45s} rel {r:.3e}")
print("own0 vs emu0", float((g0["own"] - a).norm() / a.norm()), " own1 vs emu1", float((g1["own"] - c).norm() / c.norm()))
print("reduced equal on ranks", torch.equal(g0["red"], g1["red"]), " reduced vs (a+c)/2", float((g0["red"] - (a + c) / 2).norm() / ((a + c) / 2).norm()))
