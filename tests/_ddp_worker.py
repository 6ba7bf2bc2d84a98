"""Worker of tests/test_host_cpu.py::test_ddp_mean_of_rank_means_two_ranks_gloo (CPU, gloo)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")):
    sys.path.insert(0, p)

from kzv import params as P  # noqa: E402
from kzv.config import tiny_config  # noqa: E402
from kzv.data import synthetic_batch  # noqa: E402
from kzv.trainer import bucket_plan  # noqa: E402
from oracle import trocr_oracle as O  # noqa: E402


def flat_grads(cfg, sd_np, px, lab):
    r = O.forward_backward(cfg, sd_np, px, lab)
    offs, total = P.param_offsets(cfg)
    flat = np.zeros(total, dtype=np.float32)
    for hf, eng, rel, shape in P.hf_views(cfg):
        if hf in P.TIED_ALIASES:
            continue
        g = r["grads"][hf]
        base = offs[eng][0] + rel
        flat[base:base + g.size] = g.reshape(-1)
    return flat, r["loss"]


def main():
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = tiny_config()
    sd = P.state_dict_from_flat(cfg, P.recipe_flat(cfg, 9))
    px, lab = synthetic_batch(cfg, 4, 16, seed=5, min_chars=2, max_chars=15)
    per = px.shape[0] // world
    mine, _ = flat_grads(cfg, sd, px[rank * per:(rank + 1) * per], lab[rank * per:(rank + 1) * per])
    g = torch.from_numpy(mine)
    # segment ranges exactly as libkzv reports them (decoder tail first, encoder layers descending, embeddings)
    offs, total = P.param_offsets(cfg)
    lnf = offs["enc.lnf.w"][0]
    segs = [(lnf, total)]
    for i in reversed(range(cfg.enc_layers)):
        hi = offs[f"enc.{i + 1}.ln1.w"][0] if i + 1 < cfg.enc_layers else lnf
        segs.append((offs[f"enc.{i}.ln1.w"][0], hi))
    segs.append((0, offs["enc.0.ln1.w"][0]))
    works = [dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for _, lo, hi in bucket_plan(segs, 50_000)]
    for w in works:
        w.wait()
    g /= world
    np.save(os.path.join(out, f"rank{rank}.npy"), g.numpy())
    if rank == 0:   # reference: gradient of the mean of the per-rank mean losses
        ref = sum(flat_grads(cfg, sd, px[r * per:(r + 1) * per], lab[r * per:(r + 1) * per])[0] for r in range(world)) / world
        np.save(os.path.join(out, "ref.npy"), ref)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
