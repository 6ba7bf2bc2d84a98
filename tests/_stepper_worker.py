"""Worker of tests/test_configs_gpu.py::test_two_rank_stepper_on_one_gpu: one rank of a data-parallel run of the REAL
kzv.trainer.Stepper (segmented backward, bucketed async all-reduce, clip after the reduce, optimizer) -- several ranks
share GPU 0 and talk over gloo (KZV_DIST_BACKEND=gloo KZV_FORCE_DEVICE=0), because a dev box has one GPU.  With WORLD_SIZE=1
and KZV_FORCE_DIST=1 the same worker is the one-rank RCCL rehearsal (test_one_rank_rccl_rehearsal_of_the_dp_branch)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")):
    sys.path.insert(0, p)

from kzv.data import build_decoder_dir, synthetic_batch  # noqa: E402
from kzv.model import TrOCRModel  # noqa: E402
from kzv.trainer import Stepper, init_distributed  # noqa: E402


def shard_batch(cfg, step, rank, world, per_rank):
    px, lab = synthetic_batch(cfg, per_rank * world, 18, seed=500 + step, min_chars=3, max_chars=17)
    sl = slice(rank * per_rank, (rank + 1) * per_rank)
    return {"pixel_values": torch.from_numpy(px[sl]), "labels": torch.from_numpy(lab[sl])}


def run_config():
    import dataclasses
    from kzv.config import tiny_config
    return dataclasses.replace(tiny_config(), enc_hidden_dropout=0.0, enc_attn_dropout=0.0, dec_hidden_dropout=0.0,
                               dec_attn_dropout=0.0)


STEPS, PER_RANK, LR, BETA2 = 12, 3, 5e-3, 0.99     # beta2 0.99: RAdam leaves its silent phase at step 6


def main():
    out = sys.argv[1]
    rank, world, local = init_distributed()
    cfg = run_config()
    d = build_decoder_dir(os.path.join(out, f"dec{rank}"), cfg)
    m = TrOCRModel(cfg.encoder_config_dict(), d, learning_rate=LR, beta2=BETA2, init_seed=4, load_tokenizer=False, device=f"cuda:{local}")
    opt = m.configure_optimizers()
    st = Stepper(m, opt, world=world, max_grad_norm=1.0, bucket_mb=0.2)        # several buckets even at tiny size
    assert len(st.buckets) >= 2, st.buckets
    m.train()
    norms = []
    for step in range(STEPS):
        st.step(shard_batch(cfg, step, rank, world, PER_RANK), step)
        norms.append(opt.grad_norm())
    torch.cuda.synchronize()
    m.logged.clear()
    m.log("rank_value", float(rank + 1), sync_dist=True)          # Lightning's sync_dist: mean over ranks (C2)
    import torch.distributed as dist
    torch.save({"params": m.flat_params.cpu(), "norms": norms, "buckets": st.buckets, "synced": m.logged["rank_value"][0],
                "backend": dist.get_backend() if dist.is_initialized() else None, "dp_path": st.dp_path,
                "nchannels": os.environ.get("NCCL_MAX_NCHANNELS")}, os.path.join(out, f"rank{rank}.pt"))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
