"""Training runtime standing in for the pieces of ``pl.Trainer`` the reference uses on this path
(scripts/train_trocr.py:165-176): one process per GPU, gradient all-reduce (mean) over RCCL overlapped with
backward, ``gradient_clip_val=1.0`` applied AFTER the reduce on the identical full gradient, optimizer step,
half-epoch validation, rank-0 logging/checkpoints.

DDP semantics reproduced: each rank's loss is the mean over ITS non-pad tokens; gradients are
summed across ranks and divided by world size; dropout streams differ per rank (seed mixes the rank).
"""
from __future__ import annotations

import ctypes as C
import os
import time

from . import _lib as L


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def dp_forced() -> bool:
    """Rehearsal switch: run the world > 1 code path (init_distributed, Stepper.step) with whatever world size there is."""
    return os.environ.get("KZV_FORCE_DIST", "0") not in ("", "0")


def init_distributed(backend: str | None = None):
    """torch.distributed over RCCL ("nccl" on ROCm) for GPUs, gloo on CPU.  No-op for world size 1 (unless KZV_FORCE_DIST)."""
    import torch
    import torch.distributed as dist
    rank, world, local = dist_env()
    # rehearsal knobs (1-GPU dev boxes): several ranks on one device over gloo
    if os.environ.get("KZV_FORCE_DEVICE") is not None:
        local = int(os.environ["KZV_FORCE_DEVICE"])
    backend = backend or os.environ.get("KZV_DIST_BACKEND")
    # KZV_FORCE_DIST=1: build the process group even for ONE rank, so that a 1-GPU box executes the data-parallel branch
    # end to end (RCCL communicator on the device, async all-reduce on RCCL's stream, CU reserve, segmented backward)
    if (world > 1 or dp_forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # RCCL's channel workgroups sit on CUs for the whole collective; keep them few (393 MB of gradients per
            # ~35 ms step needs little bandwidth) and tell the GEMM launchers to leave that many CUs alone, or the
            # one-workgroup-per-CU kernels would run a second round for the workgroups that found their CU taken
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
            if os.environ.get("KZV_CU_RESERVE") is None:
                L.check(L.load().kzv_set_cu_reserve(32), "set_cu_reserve")
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def bucket_plan(seg_ranges, bucket_elems: int):
    """Merge consecutive backward segments (in completion order) into all-reduce buckets of at least
    ``bucket_elems`` fp32 elements.  Returns [(last_segment_index, lo, hi)] -- a bucket is launched once its
    last segment has been enqueued.  Ranges of merged segments are contiguous by construction of the
    parameter table (decoder at the tail, encoder layers in descending order, embeddings at the head)."""
    buckets = []
    cur_lo = cur_hi = None
    for i, (lo, hi) in enumerate(seg_ranges):
        if cur_lo is None:
            cur_lo, cur_hi = lo, hi
        else:
            if hi != cur_lo and lo != cur_hi:
                raise ValueError("segments are not contiguous in completion order")
            cur_lo, cur_hi = min(lo, cur_lo), max(hi, cur_hi)
        if cur_hi - cur_lo >= bucket_elems or i == len(seg_ranges) - 1:
            buckets.append((i, cur_lo, cur_hi))
            cur_lo = cur_hi = None
    return buckets


class Stepper:
    """One training step of the hot path: forward+loss, segmented backward with overlapped bucketed
    all-reduce, clip, RAdamScheduleFree step, bf16 weight refresh."""

    def __init__(self, model, optimizer, world: int = 1, max_grad_norm: float = 1.0, bucket_mb: float = 64.0,
                 dp_path: bool | None = None):
        self.model, self.opt, self.world, self.max_grad_norm = model, optimizer, world, max_grad_norm
        # the data-parallel branch of step(); with one rank it is a rehearsal (a 1-rank all-reduce is the identity)
        self.dp_path = (world > 1 or dp_forced()) if dp_path is None else bool(dp_path)
        lib = L.load()
        n = lib.kzv_backward_segments(model._h)
        self.seg_ranges = []
        for s in range(n):
            lo, hi = C.c_int64(), C.c_int64()
            L.check(lib.kzv_backward_segment_range(model._h, s, C.byref(lo), C.byref(hi)), "segment_range")
            self.seg_ranges.append((lo.value, hi.value))
        # xGMI is point-to-point (7 links x ~153 GB/s): few large buckets keep every link busy and the launch
        # count low; 64 MB fp32 buckets -> ~7 collectives per 393 MB gradient set.
        self.buckets = bucket_plan(self.seg_ranges, int(bucket_mb * 1024 * 1024 / 4))

    def step(self, batch, batch_idx: int = 0):
        m, lib = self.model, L.load()
        loss, _ = m.forward_loss(batch["pixel_values"], batch["labels"], want_logits=False)
        st = L.stream_handle()
        L.check(lib.kzv_zero_grads(m._h, st), "zero_grads")
        works = []
        if not self.dp_path:     # no consumer between segments: let the engine overlap across them
            L.check(lib.kzv_backward(m._h, st), "backward")
            self.opt.step(max_grad_norm=self.max_grad_norm, grad_scale=1.0)
            return loss
        import torch.distributed as dist
        bi = 0
        for s in range(len(self.seg_ranges)):
            L.check(lib.kzv_backward_segment(m._h, s, st), "backward_segment")
            if s == self.buckets[bi][0]:
                _, lo, hi = self.buckets[bi]
                # async: RCCL waits for the kernels enqueued so far on this stream, then runs on its own
                # stream while the remaining backward segments keep the compute stream busy
                works.append(dist.all_reduce(m.flat_grads[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
                bi += 1
        for w in works:
            w.wait()
        self.opt.step(max_grad_norm=self.max_grad_norm, grad_scale=1.0 / self.world)
        return loss


def fit(model, train_loader, val_loader=None, max_epochs: int = 1, max_steps: int = -1, log_every: int = 50,
        val_check_interval: float = 0.5, ckpt_dir: str | None = None, world: int = 1, rank: int = 0, log=print,
        callbacks=()):
    """Minimal Trainer.fit: epochs over the loader, validation every ``val_check_interval`` of an epoch
    (scripts/train_trocr.py:174), rank-0 checkpoints named like the reference (:138)."""
    import torch
    opt = model.configure_optimizers()
    stepper = Stepper(model, opt, world=world)
    for cb in callbacks:          # e.g. kzv.ema.EMACallback (scripts/train_trocr.py:154,182)
        cb.on_fit_start(model)
    gstep = 0
    history = []
    best, save_top_k = [], 3          # (val_loss, path) of the kept checkpoints
    fit.best_model_path = None
    n_batches = len(train_loader)
    val_every = max(1, int(n_batches * val_check_interval)) if val_loader is not None else 0
    for epoch in range(max_epochs):
        if hasattr(train_loader, "set_epoch"):
            train_loader.set_epoch(epoch)          # DataLoader(shuffle=True) / DistributedSampler.set_epoch: a new order per epoch
        model.train()
        model.on_train_epoch_start()
        t0 = time.time()
        for i, batch in enumerate(train_loader):
            loss = stepper.step(batch, i)
            gstep += 1
            for cb in callbacks:
                cb.on_train_batch_end(model)
            if gstep % log_every == 0 or gstep == 1:
                lv = float(loss.item())
                history.append((gstep, lv))
                if rank == 0:
                    log(f"epoch {epoch} step {gstep} train_loss {lv:.4f} lr {opt.scheduled_lr:.3g}")
            if val_every and (i + 1) % val_every == 0:
                model.on_validation_epoch_start()
                for cb in callbacks:
                    cb.on_validation_start(model)
                vals = [model.validation_step(vb, j) for j, vb in enumerate(val_loader)]
                for cb in callbacks:
                    cb.on_validation_end(model)
                model.on_validation_epoch_end()
                model.train()
                vl = sum(vals) / max(1, len(vals))
                if rank == 0:
                    log(f"epoch {epoch} step {gstep} val_loss {vl:.4f}")
                    if ckpt_dir:
                        # ModelCheckpoint(monitor="val_loss", save_top_k=3) of scripts/train_trocr.py:136-143: keep the three
                        # best, write only when the new one ranks among them
                        if len(best) < save_top_k or vl < best[-1][0]:
                            os.makedirs(ckpt_dir, exist_ok=True)
                            path = os.path.join(ckpt_dir, f"trocr-epoch={epoch:02d}-val_loss={vl:.2f}.ckpt")
                            k = 1
                            while os.path.exists(path):        # Lightning's -v1, -v2 suffixes for equal names
                                path = os.path.join(ckpt_dir, f"trocr-epoch={epoch:02d}-val_loss={vl:.2f}-v{k}.ckpt"); k += 1
                            save_checkpoint(model, opt, path, epoch, gstep, callbacks)
                            best.append((vl, path)); best.sort(key=lambda e: e[0])
                            for _, stale in best[save_top_k:]:
                                if os.path.exists(stale):
                                    os.remove(stale)
                            del best[save_top_k:]
                            fit.best_model_path = best[0][1]
            if 0 < max_steps <= gstep:
                break
        torch.cuda.synchronize()
        if rank == 0:
            log(f"epoch {epoch} done in {time.time() - t0:.1f}s")
        if 0 < max_steps <= gstep:
            break
    if ckpt_dir and rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
        save_checkpoint(model, opt, os.path.join(ckpt_dir, "last.ckpt"), max_epochs - 1, gstep, callbacks)
        if fit.best_model_path is None:            # no validation ran: "best" falls back to the last weights, like Lightning
            fit.best_model_path = os.path.join(ckpt_dir, "last.ckpt")
    return history


def save_checkpoint(model, opt, path: str, epoch: int, global_step: int, callbacks=(), spelling: str = "hf4") -> None:
    """What Lightning's ModelCheckpoint writes for the reference (scripts/train_trocr.py:136-143): state_dict under HF
    names in the reference's order, schedulefree's per-parameter optimizer state, hyper_parameters -- kzv/checkpoint.py."""
    import torch
    from .checkpoint import build_checkpoint
    ck = build_checkpoint(model.cfg, model.flat_params.detach().cpu(), vars(model.hparams), epoch, global_step,
                          optimizer=opt.state_dict() if opt is not None else None, spelling=spelling)
    for cb in callbacks:
        ck = cb.on_save_checkpoint(model, ck)
    torch.save(ck, path)


def load_checkpoint(model, opt, path: str, callbacks=()):
    """Load a checkpoint written by this engine OR by the reference under Lightning (either ViT key spelling; optimizer
    state mapped through the checkpoint's own parameter order)."""
    import torch
    from .checkpoint import read_checkpoint
    ck = torch.load(path, map_location="cpu", weights_only=False)
    sd, osd = read_checkpoint(ck, model.cfg)
    model.load_state_dict(sd, strict=True)
    if opt is not None and osd is not None:
        opt.load_state_dict(osd)
    for cb in callbacks:
        cb.on_load_checkpoint(model, ck)
    return ck


def test(model, test_loader, ckpt_path: str | None = None, log=print, rank: int = 0):
    """``trainer.test(model, test_loader, ckpt_path="best")`` (scripts/train_trocr.py:193-195): load the best checkpoint,
    switch the optimizer to its eval parameters (on_test_epoch_start, trocr_model.py:441-445), run test_step over the
    loader, report the epoch means of test_loss / test_cer."""
    if ckpt_path:
        load_checkpoint(model, model.optimizers(), ckpt_path)
    model.eval()
    model.on_test_epoch_start()
    model.logged.pop("test_loss", None); model.logged.pop("test_cer", None)
    for j, b in enumerate(test_loader):
        model.test_step(b, j)
    if model.optimizers() is not None:
        model.optimizers().train()
    out = {k: (sum(model.logged[k]) / len(model.logged[k]) if model.logged.get(k) else float("nan")) for k in ("test_loss", "test_cer")}
    if rank == 0:
        log(f"test_loss {out['test_loss']:.4f} test_cer {out['test_cer']:.4f}")
    return out
