#!/usr/bin/env python3
"""Benchmark of the hot path: TrOCR line-OCR TRAINING step (forward + CE + hand-written backward + gradient
all-reduce + clip 1.0 + RAdamScheduleFree + bf16 weight refresh), synthetic 64x640 crops, dropout ON.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  value = line-images/s of the whole job (all ranks), inputs resident in HBM.
roofline: the bf16 MFMA GEMM family gemm_nt_kernel<*> (every nn.Linear forward and input-gradient), timed live
with HIP events inside the timed region (kzv_prof_*: every s-th launch is bracketed -- a bracket costs ~2 us of stream
time; s is the first of 4, 5, 3, 7 co-prime to the family's launches per step, counted in one untimed step, so that every
launch site is sampled equally often), priced by its
algorithmic 2*M*N*K FLOPs against the 2.5 PFLOP/s dense bf16 peak of /opt/skills/guides/MI355X_MICROARCH.md.
cpu_baseline: oracle/trocr_oracle.py (a port, not the reference) timed on this box's host cores, rank 0, N=1.
"""
from __future__ import annotations

import argparse
import ctypes as C
import dataclasses
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 (block-scaled MFMA), same table


def train_flops_per_image(cfg, T):
    """ALGORITHMIC FLOPs per image (SURVEY.md section 8(d) formulae): 2*MACs, full S^2 attention, train = 3 x fwd."""
    S, H, F = cfg.enc_seq, cfg.enc_hidden, cfg.enc_ffn
    enc_layer = 2 * S * H * H * 3 + 2 * S * S * H * 2 + 2 * S * H * H + 2 * S * H * F * 2
    patch = 2 * cfg.num_patches * cfg.patch_dim * H
    proj = 2 * cfg.num_patches * H * cfg.dec_hidden if cfg.has_proj else 0
    Hd, Fd, Sk = cfg.dec_hidden, cfg.dec_ffn, cfg.num_patches
    dec_layer = (2 * T * Hd * Hd * 3 + 2 * T * T * Hd * 2 + 2 * T * Hd * Hd
                 + 2 * T * Hd * Hd + 2 * Sk * Hd * Hd * 2 + 2 * T * Sk * Hd * 2 + 2 * T * Hd * Hd
                 + 2 * T * Hd * Fd * 2)
    head = 2 * T * Hd * Hd + 2 * T * Hd * cfg.vocab
    fwd = patch + cfg.enc_layers * enc_layer + proj + cfg.dec_layers * dec_layer + head
    return 3.0 * fwd


def source_sha16():
    """Identity of the kernel sources a profile belongs to (tools/hbm_traffic.py stores it; bench.py only quotes
    `traffic` from a profile taken on the SAME sources)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "kuzushiji-vision_amd", "csrc", "*.[hc]*"))) + [os.path.join(ROOT, "include", "kzv.h"),
                                                                                                  os.path.join(ROOT, "kuzushiji-vision_amd", "csrc", "Makefile")]:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(cfg, label_len, sample_batch, steps, budget_s=30.0):
    """Oracle (CPU port) train step like the timed GPU step: fwd (dropout 0.1 ON, masks drawn per site with torch.rand) +
    CE + autograd bwd + clip + RAdamScheduleFree, fp32.  Like-for-like with the survey probe of the reference (BASELINE.md
    section 2: batch 16): the thread count is chosen by a 3-point sweep (16 / 32 / all host cores, one step each) and the best
    one is timed for `steps` steps; `cores` = the threads used.  Falls back to batch 8 when a step would not fit the budget."""
    import numpy as np
    import torch
    from kzv import params as P
    from kzv.data import synthetic_batch
    from oracle import trocr_oracle as O
    all_threads = torch.get_num_threads()

    def make(batch):
        sd = O.leaf_state_dict(P.state_dict_from_flat(cfg, P.recipe_flat(cfg, 42)))
        px, lab = synthetic_batch(cfg, batch, label_len, seed=1)
        return {"sd": sd, "px": torch.from_numpy(px), "lab": torch.from_numpy(lab), "st": O.RAdamScheduleFreeState(),
                "z": {k: v.detach().clone() for k, v in sd.items()}, "vv": {k: torch.zeros_like(v) for k, v in sd.items()}, "B": batch}

    def draw_masks(B):      # what nn.Dropout / F.dropout do in the reference's training mode (19 % of its CPU step)
        T, Se = label_len - 1, cfg.enc_seq

        def mk(shape, p):
            return (torch.rand(shape) >= p).float() / (1.0 - p)
        m = {"enc_emb": mk((B, Se, cfg.enc_hidden), cfg.enc_hidden_dropout), "dec_emb": mk((B, T, cfg.dec_hidden), cfg.dec_hidden_dropout)}
        for i in range(cfg.enc_layers):
            m[f"enc{i}_attn"] = mk((B, cfg.enc_heads, Se, Se), cfg.enc_attn_dropout)
            m[f"enc{i}_o"] = mk((B, Se, cfg.enc_hidden), cfg.enc_hidden_dropout)
            m[f"enc{i}_mlp"] = mk((B, Se, cfg.enc_hidden), cfg.enc_hidden_dropout)
        for i in range(cfg.dec_layers):
            m[f"dec{i}_sa"] = mk((B, cfg.dec_heads, T, T), cfg.dec_attn_dropout)
            m[f"dec{i}_ca"] = mk((B, cfg.dec_heads, T, cfg.num_patches), cfg.dec_attn_dropout)
            for sfx in ("sa_o", "ca_o", "ffn"):
                m[f"dec{i}_{sfx}"] = mk((B, T, cfg.dec_hidden), cfg.dec_hidden_dropout)
        return m

    def one_step(w):
        t0 = time.perf_counter()
        sd, st = w["sd"], w["st"]
        for v in sd.values():
            v.grad = None
        _, loss = O.forward(cfg, sd, w["px"], w["lab"], masks=draw_masks(w["B"]))
        loss.backward()
        grads = [v.grad for v in sd.values()]
        total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
        coef = min(1.0, 1.0 / (total + 1e-6))
        lr, ckp1, bc2, adaptive = st.next_scalars()
        with torch.no_grad():
            for k, p_ in sd.items():
                g = p_.grad * coef
                w["vv"][k].mul_(st.beta2).addcmul_(g, g, value=1 - st.beta2)
                gn = g / ((w["vv"][k] / bc2).sqrt() + st.eps) if adaptive else g
                p_.lerp_(w["z"][k], ckp1)
                p_.add_(gn, alpha=lr * (st.beta1 * (1 - ckp1) - 1))
                w["z"][k].sub_(gn, alpha=lr)
        return time.perf_counter() - t0

    t_begin = time.perf_counter()

    def sweep_threads(w):
        out = {}
        for n in sorted({min(16, all_threads), min(32, all_threads), all_threads}):
            torch.set_num_threads(n)
            out[n] = one_step(w)
        return out
    w = make(sample_batch)
    torch.set_num_threads(min(16, all_threads))               # (at all 128 host threads the warm-up alone took 12 s and pushed the leg to batch 8)
    one_step(w)                                               # warm-up (allocations, first touch: several times a steady step)
    sweep = sweep_threads(w)
    # the stated fallback: steady steps of batch 16 that would not leave room for two timed ones inside the budget -> batch 8
    if sample_batch > 8 and (time.perf_counter() - t_begin) + 2 * min(sweep.values()) > budget_s:
        sample_batch = 8
        w = make(sample_batch)
        torch.set_num_threads(min(16, all_threads))
        one_step(w)
        sweep = sweep_threads(w)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = []
    for _ in range(max(2, steps)):
        times.append(one_step(w))
        if time.perf_counter() - t_begin > budget_s and len(times) >= 2:
            break
    torch.set_num_threads(all_threads)
    dt = float(np.mean(times))
    return {"value": sample_batch / dt, "unit": "img/s", "cores": best, "kind": "port",
            "sample": f"{len(times)} train steps of batch {sample_batch} (the survey probe's batch is 16) on the same geometry with ALL {label_len - 1} decoder positions computed (as the reference; compare with config.untrimmed), fp32, dropout 0.1 "
                      f"on, oracle/trocr_oracle.py with torch CPU autograd, on {best} threads = the best of a one-step sweep "
                      + ", ".join(f"{n}: {sample_batch / t:.2f} img/s" for n, t in sorted(sweep.items()))
                      + f" (host has {all_threads}); {dt:.2f} s/step; whole leg {time.perf_counter() - t_begin:.0f} s"}


def pick_stride(launches_per_step: int) -> int:
    """Sampling period of the per-launch events: the first of 4, 5, 3, 7, 9, 11 co-prime to the family's launches per step (every launch
    site is then bracketed equally often over that many steps); 1 = bracket everything if none is."""
    import math
    return next((c for c in (4, 5, 3, 7, 9, 11) if math.gcd(c, max(launches_per_step, 1)) == 1), 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling: global = batch * gpus)")
    ap.add_argument("--dec-layers", type=int, default=6, help="6 = BASELINE.json configs[1]; 12 = reference decoder")
    ap.add_argument("--encoder", choices=["vit_b", "vit_l"], default="vit_b",
                    help="vit_b = the benchmark (configs[1]/[2]); vit_l = configs[3]'s ViT-L/16 encoder (a side measurement)")
    ap.add_argument("--wide-decoder", action="store_true",
                    help="with --encoder vit_l: the 1024 / 16-head / FFN-4096 decoder SURVEY.md section 8(d) prices configs[3] with "
                         "(468 GFLOP per image at 12 layers) instead of the reference decoder's 256 / 4 / 768 widths")
    ap.add_argument("--fp8-mode", type=int, default=2, choices=[1, 2], help="with --fp8: 1 = forward GEMMs only, 2 = + the MLP's input-gradient GEMMs")
    ap.add_argument("--fp8", action="store_true",
                    help="side measurement (configs[4], second half): the encoder's QKV / fc1 / fc2 forward GEMMs on e4m3 operands; "
                         "NOT the benchmark line, which is bf16")
    ap.add_argument("--label-len", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=16, help="the survey probe's batch (BASELINE.md section 2)")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--dp-rehearsal", action="store_true",
                    help="one GPU only: after the normal line's timed region, time the SAME step through the data-parallel branch on a "
                         "one-rank RCCL process group (segmented backward, bucketed async all-reduce, CU reserve 32, 16 channels) and "
                         "report its img/s beside the plain path's in `dp_rehearsal`")
    args = ap.parse_args()
    if args.dp_rehearsal:
        if args.gpus != 1:
            raise SystemExit("--dp-rehearsal is a one-GPU rehearsal of the N > 1 code path")
        os.environ["KZV_FORCE_DIST"] = "1"
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")

    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner to stdout when its first communicator comes up, so
    # for the whole run file descriptor 1 points at stderr and the result line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from kzv import _lib as L
    from kzv.config import vit_b_config, vit_l_config, vit_l_wide_config
    from kzv.data import build_decoder_dir, synthetic_batch
    from kzv.model import TrOCRModel
    from kzv.trainer import Stepper, init_distributed

    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if args.wide_decoder and args.encoder != "vit_l":
        raise SystemExit("--wide-decoder goes with --encoder vit_l")
    cfg = (vit_b_config(dec_layers=args.dec_layers) if args.encoder == "vit_b" else
           vit_l_wide_config(dec_layers=args.dec_layers) if args.wide_decoder else vit_l_config(dec_layers=args.dec_layers))
    lib = L.load()

    with tempfile.TemporaryDirectory() as tmp:
        d = build_decoder_dir(os.path.join(tmp, "dec"), cfg)
        model = TrOCRModel(cfg.encoder_config_dict(), d, device=dev, init_seed=42, load_tokenizer=False, fp8=args.fp8_mode if args.fp8 else 0)
    model._step_seed = 1_000_003 * rank            # per-rank dropout streams
    opt = model.configure_optimizers()
    model.train()
    stepper = Stepper(model, opt, world=world, max_grad_norm=1.0, dp_path=world > 1)
    if args.dp_rehearsal:         # the plain path is timed first, as without the flag: no CUs held back for a collective
        L.check(lib.kzv_set_cu_reserve(0), "set_cu_reserve")
    px, lab = synthetic_batch(cfg, args.batch, args.label_len, seed=1 + rank)
    batch = {"pixel_values": torch.from_numpy(px).to(dev), "labels": torch.from_numpy(lab).to(dev)}   # resident in HBM

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        stepper.step(batch, i)
    barrier()
    # Timed region: HIP events bracket every launch of the reported kernel family only (each bracket costs ~2 us of
    # stream time; gemm_tn / attention are measured in one extra, untimed step below).
    events = not os.environ.get("KZV_BENCH_NO_EVENTS")     # dev knob: step time without any per-launch events
    stride = int(os.environ.get("KZV_BENCH_EVENT_STRIDE", "0"))
    if events and stride <= 0:
        # every s-th launch is bracketed; s must be coprime with the family's launches per step, or the sample only ever sees the
        # same residue class of positions in the step (142 launches at s = 4: the even ones only -- it read 7 % low).  One untimed
        # step counts the launches (a sampling period no launch reaches: nothing is bracketed).
        L.check(lib.kzv_prof_select(1 << 0), "prof_select")
        L.check(lib.kzv_prof_sample(1 << 30), "prof_sample")
        L.check(lib.kzv_prof_enable(1, 16), "prof_enable")
        stepper.step(batch, args.warmup)
        barrier()
        L.check(lib.kzv_prof_enable(0, 0), "prof_disable")
        stride = pick_stride(int(lib.kzv_prof_seen(0)))
    if events:
        L.check(lib.kzv_prof_select(1 << 0), "prof_select")
        L.check(lib.kzv_prof_sample(stride), "prof_sample")
        L.check(lib.kzv_prof_enable(1, 16384), "prof_enable")
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = stepper.step(batch, i)
    barrier()
    dt = time.perf_counter() - t0
    L.check(lib.kzv_prof_enable(0, 0), "prof_disable")
    nt_ms, nt_fl, nt_n = C.c_double(), C.c_double(), C.c_int64()
    L.check(lib.kzv_prof_collect(0, C.byref(nt_ms), C.byref(nt_fl), C.byref(nt_n)), "prof_collect")
    nt_seen = int(lib.kzv_prof_seen(0))
    if events:                                             # untimed: one more step for the other kernel families
        L.check(lib.kzv_prof_sample(1), "prof_sample")
        L.check(lib.kzv_prof_select((1 << 1) | (1 << 2) | (1 << 3) | (1 << 4)), "prof_select")
        L.check(lib.kzv_prof_enable(1, 16384), "prof_enable")
        stepper.step(batch, args.steps)
        barrier()
        L.check(lib.kzv_prof_enable(0, 0), "prof_disable")
        L.check(lib.kzv_prof_select(0xffffffff), "prof_select")
    # The reference computes every decoder position (trocr_model.py:274-292); the headline skips the columns that are padding in
    # every sample of the batch (exact).  The same step with the trim off, a few steps outside the headline's timed region:
    untrimmed = None
    if world == 1 and getattr(model, "trim_padding", False) and not os.environ.get("KZV_BENCH_NO_UNTRIMMED"):
        t_trim = getattr(model, "last_active_length", args.label_len - 1)
        model.trim_padding = False
        for i in range(2):
            stepper.step(batch, i)
        barrier()
        n_un = max(3, min(args.steps, 5))
        t1 = time.perf_counter()
        for i in range(n_un):
            stepper.step(batch, i)
        barrier()
        dt_un = time.perf_counter() - t1
        untrimmed = {"value": args.batch * n_un / dt_un, "unit": "img/s", "ms_per_step": dt_un / n_un * 1e3, "steps": n_un,
                     "decoder_positions": int(getattr(model, "last_active_length", args.label_len - 1)),
                     "note": "the same step with every decoder position computed, as the reference does; measured after the headline's timed region"}
        model.trim_padding = True
        model.last_active_length = t_trim
    dp_reh = None
    if args.dp_rehearsal:
        L.check(lib.kzv_set_cu_reserve(32), "set_cu_reserve")        # what init_distributed sets for world > 1 on RCCL
        dp = Stepper(model, opt, world=1, max_grad_norm=1.0, dp_path=True)
        for i in range(args.warmup):
            dp.step(batch, i)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            dp.step(batch, i)
        barrier()
        dt_dp = time.perf_counter() - t1
        dp_reh = {"value": args.batch * args.steps / dt_dp, "unit": "img/s", "ms_per_step": dt_dp / args.steps * 1e3,
                  "plain_value": args.batch * args.steps / dt, "delta_pct": (dt / dt_dp - 1.0) * 100.0,
                  "backend": torch.distributed.get_backend(), "buckets": len(dp.buckets), "cu_reserve": 32,
                  "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                  "note": "the N > 1 branch of Stepper.step on a ONE-rank RCCL group (a one-rank all-reduce moves no bytes): per-GPU "
                          "overhead of the segmented backward, the collective launches and the CU reserve, not a scaling number"}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    def collect(kind):
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        L.check(lib.kzv_prof_collect(kind, C.byref(ms), C.byref(fl), C.byref(n)), "prof_collect")
        return ms.value, fl.value, n.value

    if rank == 0:
        imgs = args.batch * world * args.steps
        T = args.label_len - 1
        ms, fl, n = nt_ms.value, nt_fl.value, nt_n.value
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        t_act = getattr(model, "last_active_length", T)     # decoder positions actually computed (trailing all-pad columns trimmed)
        step_flops = train_flops_per_image(cfg, t_act) * args.batch
        others = {}
        for kind, name in ((1, "gemm_tn_kernel"), (2, "attn_fwd_kernel"), (3, "attn_bwd_kernel")):
            m2, f2, n2 = collect(kind)
            others[name] = {"ms_per_step": m2, "TFLOP/s": (f2 / (m2 * 1e-3) / 1e12) if m2 > 0 else 0.0,
                            "launches_per_step": n2, "measured": "one extra step outside the timed region"}
        if args.fp8:
            m2, f2, n2 = collect(4)
            others["gemm_nt_fp8_kernel"] = {"ms_per_step": m2, "TFLOP/s": (f2 / (m2 * 1e-3) / 1e12) if m2 > 0 else 0.0, "launches_per_step": n2,
                                            "peak": PEAK_FP8_TFLOPS, "frac": (f2 / (m2 * 1e-3) / 1e12 / PEAK_FP8_TFLOPS) if m2 > 0 else 0.0,
                                            "measured": "one extra step outside the timed region"}
        DECPOS = (f"{t_act} of {T} computed: columns that are padding in every sample of the batch are skipped (exact: masked "
                  "keys, ignored targets); labels hold U{8..60} characters (BASELINE.md section 4)")
        # HBM-side bytes per launch come from rocprofv3 PMC passes (they cannot be collected from inside this process):
        # quoted only from a profile of THIS workload taken on THESE kernel sources (tools/hbm_traffic.py records their hash),
        # null otherwise -- never from a stale file
        traffic, traffic_src = None, None
        import glob
        for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
            j = json.load(open(tj))
            if j.get("source_sha16") == source_sha16() and args.batch == 256 and args.dec_layers == 6 and args.encoder == "vit_b" and not args.fp8 and not args.wide_decoder:
                k = j["kernels"].get("gemm_nt_kernel<*>")
                if k:
                    traffic, traffic_src = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"], os.path.relpath(tj, ROOT)
                    break
        out = {
            "metric": "line-images/sec (train)", "value": imgs / dt, "unit": "img/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not args.fp8 else "bf16 + e4m3 (encoder QKV/fc1/fc2 forward GEMMs): side measurement, not the benchmark line",
            "data": "synthetic",
            "config": {"workload": f"TrOCR train step: {'ViT-B/16 (12L/768/12h/3072)' if args.encoder == 'vit_b' else 'ViT-L/16 (24L/1024/16h/4096)'} on 64x640 crops (S_e=161) + RoBERTa "
                                   f"decoder {args.dec_layers}L/{cfg.dec_hidden}/{cfg.dec_heads}h/{cfg.dec_ffn}, V=4300 one-char vocab, labels [B,{args.label_len}], "
                                   f"dropout 0.1, clip 1.0, RAdamScheduleFree; BASELINE.json configs[{(1 if world == 1 else 2) if args.encoder == 'vit_b' else 3}]",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "decoder_positions": DECPOS,
                       "untrimmed": untrimmed,
                       "final_loss": final_loss},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_kernel<*> (bf16 MFMA 16x16x32, all nn.Linear fwd + dgrad)",
                         "achieved": ach, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS,
                         "traffic": traffic, "traffic_unit": f"bytes/launch (L2-miss fetch x2-corrected + write, rocprofv3 PMC, {traffic_src}; null = no profile of these kernel sources)",
                         "launches_per_step": nt_seen / args.steps, "avg_launch_us": (ms / n * 1e3) if n else None,
                         "kernel_ms_per_step": (ms / n * nt_seen / args.steps) if n else None,
                         "events": f"every {stride}th launch bracketed inside the timed region ({n} of {nt_seen} launches)",
                         "source_sha16": source_sha16(),
                         "whole_step_TFLOP/s": step_flops / (dt / args.steps) / 1e12,
                         "whole_step_frac": step_flops / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS,
                         "other_kernels": others},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.label_len, args.cpu_sample_batch, args.cpu_steps)
        else:
            out["cpu_baseline"] = None
        if dp_reh is not None:
            out["dp_rehearsal"] = dp_reh
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
