#!/usr/bin/env python3
"""CLI entrypoint of the MI355X engine: the flag surface of scripts/train_trocr.py:23-71 (TrOCR) plus the
ocr_lightning/train.py:161-189 spellings, wired to kzv.TrOCRModel + kzv.trainer.fit; `--model ocr` trains ocr_lightning/model.py's
ResNet34 / BiLSTM / CTC OCRModel (kzv.ocr_model) on ocr_lightning/train.py's folder datasets.

  python -m kzv.train --synthetic 64 --batch_size 32 --encoder_hidden_size 384 --encoder_num_layers 6 \
         --encoder_num_heads 6 --image_size 64 640 --max_epochs 1            # BASELINE.json configs[0] plumbing
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m kzv.train --gpus 8 ...
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train TrOCR model (MI355X engine)")
    # data (scripts/train_trocr.py:27-36)
    p.add_argument("--csv_path", type=str, default="data/processed_v2/column_info.csv")
    p.add_argument("--image_root", type=str, default="data/processed_v2/column_images")
    p.add_argument("--decoder_path", type=str,
                   default="experiments/pretrain_language_model/roberta-small-japanese-aozora-char/20250530_034458/checkpoint-200000")
    p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic line crops (no CSV / checkpoint needed)")
    p.add_argument("--train_data_dir", type=str, default=None, help="ocr_lightning/train.py spelling (unused by TrOCR)")
    p.add_argument("--val_data_dir", type=str, default=None, help="ocr_lightning/train.py spelling (unused by TrOCR)")
    # model (:39-44)
    p.add_argument("--image_size", type=int, nargs=2, default=[1024, 64])
    p.add_argument("--patch_size", type=int, nargs=2, default=[16, 16])
    p.add_argument("--encoder_hidden_size", type=int, default=768)
    p.add_argument("--encoder_num_layers", type=int, default=12)
    p.add_argument("--encoder_num_heads", type=int, default=8,
                   help="reference default 8 = head_dim 96 (scripts/train_trocr.py:43), which runs on the plain fp32 attention kernel; "
                        "hidden_size / 64 heads (12 for ViT-B/16) take the MFMA attention kernels, several times faster")
    p.add_argument("--max_length", type=int, default=128)
    # training (:47-54)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--weight_decay", type=float, default=0)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--beta2", type=float, default=0.999)
    p.add_argument("--epsilon", type=float, default=1e-8)
    p.add_argument("--max_epochs", type=int, default=50)
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--num_workers", type=int, default=8)
    p.add_argument("--train_ratio", type=float, default=0.8)
    p.add_argument("--val_ratio", type=float, default=0.1)
    p.add_argument("--test_ratio", type=float, default=0.1)
    # output (:62-63)
    p.add_argument("--output_dir", type=str, default="experiments/trocr")
    p.add_argument("--experiment_name", type=str, default="trocr_vit_roberta")
    # hardware (:66-69 and ocr_lightning/train.py:182-186)
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--devices", type=str, default=None, help="ocr_lightning spelling of --gpus")
    p.add_argument("--accelerator", type=str, default="gpu")
    p.add_argument("--precision", type=str, default="bf16-mixed", choices=["16-mixed", "32", "bf16-mixed", "fp8-mixed"],
                   help="bf16-mixed = the reference's setting and this engine's arithmetic; fp8-mixed (an extension, BASELINE configs[4]) "
                        "additionally runs the encoder's QKV / fc1 / fc2 forward GEMMs on e4m3 operands (hidden and ffn sizes must be multiples of 256)")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--device_preprocess", action="store_true",
                   help="resize / pad / normalise the decoded crops on the GPU (kzv.preprocess; byte-exact with the PIL transform)")
    p.add_argument("--skip_test", action="store_true", help="do not run the reference's post-fit test phase (scripts/train_trocr.py:193-195)")
    p.add_argument("--ema_decay", type=float, default=0.0, help="> 0 attaches kzv.ema.EMACallback (reference: decay 0.9999)")
    # the ResNet34 / BiLSTM / CTC model of ocr_lightning/model.py behind ocr_lightning/train.py's own flags (:161-189)
    p.add_argument("--model", type=str, default="trocr", choices=["trocr", "ocr"],
                   help="trocr = scripts/train_trocr.py's model (the benchmark path); ocr = ocr_lightning/train.py's OCRModel "
                        "(--train_data_dir / --val_data_dir folder datasets, Adam, early stopping on val/total_loss)")
    p.add_argument("--checkpoint_dir", type=str, default="./ocr_checkpoints")
    p.add_argument("--log_dir", type=str, default="./ocr_logs")
    p.add_argument("--max_boxes", type=int, default=50)
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--patience", type=int, default=3)
    p.add_argument("--resume", type=str, default=None,
                   help="--model ocr: continue from a checkpoint this CLI (or Lightning, for the reference) wrote -- weights, Adam moments and "
                        "step count, epoch (pl.Trainer.fit(ckpt_path=...))")
    return p.parse_args(argv)


def main_ocr(args):
    """ocr_lightning/train.py:35-158: seed, folder datasets + pad-collate loaders, OCRModel, ModelCheckpoint(monitor val/total_loss,
    save_top_k=1, save_last) + EarlyStopping(patience) over `--epochs` epochs of Adam steps; per-epoch metrics as JSON lines in --log_dir
    (the reference writes TensorBoard events; there is no tensorboard in this image)."""
    import json
    import torch
    from .ocr_data import CHAR_TO_IDX, IDX_TO_CHAR, OcrDataset, OcrLoader
    from .ocr_model import OCRModel
    if not args.train_data_dir or not args.val_data_dir:
        raise SystemExit("--model ocr needs --train_data_dir and --val_data_dir (ocr_lightning/train.py:163-164)")
    if args.accelerator not in ("gpu", "auto"):
        raise SystemExit("this engine runs on MI355X GPUs only")
    from .trainer import init_distributed
    rank, world, local = init_distributed()              # one process per GPU (torch.distributed.run), as for the TrOCR path
    torch.cuda.set_device(local)
    torch.manual_seed(args.seed)
    os.makedirs(args.checkpoint_dir, exist_ok=True); os.makedirs(args.log_dir, exist_ok=True)
    train_ds = OcrDataset(args.train_data_dir, char_to_idx=CHAR_TO_IDX)
    val_ds = OcrDataset(args.val_data_dir, char_to_idx=CHAR_TO_IDX)
    if len(train_ds) == 0:
        print(f"Error: Training dataset at {args.train_data_dir} is empty. Please check the path and data structure.")
        return None
    train_loader = OcrLoader(train_ds, args.batch_size, shuffle=True, seed=args.seed, rank=rank, world=world)
    val_loader = OcrLoader(val_ds, args.batch_size) if len(val_ds) else None
    model = OCRModel(CHAR_TO_IDX, IDX_TO_CHAR, learning_rate=args.learning_rate, max_boxes=args.max_boxes, init_seed=args.seed, device=f"cuda:{local}")
    model.configure_optimizers()
    start_epoch = 0
    if getattr(args, "resume", None):                       # what pl.Trainer.fit(ckpt_path=...) restores: weights, Adam state, epoch
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["state_dict"], strict=True)
        if ck.get("optimizer_states"):
            model.load_optimizer_state_dict(ck["optimizer_states"][0])
        start_epoch = int(ck.get("epoch", -1)) + 1
        if rank == 0:
            print(f"Resumed from {args.resume}: epoch {start_epoch}, optimizer step {model._optimizer.step_count}")
    state, hist = {"best": float("inf"), "best_path": None, "bad": 0}, []
    log = open(os.path.join(args.log_dir, "metrics.jsonl") if rank == 0 else os.devnull, "a", encoding="utf-8")
    for epoch in range(start_epoch, args.epochs):
        train_loader.set_epoch(epoch)
        model.logged.clear()
        for i, batch in enumerate(train_loader):
            model.fit_step(batch, i)
        rec = {"epoch": epoch, **{k: sum(v) / len(v) for k, v in model.logged.items() if k.startswith("train/")}}
        if val_loader is not None:
            model.eval(); model.logged.clear()
            for i, batch in enumerate(val_loader):
                model.validation_step(batch, i)
            rec.update({k: sum(v) / len(v) for k, v in model.logged.items() if k.startswith("val/")})
            model.train()
        hist.append(rec)
        stop = [False]
        if rank == 0:
            stop[0] = _ocr_epoch_end(args, model, rec, epoch, log, state)
        if world > 1:                                        # every rank leaves the loop together (rank 0 monitors val/total_loss)
            torch.distributed.broadcast_object_list(stop, src=0)
        if stop[0]:
            break
    log.close()
    best_path = state["best_path"]
    if rank == 0:
        print(f"Training finished.\nBest model checkpoint saved at: {best_path}" if best_path else "No best model checkpoint was saved.")
    main_ocr.best_model_path = best_path
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return hist


def _ocr_epoch_end(args, model, rec, epoch, log, state):
    """rank 0: metrics line, last.ckpt, ModelCheckpoint(monitor val/total_loss, save_top_k=1), EarlyStopping(patience) -> stop?"""
    import json
    import torch
    best, best_path, bad = state["best"], state["best_path"], state["bad"]
    log.write(json.dumps(rec) + "\n"); log.flush()
    print(" ".join(f"{k}={v:.4f}" if isinstance(v, float) else f"{k}={v}" for k, v in rec.items()))
    def ckpt():      # Lightning's layout for the keys the reference's checkpoints carry (ModelCheckpoint(save_last=True), train.py:118-126)
        return {"state_dict": {k: v.cpu() for k, v in model.state_dict().items()}, "hyper_parameters": vars(model.hparams), "epoch": epoch,
                "global_step": model._optimizer.step_count if model._optimizer else 0, "optimizer_states": [model.optimizer_state_dict()], "lr_schedulers": []}
    torch.save(ckpt(), os.path.join(args.checkpoint_dir, "last.ckpt"))
    vl = rec.get("val/total_loss")
    if vl is not None:
        if vl < best:
            best, bad = vl, 0
            if best_path and os.path.exists(best_path):
                os.remove(best_path)
            best_path = os.path.join(args.checkpoint_dir, f"ocr-epoch={epoch:02d}-val_total_loss={vl:.2f}.ckpt")
            torch.save(ckpt(), best_path)
        else:
            bad += 1
            if bad >= args.patience:
                print(f"Early stopping: val/total_loss has not improved for {bad} epochs")
                state.update(best=best, best_path=best_path, bad=bad)
                return True
    state.update(best=best, best_path=best_path, bad=bad)
    return False


def main(argv=None):
    args = parse_args(argv)
    if args.model == "ocr":
        return main_ocr(args)
    import torch
    from .config import ModelConfig
    from .data import LineCsvDataset, SyntheticLineDataset, build_decoder_dir, make_loader
    from .model import TrOCRModel
    from .trainer import fit, init_distributed, test as run_test

    if args.devices is not None:
        args.gpus = int(args.devices) if str(args.devices).isdigit() else len(str(args.devices).split(","))
    if args.accelerator != "gpu" or args.gpus < 1:
        raise SystemExit("this engine runs on MI355X GPUs only (the reference CPU Trainer branch is not provided)")
    if args.precision not in ("bf16-mixed", "fp8-mixed"):
        raise SystemExit("the engine implements bf16-mixed (scripts/train_trocr.py:68 default) and its fp8-mixed extension only")
    hd = args.encoder_hidden_size / max(1, args.encoder_num_heads)
    if hd != int(hd) or int(hd) % 8 or hd > 128:
        raise SystemExit(f"encoder head_dim {hd:g} is not supported: --encoder_hidden_size / --encoder_num_heads must be a multiple of 8 up to 128")
    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU with torch.distributed.run")
    torch.manual_seed(args.seed)
    out_dir = os.path.join(args.output_dir, args.experiment_name)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        print(f"Output directory: {out_dir}\nDecoder path: {args.decoder_path}")
    tmp = None
    decoder_path = args.decoder_path
    if args.synthetic and not os.path.exists(decoder_path):
        tmp = tempfile.TemporaryDirectory()
        decoder_path = build_decoder_dir(os.path.join(tmp.name, "decoder"), ModelConfig(max_pos=max(128, args.max_length)))
    if not os.path.exists(decoder_path):
        raise FileNotFoundError(f"Decoder path not found: {decoder_path}")      # scripts/train_trocr.py:88-89
    encoder_config = {"image_size": tuple(args.image_size), "patch_size": tuple(args.patch_size), "num_channels": 3,
                      "hidden_size": args.encoder_hidden_size, "num_hidden_layers": args.encoder_num_layers,
                      "num_attention_heads": args.encoder_num_heads, "intermediate_size": args.encoder_hidden_size * 4,
                      "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1}   # :111-121
    model = TrOCRModel(encoder_config, decoder_path, learning_rate=args.learning_rate, beta1=args.beta1, beta2=args.beta2,
                       epsilon=args.epsilon, weight_decay=args.weight_decay, device=f"cuda:{local}", init_seed=args.seed,
                       fp8=args.precision == "fp8-mixed")
    model._step_seed = 1_000_003 * rank
    if args.synthetic:
        n_val = max(args.batch_size, args.synthetic // 10)
        train_ds = SyntheticLineDataset(model.cfg, args.synthetic, args.max_length, seed=args.seed)
        val_ds = SyntheticLineDataset(model.cfg, n_val, args.max_length, seed=args.seed + 1)
        test_ds = SyntheticLineDataset(model.cfg, n_val, args.max_length, seed=args.seed + 2)
    else:
        kw = dict(csv_path=args.csv_path, image_root=args.image_root, tokenizer=model.tokenizer, image_size=tuple(args.image_size),
                  max_length=args.max_length, train_ratio=args.train_ratio, val_ratio=args.val_ratio, test_ratio=args.test_ratio)
        kw["device_preprocess"] = args.device_preprocess
        train_ds, val_ds, test_ds = LineCsvDataset(split="train", **kw), LineCsvDataset(split="val", **kw), LineCsvDataset(split="test", **kw)
    pre = None
    if args.device_preprocess and not args.synthetic:
        from .preprocess import DevicePreprocessor
        pre = DevicePreprocessor(tuple(args.image_size), device=f"cuda:{local}")
    nw = 0 if args.synthetic else max(0, args.num_workers)      # decode (and, without --device_preprocess, resize) in worker processes
    train_loader = make_loader(train_ds, args.batch_size, True, args.seed, rank, world, num_workers=nw, preprocessor=pre)
    val_loader = make_loader(val_ds, args.batch_size, False, args.seed, rank, world, num_workers=nw, preprocessor=pre)
    test_loader = make_loader(test_ds, args.batch_size, False, args.seed, rank, world, num_workers=nw, preprocessor=pre)   # scripts/train_trocr.py:93
    if rank == 0:
        print(f"Train samples: {len(train_ds)}\nVal samples: {len(val_ds)}\nTest samples: {len(test_ds)}\nParameters: {model.num_parameters():,}")
    cbs = []
    if args.ema_decay > 0:
        from .ema import EMACallback
        cbs.append(EMACallback(args.ema_decay))
    hist = fit(model, train_loader, val_loader, max_epochs=args.max_epochs, max_steps=args.max_steps, log_every=50,
               val_check_interval=0.5, ckpt_dir=os.path.join(out_dir, "checkpoints"), world=world, rank=rank,
               callbacks=cbs)
    if rank == 0:
        print(f"Training completed! Checkpoints saved to: {os.path.join(out_dir, 'checkpoints')}")
    # scripts/train_trocr.py:193-195: if test_loader is not None: trainer.test(model, test_loader, ckpt_path="best")
    main.test_metrics = None
    if not args.skip_test and len(test_ds) > 0:
        if world > 1:
            torch.distributed.barrier()            # rank 0 has written the checkpoint the others load
        path = fit.best_model_path if rank == 0 else None
        if world > 1:
            box = [path]
            torch.distributed.broadcast_object_list(box, src=0)
            path = box[0]
        main.test_metrics = run_test(model, test_loader, ckpt_path=path, rank=rank)
    if world > 1:
        torch.distributed.destroy_process_group()
    if tmp is not None:
        tmp.cleanup()
    return hist


if __name__ == "__main__":
    sys.exit(0 if main() is not None else 1)
