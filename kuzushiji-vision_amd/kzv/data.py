"""Host-side input plumbing: one-char tokenizer directory + synthetic line-crop batches.

Reference behaviour restated (no reference code is imported here):
  * tokenizer: WordLevel + NFKC + split-every-char, specials [UNK] [PAD] [CLS] [SEP] [MASK] = ids 0..4,
    bos=[CLS], eos=[SEP]; adds NO BOS/EOS on encode  -- train_tokenizer_one_char.py:66-95,129-141
  * dataset item: pixel_values fp32 [3,H,W] in [-1,1]; labels = tokenizer(text, max_length,
    padding="max_length", truncation=True) int64 [max_length]  -- src/data/trocr_dataset.py:97-104,173-201
  * synthetic batch recipe: BASELINE.md section 4 / SURVEY.md section 8(d)
"""
from __future__ import annotations

import dataclasses
import json
import os

import numpy as np

from .config import ModelConfig

SPECIAL_TOKENS = ["[UNK]", "[PAD]", "[CLS]", "[SEP]", "[MASK]"]


def synthetic_charset(n: int) -> list[str]:
    """n distinct NFKC-stable CJK ideographs starting at U+4E00 (ids 5.. after the 5 specials)."""
    import unicodedata
    out, cp = [], 0x4E00
    while len(out) < n:
        ch = chr(cp)
        if unicodedata.normalize("NFKC", ch) == ch:
            out.append(ch)
        cp += 1
    return out


def build_decoder_dir(path: str, cfg: ModelConfig, with_weights: bool = False) -> str:
    """Create the local ``decoder_path`` directory the reference expects
    (config.json + tokenizer files; scripts/train_trocr.py:31-36,88-89) for a synthetic charset
    of ``cfg.vocab - 5`` characters.  Character ids are assigned in code-point order so the
    mapping is reproducible without a corpus."""
    from tokenizers import Regex, Tokenizer, decoders, normalizers, pre_tokenizers
    from tokenizers.models import WordLevel

    os.makedirs(path, exist_ok=True)
    chars = synthetic_charset(cfg.vocab - len(SPECIAL_TOKENS))
    vocab = {t: i for i, t in enumerate(SPECIAL_TOKENS)}
    for c in chars:
        vocab[c] = len(vocab)
    tok = Tokenizer(WordLevel(vocab=vocab, unk_token="[UNK]"))
    tok.normalizer = normalizers.Sequence([normalizers.NFKC()])
    tok.pre_tokenizer = pre_tokenizers.Split(Regex(r"[\s\S]"), behavior="isolated")
    tok.decoder = decoders.Sequence([])
    tok.add_special_tokens(SPECIAL_TOKENS)
    tok.save(os.path.join(path, "tokenizer.json"))
    with open(os.path.join(path, "tokenizer_config.json"), "w", encoding="utf-8") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "unk_token": "[UNK]", "pad_token": "[PAD]",
                   "cls_token": "[CLS]", "sep_token": "[SEP]", "mask_token": "[MASK]",
                   "bos_token": "[CLS]", "eos_token": "[SEP]", "do_lower_case": False}, f)
    with open(os.path.join(path, "special_tokens_map.json"), "w", encoding="utf-8") as f:
        json.dump({"unk_token": "[UNK]", "pad_token": "[PAD]", "cls_token": "[CLS]", "sep_token": "[SEP]",
                   "mask_token": "[MASK]", "bos_token": "[CLS]", "eos_token": "[SEP]"}, f)
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg.decoder_config_dict(), f, indent=1)
    return path


def synthetic_batch(cfg: ModelConfig, batch: int, label_len: int = 128, seed: int = 1,
                    min_chars: int = 8, max_chars: int = 60):
    """(pixel_values fp32 [B,3,H,W] ~ N(0,1), labels int64 [B,label_len]).

    randn crops are what the reference smoke tests feed (scripts/test_trocr_setup.py:124);
    labels: n ~ U{min..max} ids drawn from 5..V-1, PAD elsewhere, no BOS/EOS.
    """
    rng = np.random.default_rng(seed)
    px = rng.standard_normal((batch, cfg.channels, cfg.image_h, cfg.image_w), dtype=np.float32)
    labels = np.full((batch, label_len), cfg.pad_id, dtype=np.int64)
    hi = min(max_chars, label_len)
    lo = min(min_chars, hi)
    for b in range(batch):
        n = int(rng.integers(lo, hi + 1))
        labels[b, :n] = rng.integers(5, cfg.vocab, size=n)
    return px, labels


class SyntheticLineDataset:
    """Map-style dataset with the reference's item dict (src/data/trocr_dataset.py:196-201)."""

    def __init__(self, cfg: ModelConfig, n: int, label_len: int = 128, seed: int = 1, width_buckets=None):
        self.cfg, self.n, self.label_len, self.seed = cfg, n, label_len, seed
        self._chars = None
        # width buckets (configs[4]): sample i is a crop of width widths[i] (drawn per sample), same height
        self.widths = None
        if width_buckets:
            rng = np.random.default_rng(seed * 7919 + 11)
            self.widths = [int(w) for w in rng.choice(sorted(width_buckets), size=n)]

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int):
        import torch
        cfg = self.cfg if self.widths is None else dataclasses.replace(self.cfg, image_w=self.widths[i])
        px, lab = synthetic_batch(cfg, 1, self.label_len, seed=self.seed * 1_000_003 + i)
        if self._chars is None:
            self._chars = synthetic_charset(self.cfg.vocab - len(SPECIAL_TOKENS))
        text = "".join(self._chars[t - 5] for t in lab[0] if t != self.cfg.pad_id)
        return {"pixel_values": torch.from_numpy(px[0]), "labels": torch.from_numpy(lab[0]),
                "text": text, "image_path": f"synthetic://{i}"}


# ---- real-data plumbing (src/data/trocr_dataset.py) ---------------------------------------------------------
def resize_with_padding(image, target_size, fill_color=(255, 255, 255)):
    """ResizeWithPadding.__call__ (src/data/trocr_dataset.py:24-53): scale = min(w-ratio, h-ratio), int()
    truncation, LANCZOS, paste centred on a white canvas.  image: PIL.Image; target_size = (H, W)."""
    from PIL import Image
    th, tw = target_size
    ow, oh = image.size
    scale = min(tw / ow, th / oh)
    nw, nh = int(ow * scale), int(oh * scale)
    image = image.resize((nw, nh), Image.Resampling.LANCZOS)
    canvas = Image.new("RGB", (tw, th), fill_color)
    canvas.paste(image, ((tw - nw) // 2, (th - nh) // 2))
    return canvas


def image_to_tensor(image):
    """ToTensor + Normalize(0.5, 0.5) (trocr_dataset.py:101-102): uint8 HWC -> fp32 CHW in [-1, 1]."""
    import torch
    a = np.asarray(image, dtype=np.float32) / 255.0
    return torch.from_numpy(((a - 0.5) / 0.5).transpose(2, 0, 1).copy())


def reference_image_path(column_image: str) -> str:
    """``create_full_path`` of TrOCRDataset._load_data (src/data/trocr_dataset.py:121-130): absolute paths as they are,
    ``processed_v2/...`` rewritten to ``data/processed_v2/...``, anything else relative to the working directory.
    (The reference stores ``image_root`` but never uses it.)"""
    if column_image.startswith("/"):
        return column_image
    if column_image.startswith("processed_v2/"):
        return column_image.replace("processed_v2/", "data/processed_v2/", 1)
    return os.path.join(os.getcwd(), column_image)


class LineCsvDataset:
    """TrOCRDataset (src/data/trocr_dataset.py:56-201): column_info.csv rows -> {"pixel_values","labels","text",
    "image_path"}.  Rows are kept in FILE order (no shuffle); rows with missing fields, empty text or -- like the reference
    (:132-136) -- a non-existent image file are dropped BEFORE the train/val/test slices are cut, so the split boundaries
    are the reference's.  Images that exist but cannot be decoded become all-zero tensors (:182-185).

    ``resolve="reference"`` (default) locates images by the reference's rules (reference_image_path);
    ``resolve="image_root"`` joins ``image_root`` with the file's basename instead (an extension for relocated datasets;
    the existence filter applies to the resolved path either way)."""

    def __init__(self, csv_path, image_root, tokenizer, image_size=(1024, 64), max_length=128, split="train",
                 train_ratio=0.8, val_ratio=0.1, test_ratio=0.1, device_preprocess=False, resolve="reference", width_buckets=None):
        import ast
        import pandas as pd
        if abs(train_ratio + val_ratio + test_ratio - 1.0) >= 1e-6:
            raise AssertionError("Ratios must sum to 1.0")
        self.image_root, self.image_size, self.max_length, self.tokenizer = image_root, tuple(image_size), max_length, tokenizer
        self.device_preprocess = bool(device_preprocess)
        df = pd.read_csv(csv_path).dropna()
        df["unicode_ids"] = df["unicode_ids"].apply(ast.literal_eval)
        df["text"] = df["unicode_ids"].apply(self._ids_to_text)
        df = df[df["text"].str.len() > 0]
        if resolve == "reference":
            df["full_image_path"] = df["column_image"].apply(reference_image_path)
        elif resolve == "image_root":
            df["full_image_path"] = df["column_image"].apply(lambda r: os.path.join(image_root, os.path.basename(r)))
        else:
            raise ValueError(f"Invalid resolve: {resolve}")
        df = df[df["full_image_path"].apply(os.path.exists)]      # trocr_dataset.py:135
        df = df.reset_index(drop=True)          # the reference slices in file order (trocr_dataset.py:153-171)
        n = len(df)
        a, b = int(n * train_ratio), int(n * (train_ratio + val_ratio))
        if split == "train":
            self.data = df[:a].reset_index(drop=True)
        elif split == "val":
            self.data = df[a:b].reset_index(drop=True)
        elif split == "test":
            self.data = df[b:].reset_index(drop=True)
        else:
            raise ValueError(f"Invalid split: {split}")
        # width buckets (configs[4]): every crop is resized/padded to (H, its bucket) instead of (H, W); the bucket comes from
        # the image header (no decode); unreadable files go to the widest bucket (they become all-zero tensors anyway)
        self.widths = None
        if width_buckets:
            from PIL import Image
            self.widths = []
            for pth in self.data["full_image_path"]:
                try:
                    with Image.open(pth) as im:
                        w, h = im.size
                    self.widths.append(bucket_width(w, h, self.image_size[0], width_buckets))
                except Exception:
                    self.widths.append(max(width_buckets))

    def _target(self, idx):
        return self.image_size if self.widths is None else (self.image_size[0], self.widths[idx])

    @staticmethod
    def _ids_to_text(ids):
        out = ""
        for u in ids:
            try:
                if isinstance(u, str) and u.startswith("U+"):
                    out += chr(int(u[2:], 16))
            except (ValueError, OverflowError):
                continue
        return out

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        import torch
        from PIL import Image
        row = self.data.iloc[idx]
        path = row["full_image_path"]
        enc = self.tokenizer(row["text"], max_length=self.max_length, padding="max_length", truncation=True,
                             return_tensors="pt")
        item = {"labels": enc["input_ids"].squeeze(0), "text": row["text"], "image_path": path}
        if self.device_preprocess:
            item["target_w"] = self._target(idx)[1]
            # N2: hand the decoded uint8 crop (and its resampling plan) to kzv.preprocess.DevicePreprocessor; an unreadable
            # image becomes None -> an all-zero tensor after the device pass, like the reference's fallback
            from .preprocess import plan_line
            try:
                u8 = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
                th, tw = self._target(idx)
                item["image_u8"], item["plan"] = u8, plan_line(u8.shape[0], u8.shape[1], th, tw)
            except Exception:
                item["image_u8"], item["plan"] = None, None
            return item
        try:
            px = image_to_tensor(resize_with_padding(Image.open(path).convert("RGB"), self._target(idx)))
        except Exception:
            px = torch.zeros(3, *self._target(idx))
        item["pixel_values"] = px
        return item


def collate_raw(items):
    """collate for device_preprocess datasets: crops stay a list (ragged sizes); see device_batch"""
    import torch
    return {"image_u8": [i["image_u8"] for i in items], "plan": [i["plan"] for i in items], "target_w": items[0].get("target_w"),
            "labels": torch.stack([i["labels"] for i in items]),
            "text": [i["text"] for i in items], "image_path": [i["image_path"] for i in items]}


def device_batch(batch, preprocessor):
    """raw batch -> the reference's batch dict with pixel_values produced on the GPU (kzv.preprocess.DevicePreprocessor)"""
    ok = [k for k, im in enumerate(batch["image_u8"]) if im is not None]
    import torch
    n = len(batch["image_u8"])
    tw = batch.get("target_w") or preprocessor.target_w            # width buckets: the batch's own canvas width
    px = torch.zeros(n, 3, preprocessor.target_h, tw, device=preprocessor.device)
    if ok:
        got = preprocessor([batch["image_u8"][k] for k in ok], plans=[batch["plan"][k] for k in ok], target_w=tw)
        px[torch.tensor(ok, device=px.device)] = got
    out = {k: v for k, v in batch.items() if k not in ("image_u8", "plan", "target_w")}
    out["pixel_values"] = px
    return out


def collate(items):
    import torch
    if "image_u8" in items[0]:
        return collate_raw(items)
    return {"pixel_values": torch.stack([i["pixel_values"] for i in items]),
            "labels": torch.stack([i["labels"] for i in items]),
            "text": [i["text"] for i in items], "image_path": [i["image_path"] for i in items]}


class DeviceBatchLoader:
    """Wraps a loader of raw batches (device_preprocess datasets): every batch leaves with pixel_values made on the GPU."""

    def __init__(self, loader, preprocessor):
        self.loader, self.preprocessor = loader, preprocessor

    def __len__(self):
        return len(self.loader)

    def set_epoch(self, epoch):
        self.loader.set_epoch(epoch)

    def __iter__(self):
        for b in self.loader:
            yield device_batch(b, self.preprocessor)


class EpochSampler:
    """Index stream of one rank: ``DataLoader(shuffle=True)`` reshuffles every epoch (src/data/trocr_dataset.py:256-262)
    and Lightning wraps it in a DistributedSampler (seed + epoch permutation, padded to a multiple of the world size,
    rank r takes indices r, r + world, ...; ``set_epoch`` is called by the fit loop)."""

    def __init__(self, n: int, shuffle: bool, seed: int = 42, rank: int = 0, world: int = 1):
        self.n, self.shuffle, self.seed, self.rank, self.world, self.epoch = n, shuffle, seed, rank, world, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        return (self.n + self.world - 1) // self.world

    def __iter__(self):
        import torch
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        total = len(self) * self.world
        if total > len(idx) and idx:
            idx = idx + (idx * ((total - len(idx)) // len(idx) + 1))[: total - len(idx)]
        return iter(idx[self.rank:total:self.world])


def bucket_width(in_w: int, in_h: int, target_h: int, buckets) -> int:
    """Smallest bucket that holds the crop once it is scaled to the target height (ResizeWithPadding keeps the aspect
    ratio, src/data/trocr_dataset.py:24-53): crops wider than the widest bucket shrink into it like the reference."""
    need = in_w * target_h / max(1, in_h)
    for b in sorted(buckets):
        if need <= b:
            return b
    return max(buckets)


class BucketBatchSampler:
    """Batches of same-width-bucket samples (BASELINE.json configs[4]: bucketed 384-1024 px crops, dynamic patch-sequence
    length): `widths[i]` = bucket of sample i.  Each epoch shuffles inside the buckets (seed + epoch), cuts every bucket into
    batches (short last batch kept, like the reference's loaders), shuffles the batch order, pads the batch list to a multiple
    of the world size and deals batches round-robin to the ranks -- so every rank sees the same NUMBER of batches (the gradient
    all-reduce needs that) while batch shapes may differ between ranks."""

    def __init__(self, widths, batch_size: int, shuffle: bool = True, seed: int = 42, rank: int = 0, world: int = 1):
        self.widths, self.batch_size, self.shuffle, self.seed, self.rank, self.world, self.epoch = list(widths), batch_size, shuffle, seed, rank, world, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _batches(self):
        import torch
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        by = {}
        for i, w in enumerate(self.widths):
            by.setdefault(w, []).append(i)
        out = []
        for w in sorted(by):
            idx = by[w]
            if self.shuffle:
                idx = [idx[j] for j in torch.randperm(len(idx), generator=g).tolist()]
            out += [idx[k:k + self.batch_size] for k in range(0, len(idx), self.batch_size)]
        if self.shuffle:
            out = [out[j] for j in torch.randperm(len(out), generator=g).tolist()]
        pad = (-len(out)) % self.world
        return out + out[:pad]

    def __len__(self) -> int:
        return len(self._batches()) // self.world

    def __iter__(self):
        return iter(self._batches()[self.rank::self.world])


def make_loader(dataset, batch_size, shuffle, seed=42, rank=0, world=1, drop_last=False, num_workers=0, preprocessor=None):
    """DataLoader over an EpochSampler (per-epoch reshuffle, DistributedSampler-style shard).  drop_last defaults to
    False like the reference's loaders.  `preprocessor` (kzv.preprocess.DevicePreprocessor): for datasets built with
    device_preprocess=True.  The returned loader exposes ``set_epoch``."""
    from torch.utils.data import DataLoader
    if getattr(dataset, "widths", None) is not None:     # width buckets: batches of one width each
        sampler = BucketBatchSampler(dataset.widths, batch_size, shuffle, seed, rank, world)
        loader = DataLoader(dataset, batch_sampler=sampler, collate_fn=collate, num_workers=num_workers)
    else:
        sampler = EpochSampler(len(dataset), shuffle, seed, rank, world)
        loader = DataLoader(dataset, batch_size=batch_size, sampler=sampler, collate_fn=collate, drop_last=drop_last,
                            num_workers=num_workers)
    loader.set_epoch = sampler.set_epoch
    return DeviceBatchLoader(loader, preprocessor) if preprocessor is not None else loader
