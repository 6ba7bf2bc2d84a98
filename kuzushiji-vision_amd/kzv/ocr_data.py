"""Input side of the ocr_lightning model: the folder dataset and the padding collate of
``ocr_lightning/dataset.py`` and the greedy CTC decode of ``ocr_lightning/predict.py:21-55`` -- same names, same dict keys, same
skipping rules, so the reference's own dataset tests (ocr_lightning/tests/test_dataset.py) read the same against this module."""
from __future__ import annotations

import json
import os

# the placeholder character set of ocr_lightning/train.py:15-17 ('<blank>' spelled out character by character, index 0 = '<')
VOCAB = '<blank>' + 'abcdefghijklmnopqrstuvwxyz0123456789' + '帝都書肆尚書堂梓 .,:;!?\'"`-()'
CHAR_TO_IDX = {char: idx for idx, char in enumerate(VOCAB)}
IDX_TO_CHAR = {idx: char for idx, char in enumerate(VOCAB)}


_IMAGE_SUFFIXES = {".png", ".jpg", ".jpeg"}
# sidecar files of a page image: (sub-directory of the split, suffix, what the reference calls it when it is missing)
_SIDECARS = (("labels", ".txt", "Label file"), ("bounding_boxes", ".json", "Bounding box file"))


def _page_samples(split_dir):
    """Yield (image, label, boxes) path triples of one split: <split>/images/<book>/<page>.<png|jpg|jpeg> together with
    <split>/labels/<book>/<page>.txt and <split>/bounding_boxes/<book>/<page>.json, in directory-listing order (dataset.py:14-44).
    A page that lacks a sidecar is reported with the reference's warning and left out."""
    from pathlib import Path
    root = Path(split_dir)
    for book in (root / "images").iterdir():
        if not book.is_dir():
            continue
        for page in book.iterdir():
            if page.suffix.lower() not in _IMAGE_SUFFIXES:
                continue
            sidecars = [root / sub / book.name / (page.stem + suffix) for sub, suffix, _ in _SIDECARS]
            missing = next((what for path, (_, _, what) in zip(sidecars, _SIDECARS) if not path.exists()), None)
            if missing is not None:
                print(f"Warning: {missing} not found for image {page}, skipping sample.")
                continue
            yield (str(page), *map(str, sidecars))


class OcrDataset:
    """Folder dataset of ocr_lightning/dataset.py: same constructor, ``file_samples`` triples, ``__getitem__`` dict and skipping rule."""

    def __init__(self, data_split_dir, image_transforms=None, char_to_idx=None):
        self.data_split_dir = str(data_split_dir)
        self.image_transforms = image_transforms
        self.char_to_idx = char_to_idx
        self.file_samples = list(_page_samples(self.data_split_dir))

    def __len__(self):
        return len(self.file_samples)

    def __getitem__(self, idx):
        import numpy as np
        import torch
        from PIL import Image
        image_path, label_path, bbox_path = self.file_samples[idx]
        image = Image.open(image_path).convert("RGB")
        with open(label_path, "r", encoding="utf-8") as f:
            label_text = f.read().strip()
        with open(bbox_path, "r", encoding="utf-8") as f:
            bounding_boxes = json.load(f)
        if self.image_transforms:
            image = self.image_transforms(image)
        else:      # transforms.ToTensor(): HWC uint8 -> CHW float32 in [0, 1]
            image = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).to(torch.float32) / 255.0
        return {"image": image, "label_text": label_text, "bounding_boxes": bounding_boxes, "image_path": image_path}


def ocr_collate_fn(batch):
    """dataset.py:78-130: images zero-padded (right / bottom) to the batch's largest height and width, box lists padded with
    [-1, -1, -1, -1] to the longest one; target_lengths = character counts."""
    import torch
    images = [item["image"] for item in batch]
    label_texts = [item["label_text"] for item in batch]
    boxes_list = [item["bounding_boxes"] for item in batch]
    image_paths = [item["image_path"] for item in batch]
    max_h = max((img.shape[1] for img in images), default=0)
    max_w = max((img.shape[2] for img in images), default=0)
    padded = []
    for img in images:
        c, h, w = img.shape
        out = torch.zeros(c, max_h, max_w, dtype=img.dtype)
        out[:, :h, :w] = img
        padded.append(out)
    images_tensor = torch.stack(padded)
    bbox_counts = [len(b) for b in boxes_list]
    max_bboxes = max(bbox_counts) if bbox_counts else 0
    padded_boxes = []
    for b in boxes_list:
        cur = list(b)
        while len(cur) < max_bboxes:
            cur.append([-1, -1, -1, -1])
        padded_boxes.append(cur)
    boxes_tensor = torch.tensor(padded_boxes, dtype=torch.float32)
    return {"images": images_tensor, "label_texts": label_texts, "bounding_boxes_batch": boxes_tensor,
            "target_lengths": [len(t) for t in label_texts], "bbox_counts": bbox_counts, "image_paths": image_paths}


def decode_ctc_output(logits, idx_to_char, blank_idx):
    """predict.py:21-55: greedy CTC decode of logits [sequence_length, num_classes] -- argmax per step, repeats collapsed unless a
    blank separates them, blanks dropped; the model's length-1 sequences give one character or ''."""
    import torch
    logits = torch.as_tensor(logits)
    pred = torch.argmax(logits, dim=1).tolist()
    if logits.ndim == 2 and logits.size(0) == 1:
        return "" if pred[0] == blank_idx else idx_to_char.get(pred[0], "")
    out, last = [], None
    for ci in pred:
        if ci == blank_idx:
            last = None
            continue
        if ci == last:
            continue
        ch = idx_to_char.get(ci)
        if ch:
            out.append(ch)
        last = ci
    return "".join(out)


class OcrLoader:
    """DataLoader(dataset, batch_size, shuffle, collate_fn=ocr_collate_fn) of ocr_lightning/train.py:65-82, single process."""

    def __init__(self, dataset, batch_size, shuffle=False, seed=0, rank=0, world=1):
        self.dataset, self.batch_size, self.shuffle, self.seed, self.epoch = dataset, batch_size, shuffle, seed, 0
        self.rank, self.world = rank, world                 # DistributedSampler: every rank takes each world-th sample of the epoch's order

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        n = (len(self.dataset) + self.world - 1) // self.world
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        import random
        order = list(range(len(self.dataset)))
        if self.shuffle:
            random.Random(self.seed * 7919 + self.epoch).shuffle(order)
        if self.world > 1:                                   # padded to equal length like DistributedSampler (drop_last=False)
            total = (len(order) + self.world - 1) // self.world * self.world
            order = (order + order[:total - len(order)])[self.rank::self.world]
        for i in range(0, len(order), self.batch_size):
            yield ocr_collate_fn([self.dataset[j] for j in order[i:i + self.batch_size]])
