"""Loader of tests/golden/tiny_trained.npz (tools/gen_golden_trained.py: the reference fitted on 8 crops; its own
teacher-forced logits / strings / CER and step-wise greedy decodes).  Weights are stored as bf16 bit patterns."""
import os

import numpy as np

from kzv.config import tiny_config
from kzv.data import synthetic_batch

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    g = np.load(os.path.join(HERE, "golden", "tiny_trained.npz"), allow_pickle=False)
    cfg = tiny_config()
    Lh = int(g["label_len"])
    sd = {}
    for k in g.files:
        if k.startswith("w/"):
            sd[k[2:]] = (g[k].astype(np.uint32) << 16).view(np.float32)
    px_fit, _ = synthetic_batch(cfg, 8, Lh, seed=int(g["fit_seed"]), min_chars=3, max_chars=Lh - 3)
    px_uns, _ = synthetic_batch(cfg, 4, Lh, seed=int(g["unseen_seed"]), min_chars=3, max_chars=Lh - 3)
    return g, cfg, sd, {"fit": (px_fit, g["labels"]), "unseen": (px_uns, g["labels_unseen"])}


def pad_to(ids, width, pad_id):
    out = np.full((ids.shape[0], width), pad_id, dtype=np.int64)
    out[:, :ids.shape[1]] = ids
    return out
