"""ocr_lightning row, CPU side: the folder dataset / pad-collate / greedy CTC decode against the behaviour the
reference's own tests pin (ocr_lightning/tests/test_dataset.py:14-147: temp-dir fixture of PNG + TXT + JSON triplets, skipping
rules, collate padding with [-1, -1, -1, -1] boxes), and the oracle's model against the reference's model tests
(ocr_lightning/tests/test_model.py:28-86: output keys / shapes on randn(2, 3, 64, 128), singles == batched in eval mode)."""
import json
import os

import numpy as np
import torch

from kzv.ocr_data import CHAR_TO_IDX, IDX_TO_CHAR, VOCAB, OcrDataset, OcrLoader, decode_ctc_output, ocr_collate_fn
from oracle.ocr_oracle import OCROracle


def _make_split(root, samples):
    from PIL import Image
    for book, name, size, text, boxes in samples:
        for sub in ("images", "labels", "bounding_boxes"):
            os.makedirs(os.path.join(root, sub, book), exist_ok=True)
        Image.new("RGB", size, color=(200, 30, 90)).save(os.path.join(root, "images", book, name + ".png"))
        if text is not None:
            with open(os.path.join(root, "labels", book, name + ".txt"), "w", encoding="utf-8") as f:
                f.write(text + "\n")
        if boxes is not None:
            with open(os.path.join(root, "bounding_boxes", book, name + ".json"), "w", encoding="utf-8") as f:
                json.dump(boxes, f)


def test_dataset_items_and_skipping_rules(tmp_path):
    root = str(tmp_path)
    _make_split(root, [("book1", "img1", (100, 50), "text1", [[10, 10, 50, 20]]),
                       ("book1", "img2", (120, 60), "te", [[5, 5, 10, 10], [20, 20, 30, 30]]),
                       ("book2", "nolabel", (30, 30), None, [[1, 2, 3, 4]]),
                       ("book2", "nobox", (30, 30), "x", None)])
    open(os.path.join(root, "images", "book1", "notes.txt"), "w").close()          # not an image: ignored
    ds = OcrDataset(root, char_to_idx=CHAR_TO_IDX)
    assert len(ds) == 2                                                             # the two incomplete samples are skipped
    items = sorted((ds[i] for i in range(2)), key=lambda d: d["image_path"])
    assert items[0]["image"].shape == (3, 50, 100) and items[0]["image"].dtype == torch.float32      # (C, H, W), ToTensor scaling
    assert abs(float(items[0]["image"][0, 0, 0]) - 200 / 255) < 1e-6 and items[0]["label_text"] == "text1"
    assert items[1]["bounding_boxes"] == [[5, 5, 10, 10], [20, 20, 30, 30]] and items[1]["image_path"].endswith("img2.png")


def test_collate_pads_images_and_boxes_like_the_reference(tmp_path):
    root = str(tmp_path)
    _make_split(root, [("b", "a", (100, 50), "text1", [[10, 10, 50, 20]]), ("b", "c", (120, 60), "", [[5, 5, 10, 10], [20, 20, 30, 30]])])
    ds = OcrDataset(root)
    batch = ocr_collate_fn(sorted((ds[i] for i in range(2)), key=lambda d: d["image_path"]))
    assert batch["images"].shape == (2, 3, 60, 120)
    assert torch.all(batch["images"][0, :, 50:, :] == 0) and torch.all(batch["images"][0, :, :, 100:] == 0)      # zero padding right / bottom
    assert batch["bounding_boxes_batch"].shape == (2, 2, 4) and batch["bounding_boxes_batch"].dtype == torch.float32
    assert batch["bounding_boxes_batch"][0, 1].tolist() == [-1, -1, -1, -1]                                  # the dummy box
    assert batch["bbox_counts"] == [1, 2] and batch["target_lengths"] == [5, 0] and batch["label_texts"] == ["text1", ""]
    assert len(list(OcrLoader(ds, 1))) == 2 and len(OcrLoader(ds, 2)) == 1
    empty = ocr_collate_fn([{"image": torch.zeros(3, 4, 4), "label_text": "a", "bounding_boxes": [], "image_path": "p"}])
    assert empty["bounding_boxes_batch"].shape == (1, 0) and empty["bbox_counts"] == [0]


def test_greedy_ctc_decode_known_answers():
    blank = CHAR_TO_IDX["<"]                     # the reference's VOCAB spells '<blank>' out: index 0 is '<' (train.py:15-17)
    assert blank == 0 and VOCAB.startswith("<blank>") and CHAR_TO_IDX.get("<blank>", 0) == 0
    idx = {0: "", 1: "a", 2: "b", 3: "c"}
    def onehot(seq, n=4):
        return torch.eye(n)[torch.tensor(seq)] * 5.0
    assert decode_ctc_output(onehot([2]), idx, 0) == "b" and decode_ctc_output(onehot([0]), idx, 0) == ""     # length-1 branch
    assert decode_ctc_output(onehot([1, 1, 0, 1, 2, 2, 0, 0, 3]), idx, 0) == "aabc"                           # repeats collapse unless a blank separates
    assert decode_ctc_output(onehot([0, 0, 0]), idx, 0) == ""
    assert decode_ctc_output(onehot([1, 2, 1]), {0: "", 1: "a"}, 0) == "aa"                                   # unknown index dropped but still breaks the run


def test_oracle_model_shapes_and_batch_consistency():
    """ocr_lightning/tests/test_model.py:28-76 on the oracle (full ResNet34 depth, random init, eval mode)."""
    torch.manual_seed(0)
    vocab = "<blank>" + "abcdefghijklmnopqrstuvwxyz0123456789"
    c2i = {ch: i for i, ch in enumerate(vocab)}
    m = OCROracle(len(c2i), c2i.get("<blank>", 0), max_boxes=10).eval()
    x = torch.randn(2, 3, 64, 128)
    with torch.no_grad():
        out = m(x)
        a, b = m(x[:1]), m(x[1:])
    assert out["pred_boxes"].shape == (2, 10, 4) and out["pred_logits"].shape == (2, 1, len(c2i))
    for k in ("pred_boxes", "pred_logits"):
        assert torch.allclose(out[k][0], a[k][0], atol=1e-5) and torch.allclose(out[k][1], b[k][0], atol=1e-5)
    keys = list(m.state_dict())
    assert keys[0] == "feature_extractor.0.weight" and "feature_extractor.5.0.downsample.1.running_var" in keys
    assert "recognition_rnn.weight_hh_l1_reverse" in keys and keys[-1] == "recognition_fc.bias"


def test_oracle_shared_step_edge_cases():
    """_shared_step's special cases (model.py:100-176): samples without boxes are left out of the box mean, empty labels out of
    the CTC batch, labels longer than the length-1 sequence have infinite CTC loss -> zero_infinity, all-empty -> 0."""
    torch.manual_seed(1)
    c2i = {ch: i for i, ch in enumerate("_abc")}
    m = OCROracle(4, 0, max_boxes=3, blocks=(1,), widths=(64,)).train()
    x = torch.randn(3, 3, 32, 32)
    gt = torch.tensor([[[1., 2, 3, 4], [5, 6, 7, 8]], [[-1, -1, -1, -1], [-1, -1, -1, -1]], [[0, 0, 1, 1], [-1, -1, -1, -1]]])
    total, loc, rec = m.shared_step({"images": x, "label_texts": ["a", "", "abc"], "bounding_boxes_batch": gt, "bbox_counts": [2, 0, 1]}, c2i)
    assert torch.isfinite(total) and float(loc) > 0 and float(rec) > 0
    _, _, rec2 = m.shared_step({"images": x, "label_texts": ["", "", ""], "bounding_boxes_batch": gt, "bbox_counts": [0, 0, 0]}, c2i)
    assert float(rec2) == 0.0
    _, _, rec3 = m.shared_step({"images": x, "label_texts": ["ab", "abc", "cc"], "bounding_boxes_batch": gt, "bbox_counts": [2, 0, 1]}, c2i)
    assert float(rec3) == 0.0                     # every target longer than T = 1: infinite, zeroed
