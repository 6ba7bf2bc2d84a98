"""N2 input pipeline: ResizeWithPadding + ToTensor + Normalize of the reference, byte-exact.

CPU: the numpy oracle and the library's host-side coefficient routine against the golden vectors produced with Pillow
itself (tools/gen_golden_resize.py), and against Pillow directly where it is importable.
GPU: the HIP path (kzv.preprocess.DevicePreprocessor -> kzv_preprocess_lines) against both."""
import os

import numpy as np
import pytest

from oracle import resize_oracle as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "resize_kat.npz")
TARGETS = ((64, 640), (32, 64))


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _crops(gold):
    return [gold[f"in{i}"] for i in range(int(gold["n"]))]


def _expected(gold, i, th, tw):
    """golden fp32 tensor: stored for the small target, lut256[canvas] (both from the generator) for the large one"""
    key = f"tensor{i}_{th}x{tw}"
    if key in gold.files:
        return gold[key]
    return gold["lut256"][gold[f"canvas{i}_{th}x{tw}"]].transpose(2, 0, 1)


def test_oracle_matches_pillow_golden_vectors(gold):
    lut = R.normalise_lut()
    for i, img in enumerate(_crops(gold)):
        for th, tw in TARGETS:
            got = R.transform(img, th, tw)
            assert np.array_equal(got, _expected(gold, i, th, tw))                        # fp32 bit patterns
            assert np.array_equal(lut[gold[f"canvas{i}_{th}x{tw}"]].transpose(2, 0, 1), got)
    assert np.array_equal(lut, gold["lut256"])


def test_oracle_matches_pillow_directly_on_fresh_sizes():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for _ in range(25):
        h, w = int(rng.integers(4, 200)), int(rng.integers(4, 1200))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nw, nh, _, _ = R.target_geometry(w, h, 64, 640)
        if nw < 1 or nh < 1:
            continue
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.Resampling.LANCZOS))
        assert np.array_equal(R.resize_lanczos(img, nw, nh), ref)


def test_host_coefficients_equal_the_oracle():
    """kzv_lanczos_coeffs is host code inside libkzv.so (no GPU): same double arithmetic as Pillow's precompute_coeffs."""
    from kzv import preprocess as PP
    for in_size, out_size in [(1000, 640), (100, 64), (37, 64), (640, 640), (5, 64), (2000, 640), (63, 64), (65, 64), (17, 3), (1, 1), (3, 640)]:
        b, k = PP.lanczos_coeffs(in_size, out_size)
        rb, rk = R.coefficients(in_size, out_size)
        assert np.array_equal(b, rb) and np.array_equal(k, rk), (in_size, out_size)


def test_geometry_and_degenerate_crops():
    from kzv import preprocess as PP
    assert PP.target_geometry(800, 100, 64, 640) == R.target_geometry(800, 100, 64, 640) == (512, 64, 64, 0)
    assert PP.target_geometry(100, 300, 64, 640) == (21, 64, 309, 0)
    with pytest.raises(ValueError):
        PP.target_geometry(5000, 1, 64, 640)       # height collapses to 0: Pillow refuses such a resize as well


@pytest.mark.gpu
def test_device_pipeline_is_byte_exact(gold):
    import torch
    from kzv.preprocess import DevicePreprocessor
    crops = _crops(gold)
    for th, tw in TARGETS:
        pre = DevicePreprocessor((th, tw))
        out = pre(crops).cpu().numpy()
        for i in range(len(crops)):
            assert np.array_equal(out[i], _expected(gold, i, th, tw)), (i, th, tw)
    # a ragged batch of fresh crops against the oracle, written into a caller-provided tensor; unchanged-size crops skip a pass
    rng = np.random.default_rng(3)
    fresh = [rng.integers(0, 256, (int(rng.integers(8, 180)), int(rng.integers(8, 1500)), 3), dtype=np.uint8) for _ in range(33)]
    fresh += [rng.integers(0, 256, (64, 640, 3), dtype=np.uint8), rng.integers(0, 256, (64, 200, 3), dtype=np.uint8),
              rng.integers(0, 256, (10, 640, 3), dtype=np.uint8)]
    pre = DevicePreprocessor((64, 640))
    buf = torch.empty(len(fresh), 3, 64, 640, device="cuda")
    got = pre(fresh, out=buf).cpu().numpy()
    for i, img in enumerate(fresh):
        assert np.array_equal(got[i], R.transform(img, 64, 640)), i
    with pytest.raises(ValueError):
        pre([np.zeros((4, 4), np.uint8)])


@pytest.mark.gpu
def test_dataset_with_device_preprocess_equals_the_host_transform(tmp_path):
    """LineCsvDataset(device_preprocess=True) -> collate -> device_batch == the same dataset with the PIL transform on
    the host (pixel for pixel), including the all-zero fallback for an unreadable image."""
    import torch
    from PIL import Image
    from transformers import AutoTokenizer
    from kzv.config import tiny_config
    from kzv.data import LineCsvDataset, build_decoder_dir, collate, device_batch
    from kzv.preprocess import DevicePreprocessor
    tok = AutoTokenizer.from_pretrained(build_decoder_dir(str(tmp_path / "dec"), tiny_config()))
    root = tmp_path / "img"
    root.mkdir()
    rng = np.random.default_rng(5)
    rows = []
    for k, (h, w) in enumerate([(100, 30), (210, 17), (64, 32), (33, 64), (500, 40)]):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / f"c{k}.png")
        rows.append(f"c{k}.png,\"['U+4E00', 'U+4E01']\"")
    (root / "broken.png").write_bytes(b"not an image")       # exists, cannot be decoded -> all-zero tensor (trocr_dataset.py:182-185)
    rows.append("broken.png,\"['U+4E00']\"")
    rows.append("missing.png,\"['U+4E00']\"")              # does not exist -> dropped (trocr_dataset.py:135)
    csv = tmp_path / "column_info.csv"
    csv.write_text("column_image,unicode_ids\n" + "\n".join(rows) + "\n", encoding="utf-8")
    kw = dict(image_size=(64, 32), max_length=8, split="train", train_ratio=1.0, val_ratio=0.0, test_ratio=0.0, resolve="image_root")
    host = LineCsvDataset(str(csv), str(root), tok, **kw)
    dev = LineCsvDataset(str(csv), str(root), tok, device_preprocess=True, **kw)
    assert len(host) == len(dev) == 6
    hb = collate([host[i] for i in range(6)])
    db = device_batch(collate([dev[i] for i in range(6)]), DevicePreprocessor((64, 32)))
    assert torch.equal(db["pixel_values"].cpu(), hb["pixel_values"])
    assert torch.equal(db["labels"], hb["labels"]) and db["text"] == hb["text"]
    assert torch.all(db["pixel_values"][5] == 0)
