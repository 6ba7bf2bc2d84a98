"""dev: gradients with the LM head's input gradient inside head_ce_kernel (KZV_HEAD_DGRAD=1, set per process) -- run twice and compare the printed checksums."""
import dataclasses, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv.config import tiny_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
for (B, Lh, vocab) in ((5, 30, 4300), (3, 12, 100), (70, 9, 777), (256, 128, 4300)):
    cfg = dataclasses.replace(tiny_config(), dec_hidden=256, dec_heads=4, dec_ffn=768, dec_layers=2, vocab=vocab, max_pos=130)
    with tempfile.TemporaryDirectory() as tmp:
        m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), init_seed=3, load_tokenizer=False)
    px, lab = synthetic_batch(cfg, B, Lh, seed=5, min_chars=1, max_chars=Lh - 2)
    m.train(); m.zero_grad(); m.trim_padding = False
    loss, _ = m.forward_loss(torch.from_numpy(px).cuda(), torch.from_numpy(lab).cuda(), want_logits=False, seed=13); m.backward(); torch.cuda.synchronize()
    g = m.flat_grads.double()
    print(f"KZV_HEAD_DGRAD={os.environ.get('KZV_HEAD_DGRAD')} B={B} L={Lh} V={vocab}: loss {float(loss):.7f} sum {float(g.sum()):.12e} abs {float(g.abs().sum()):.12e} max {float(g.abs().max()):.9e}")
