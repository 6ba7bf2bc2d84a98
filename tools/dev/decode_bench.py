"""N1 timing: greedy and beam-4 generate() on a ViT-B + 6-layer decoder batch of 256 crops, graph-replayed vs eager step."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kuzushiji-vision_amd")]
import torch
from kzv.config import vit_b_config
from kzv.data import build_decoder_dir, synthetic_batch
from kzv.model import TrOCRModel
cfg = vit_b_config(dec_layers=6)
with tempfile.TemporaryDirectory() as tmp:
    m = TrOCRModel(cfg.encoder_config_dict(), build_decoder_dir(os.path.join(tmp, "d"), cfg), load_tokenizer=False)
px = torch.from_numpy(synthetic_batch(cfg, 256, 128, seed=1)[0]).cuda()
m.eval()
for graph in (os.environ.get("DEC_GRAPHS", "0,1").split(",")):
    os.environ["KZV_DECODE_GRAPH"] = graph
    for beams in [int(b) for b in os.environ.get("DEC_BEAMS", "1,4").split(",")]:
        ts = []
        for rep in range(int(os.environ.get("DEC_REPS", "6"))):     # the first repetitions capture the graph and ramp the clocks
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = m.generate(px, max_length=128, num_beams=beams, early_stopping=False)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        dt = sorted(ts[2:])[len(ts[2:]) // 2] if len(ts) > 2 else ts[-1]
        print(f"graph={graph} beams={beams}: {dt * 1e3:.1f} ms for {out.shape[1]} tokens x 256 crops ({256 / dt:.0f} img/s; encoder included; "
              f"median of the last {max(1, len(ts) - 2)} of {len(ts)} runs)")
