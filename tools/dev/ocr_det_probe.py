import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "kuzushiji-vision_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch
import _ocr_ddp_worker as W
m = W.make()
for rank in (0, 1, 0, 1):
    b = W.shard(0, rank, 2)
    m.train()
    o1 = m(b["images"]); l1 = o1["pred_logits"].clone(); x1 = o1["pred_boxes"].clone()
    o2 = m(b["images"]); l2 = o2["pred_logits"].clone(); x2 = o2["pred_boxes"].clone()
    torch.cuda.synchronize()
    print("rank", rank, "forward twice: logits max diff", float((l1 - l2).abs().max()), "boxes", float((x1 - x2).abs().max()), "|logits|", float(l1.abs().max()))
    tape1, tape2 = [], []
    m.forward(b["images"], _tape=tape1); m.forward(b["images"], _tape=tape2); torch.cuda.synchronize()
    for (k1, *r1), (k2, *r2) in zip(tape1[:-1], tape2[:-1]):
        ks1 = [r1[0]] if k1 == "stem" else [x for x in r1[1] if x is not None]
        ks2 = [r2[0]] if k2 == "stem" else [x for x in r2[1] if x is not None]
        for a, c in zip(ks1, ks2):
            print("   ", k1, tuple(a["y"].shape), "y diff", float((a["y"] - c["y"]).abs().max()), "out diff", float((a["out"].float() - c["out"].float()).abs().max()),
                  "mean diff", float((a["mean"] - c["mean"]).abs().max()), "cols diff", float((a["cols"].float() - c["cols"].float()).abs().max()))
