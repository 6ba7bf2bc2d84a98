"""TEST INFRASTRUCTURE (oracle side; never imported by the product path).

CPU statement of the reference's ``ocr_lightning/model.py::OCRModel`` in plain torch: the forward of model.py:61-88 and the
loss of ``_shared_step`` (:90-195), built from torch's own ``nn.Conv2d / BatchNorm2d / MaxPool2d / LSTM / Linear / CTCLoss /
SmoothL1Loss`` -- the reference's own dependencies for everything except the trunk.  The trunk: the reference takes
``torchvision.models.resnet34`` (absent in this image, and its weights are a download), so the BasicBlock topology (conv3x3-bn-relu,
conv3x3-bn, + identity or conv1x1/2-bn shortcut, relu; stem conv7x7/2-bn-relu-maxpool3x3/2; stages (3, 4, 6, 3) x (64, 128, 256, 512))
is RESTATED here from the published architecture: "parity unpinned" for the topology, pinned for every operator in it.
The reference cannot be imported to generate fixtures for this row (ordinary ModuleNotFoundError: torchvision, pytorch_lightning)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Act(nn.Module):
    """ReLU -- or, for mask-replay parity (tests/test_ocr_gpu.py), the multiplication by the 0/1 masks another implementation's
    ReLUs produced, taken in call order from ``masks`` (a list the test fills; empty -> plain ReLU).  A bf16 forward flips the
    sign of ~0.5 % of the pre-activations that sit next to zero, and each flipped mask element is a 100 % local error of the
    gradient: with the masks replayed, what remains is the rounding of the GEMM operands."""

    def __init__(self, masks, record=None):
        super().__init__()
        self.masks = masks
        self.record = record if record is not None else {}          # {"seen": list} -> the masks THIS forward applies, in call order

    def forward(self, x):
        seen = self.record.get("seen")
        if self.masks:
            mk = self.masks.pop(0)
            if seen is not None:
                seen.append(mk.bool())
            return x * mk.to(x.dtype)
        if seen is not None:
            seen.append(x > 0)
        return F.relu(x)


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride, masks=None, record=None):
        super().__init__()
        self.act = _Act(masks if masks is not None else [], record)
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        out = self.act(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.act(out + idn)


class OCROracle(nn.Module):
    """Same registration order and key names as the reference: feature_extractor = Sequential(conv1, bn1, relu, maxpool, layer1..4)."""

    def __init__(self, num_chars, blank, max_boxes=50, blocks=(3, 4, 6, 3), widths=(64, 128, 256, 512)):
        super().__init__()
        widths = tuple(widths[:len(blocks)])
        self.relu_masks = []            # mask replay: see _Act
        self.relu_record = {}           # set ["seen"] = [] to collect the masks a forward applies
        layers = [nn.Conv2d(3, widths[0], 7, 2, 3, bias=False), nn.BatchNorm2d(widths[0]), _Act(self.relu_masks, self.relu_record), nn.MaxPool2d(3, 2, 1)]
        cin = widths[0]
        for si, (nb, wd) in enumerate(zip(blocks, widths)):
            stage = []
            for bi in range(nb):
                stage.append(BasicBlock(cin, wd, 2 if (bi == 0 and si > 0) else 1, self.relu_masks, self.relu_record))
                cin = wd
            layers.append(nn.Sequential(*stage))
        self.feature_extractor = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.localization_head = nn.Linear(cin, max_boxes * 4)                                        # model.py:37
        self.recognition_rnn = nn.LSTM(input_size=cin, hidden_size=256, num_layers=2, bidirectional=True, batch_first=True)   # :40-47
        self.recognition_fc = nn.Linear(512, num_chars)                                               # :48
        self.max_boxes, self.blank = max_boxes, blank
        self.loc = nn.SmoothL1Loss(reduction="mean")                                                  # :50
        self.ctc = nn.CTCLoss(blank=blank, zero_infinity=True, reduction="mean")                      # :51-55

    def forward(self, images):                                                                       # model.py:61-88
        feats = torch.flatten(self.avgpool(self.feature_extractor(images)), 1)
        boxes = self.localization_head(feats).view(-1, self.max_boxes, 4)
        rnn_out, _ = self.recognition_rnn(feats.unsqueeze(1))
        return {"pred_boxes": boxes, "pred_logits": self.recognition_fc(rnn_out)}

    def shared_step(self, batch, char_to_idx):                                                       # model.py:90-195
        images, texts, gt, counts = batch["images"], batch["label_texts"], batch["bounding_boxes_batch"], batch["bbox_counts"]
        out = self(images)
        B = images.size(0)
        loc_sum, nval = 0.0, 0
        for i in range(B):
            n = min(int(counts[i]), self.max_boxes)
            if n == 0:
                continue
            loc_sum = loc_sum + self.loc(out["pred_boxes"][i, :n, :], gt[i, :n, :])
            nval += 1
        loc = loc_sum / nval if nval > 0 else torch.tensor(0.0, requires_grad=True)
        lp = F.log_softmax(out["pred_logits"], dim=2).permute(1, 0, 2)
        enc = [torch.tensor([char_to_idx.get(ch, self.blank) for ch in t], dtype=torch.long) for t in texts]
        lens = torch.tensor([len(e) for e in enc], dtype=torch.long)
        mx = int(lens.max()) if len(enc) else 0
        valid = lens > 0
        if mx == 0 or not bool(valid.any()):
            rec = torch.tensor(0.0, requires_grad=True)
        else:
            tg = torch.full((B, mx), self.blank, dtype=torch.long)
            for i, e in enumerate(enc):
                tg[i, :len(e)] = e
            lv = lens[valid]
            rec = self.ctc(lp[:, valid, :], tg[valid, :max(1, int(lv.max()))], torch.full((int(valid.sum()),), lp.size(0), dtype=torch.long), lv)
        return loc + rec, loc, rec
