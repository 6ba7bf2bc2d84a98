"""Host mirror of the reference's ``ocr_lightning/model.py::OCRModel``: ResNet34 trunk ->
global average pool -> {Linear localisation head, 2-layer bidirectional LSTM over a length-1 sequence -> Linear -> CTC}, with the
SmoothL1 + CTC loss of ``_shared_step`` and ``optim.Adam`` -- every FLOP in libkzv.so (csrc/ocr.hip + the MFMA GEMMs), none in
torch: torch tensors are device buffers here, nothing else.  There is no fallback: without the HIP library this module raises.

Same surface as the reference class (model.py:8-213): ``OCRModel(char_to_idx, idx_to_char, learning_rate=1e-4, max_boxes=50)``,
``forward(images) -> {'pred_boxes': [B, max_boxes, 4], 'pred_logits': [B, 1, num_chars]}``, ``training_step / validation_step /
test_step(batch, batch_idx)`` on the dict ``ocr_collate_fn`` builds, ``configure_optimizers()``, ``hparams`` (incl. the derived
``blank_char_idx`` / ``num_chars``), ``state_dict()`` / ``load_state_dict()`` under the reference's key names
(``feature_extractor.<i>...`` of ``nn.Sequential(*list(resnet34.children())[:-2])``, ``localization_head.*``,
``recognition_rnn.weight_ih_l0[_reverse]`` ..., ``recognition_fc.*``).

What differs, and why:
  * the ResNet34 weights: the reference downloads ``ResNet34_Weights.DEFAULT`` (model.py:31); there is no network here and no
    torchvision, so the trunk starts from torchvision's own initialisation recipe (kaiming-normal fan-out convolutions, unit BatchNorm)
    unless a state_dict is loaded.  The TOPOLOGY is restated from the published ResNet34 (BasicBlock x (3, 4, 6, 3), widths 64..512):
    parity of the trunk is pinned against torch's own Conv2d / BatchNorm2d / MaxPool2d modules composed the same way
    (oracle/ocr_oracle.py), not against torchvision -- "unpinned" in that one respect.
  * precision: the reference trains this model in fp32 (pl.Trainer default).  ``precision="fp32"`` is that arithmetic: fp32
    activations and GEMM operands on the f32-input matrix instruction (csrc/gemm_f32.hip: exact fp32 products, fp32 accumulation),
    the tests hold it to fp32 tolerances.  ``precision="bf16"`` (the default of round 3, kept for speed) rounds the GEMM operands to
    bf16 with fp32 accumulation (the engine's MFMA path) -- narrower than the reference, tolerances are the bf16 ones.
"""
from __future__ import annotations

import ctypes as C
import math
from types import SimpleNamespace

from . import _lib as L

RESNET34_BLOCKS = (3, 4, 6, 3)
RESNET34_WIDTHS = (64, 128, 256, 512)
LSTM_HIDDEN = 256            # model.py:42
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _r64(n: int) -> int:
    return (n + 63) // 64 * 64


class _Conv:
    """One nn.Conv2d (bias=False) + the nn.BatchNorm2d behind it, by state_dict prefix."""

    def __init__(self, conv_key, bn_key, cin, cout, k, stride, pad):
        self.conv_key, self.bn_key = conv_key, bn_key
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.Kp = _r64(k * k * cin)


class OCRModel:
    def __init__(self, char_to_idx, idx_to_char, learning_rate=1e-4, max_boxes=50, blocks=RESNET34_BLOCKS, widths=RESNET34_WIDTHS,
                 device="cuda", init_seed=0, precision="fp32"):
        import torch
        self.lib = L.load()                     # raises without libkzv.so: no CPU path
        self.device = torch.device(device)
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' (the reference's arithmetic) or 'bf16' (bf16 GEMM operands)")
        self.precision = precision
        self.f32 = precision == "fp32"
        self.adt = torch.float32 if self.f32 else torch.bfloat16           # activations / GEMM operands
        num_chars = len(char_to_idx)
        blank = char_to_idx.get("<blank>", 0)   # model.py:15
        if blank != 0 and list(char_to_idx.keys())[0] != "<blank>":
            print(f"Warning: blank_char_idx is {blank} but char_to_idx suggests it might be 0. Ensure consistency for CTCLoss.")
        self.hparams = SimpleNamespace(char_to_idx=char_to_idx, idx_to_char=idx_to_char, learning_rate=learning_rate, max_boxes=max_boxes,
                                       blank_char_idx=blank, num_chars=num_chars)
        self.blocks, self.widths = tuple(blocks), tuple(widths[:len(blocks)])
        self.loc_loss_weight = self.rec_loss_weight = 1.0          # model.py:58-59
        self.training = True
        self.logged: dict[str, list[float]] = {}
        self._optimizer = None
        self._step = 0
        self._graphs = {}
        # ---- parameter table in the reference's registration order ------------------------------------------------
        self.convs: list[_Conv] = []
        shapes: list[tuple[str, tuple[int, ...]]] = []
        self.buffers: dict[str, "torch.Tensor"] = {}

        def add_conv(conv_key, bn_key, cin, cout, k, stride, pad):
            c = _Conv(conv_key, bn_key, cin, cout, k, stride, pad)
            self.convs.append(c)
            shapes.append((conv_key + ".weight", (cout, cin, k, k)))
            return c

        def add_bn(key, ch):
            shapes.append((key + ".weight", (ch,))); shapes.append((key + ".bias", (ch,)))
            self.buffers[key + ".running_mean"] = torch.zeros(ch, device=self.device)
            self.buffers[key + ".running_var"] = torch.ones(ch, device=self.device)
            self.buffers[key + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=self.device)

        fe = "feature_extractor"
        stem = self.widths[0]
        self.stem = add_conv(f"{fe}.0", f"{fe}.1", 3, stem, 7, 2, 3); add_bn(f"{fe}.1", stem)
        self.stages = []
        cin = stem
        for si, (nb, wd) in enumerate(zip(self.blocks, self.widths)):
            stage = []
            for bi in range(nb):
                stride = 2 if (bi == 0 and si > 0) else 1
                pre = f"{fe}.{4 + si}.{bi}"
                c1 = add_conv(pre + ".conv1", pre + ".bn1", cin, wd, 3, stride, 1); add_bn(pre + ".bn1", wd)
                c2 = add_conv(pre + ".conv2", pre + ".bn2", wd, wd, 3, 1, 1); add_bn(pre + ".bn2", wd)
                ds = None
                if stride != 1 or cin != wd:
                    ds = add_conv(pre + ".downsample.0", pre + ".downsample.1", cin, wd, 1, stride, 0); add_bn(pre + ".downsample.1", wd)
                stage.append((c1, c2, ds))
                cin = wd
            self.stages.append(stage)
        self.feat = cin                                               # resnet.fc.in_features (512 for ResNet34)
        if self.feat % 64:
            raise ValueError("the trunk's output width must be a multiple of 64 (the GEMM's K)")
        shapes.append(("localization_head.weight", (max_boxes * 4, self.feat))); shapes.append(("localization_head.bias", (max_boxes * 4,)))
        for layer in range(2):
            inp = self.feat if layer == 0 else 2 * LSTM_HIDDEN
            for sfx in ("", "_reverse"):
                shapes.append((f"recognition_rnn.weight_ih_l{layer}{sfx}", (4 * LSTM_HIDDEN, inp)))
                shapes.append((f"recognition_rnn.weight_hh_l{layer}{sfx}", (4 * LSTM_HIDDEN, LSTM_HIDDEN)))
                shapes.append((f"recognition_rnn.bias_ih_l{layer}{sfx}", (4 * LSTM_HIDDEN,)))
                shapes.append((f"recognition_rnn.bias_hh_l{layer}{sfx}", (4 * LSTM_HIDDEN,)))
        shapes.append(("recognition_fc.weight", (num_chars, 2 * LSTM_HIDDEN))); shapes.append(("recognition_fc.bias", (num_chars,)))
        self.offsets, off = {}, 0
        for name, shp in shapes:
            n = int(math.prod(shp))
            self.offsets[name] = (off, shp)
            off += (n + 63) // 64 * 64                                # 256-byte aligned entries
        self.total = off
        self.flat_params = torch.zeros(off, device=self.device)
        self.flat_grads = torch.zeros(off, device=self.device)
        self._init_parameters(init_seed)
        # bf16 operand copies (refreshed after every optimizer step / load)
        self.Cp = _r64(num_chars)                                    # logits columns incl. zero padding (GEMM K of the fc input gradient)
        self._w16: dict[str, "torch.Tensor"] = {}
        self.sync_weights()

    # ------------------------------------------------------------------------------------------------ parameters
    def _view(self, flat, name):
        o, shp = self.offsets[name]
        return flat[o:o + int(math.prod(shp))].view(shp)

    def param(self, name):
        return self._view(self.flat_params, name)

    def grad(self, name):
        return self._view(self.flat_grads, name)

    def _init_parameters(self, seed):
        """torchvision's ResNet recipe (kaiming_normal_(fan_out, relu) convolutions, BatchNorm weight 1 / bias 0) and torch's
        defaults for nn.Linear (U(+-1/sqrt(fan_in))) and nn.LSTM (U(+-1/sqrt(hidden)))."""
        import torch
        g = torch.Generator().manual_seed(seed)
        for name, (o, shp) in self.offsets.items():
            if name.endswith(".weight") and len(shp) == 4:
                v = torch.randn(shp, generator=g) * math.sqrt(2.0 / (shp[0] * shp[2] * shp[3]))
            elif name.startswith("recognition_rnn"):
                v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(LSTM_HIDDEN)
            elif name.startswith(("localization_head", "recognition_fc")):
                fan_in = self.offsets[name.rsplit(".", 1)[0] + ".weight"][1][1]
                v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
            elif name.endswith(".weight"):                            # BatchNorm gamma
                v = torch.ones(shp)
            else:                                                      # BatchNorm beta
                v = torch.zeros(shp)
            self.param(name).copy_(v.to(self.device))

    def state_dict(self):
        sd = {name: self.param(name).detach().clone() for name in self.offsets}
        sd.update({k: v.detach().clone() for k, v in self.buffers.items()})
        return sd

    def load_state_dict(self, sd, strict=True):
        import torch
        missing = [k for k in list(self.offsets) + list(self.buffers) if k not in sd]
        extra = [k for k in sd if k not in self.offsets and k not in self.buffers]
        if strict and (missing or extra):
            raise KeyError(f"load_state_dict: missing {missing[:4]} unexpected {extra[:4]}")
        for k, v in sd.items():
            t = torch.as_tensor(v)
            if k in self.offsets:
                self.param(k).copy_(t.to(self.device, torch.float32).reshape(self.offsets[k][1]))
            elif k in self.buffers:
                self.buffers[k].copy_(t.to(self.device, self.buffers[k].dtype))
        self.sync_weights()

    def sync_weights(self):
        """bf16 copies of every GEMM operand: packed [Cout, Kp] (+ transposed) conv weights, Linear / LSTM weights and their transposes."""
        import torch
        lib, st = self.lib, L.stream_handle()
        self._set_precision()
        for c in self.convs:
            self._w16.setdefault(c.conv_key, torch.empty(c.cout, c.Kp, dtype=self.adt, device=self.device))
            self._w16.setdefault(c.conv_key + ".T", torch.empty(c.Kp, c.cout, dtype=self.adt, device=self.device))
        n = len(self.convs)
        if n <= 40:                                              # every convolution's packed copies in one launch
            t = self._conv_tables()
            L.check(lib.kzv_ocr_conv_weight_multi(n, t.w, t.wp, t.wpT, t.geom, st), "conv_weight_multi")
        else:
            for c in self.convs:
                L.check(lib.kzv_ocr_conv_weight(self.param(c.conv_key + ".weight").data_ptr(), self._w16[c.conv_key].data_ptr(), self._w16[c.conv_key + ".T"].data_ptr(),
                                                c.cout, c.cin, c.k, c.k, c.Kp, st), "conv_weight")
        for name, (o, shp) in self.offsets.items():
            if len(shp) != 2 or "weight_hh" in name:
                continue
            rows, cols = shp
            rp = _r64(rows) if name == "recognition_fc.weight" else rows
            w = self._w16.get(name)
            if w is None:
                w = self._w16[name] = torch.zeros(rp, cols, dtype=self.adt, device=self.device)
                self._w16[name + ".T"] = torch.zeros(cols, rp, dtype=self.adt, device=self.device)
            L.check(lib.kzv_ocr_cast_bf16(self.param(name).data_ptr(), w.data_ptr(), rows * cols, st), "cast")
            if rp == rows:
                L.check(lib.kzv_ocr_cast_transpose(self.param(name).data_ptr(), self._w16[name + ".T"].data_ptr(), rows, cols, st), "cast_T")
            else:                                                      # zero-padded rows -> zero-padded columns of the transpose
                self._w16[name + ".T"][:, :rows].copy_(w[:rows].t())

    # ------------------------------------------------------------------------------------------------ module surface
    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    def parameters(self):
        return [self.param(n) for n in self.offsets]

    def log(self, name, value, **kw):
        self.logged.setdefault(name, []).append(float(value))

    def __call__(self, images):
        return self.forward(images)

    def _conv_tables(self):
        """Host arrays for the one-launch forms (packed weights, packed weight gradients): built once, the buffers never move."""
        t = getattr(self, "_ctab", None)
        if t is None:
            import torch
            n = len(self.convs)
            PA = C.c_void_p * n
            self._gp = [torch.zeros(c.cout, c.Kp, device=self.device) for c in self.convs]           # packed weight gradients (zeroed per backward)
            geom = (C.c_int32 * (5 * n))(*[v for c in self.convs for v in (c.cout, c.cin, c.k, c.k, c.Kp)])
            t = self._ctab = SimpleNamespace(
                w=PA(*[self.param(c.conv_key + ".weight").data_ptr() for c in self.convs]), wp=PA(*[self._w16[c.conv_key].data_ptr() for c in self.convs]),
                wpT=PA(*[self._w16[c.conv_key + ".T"].data_ptr() for c in self.convs]), gp=PA(*[g.data_ptr() for g in self._gp]),
                g=PA(*[self.grad(c.conv_key + ".weight").data_ptr() for c in self.convs]), geom=geom, index={id(c): i for i, c in enumerate(self.convs)})
        return t

    def _set_precision(self):
        """The element type of the library's OCR kernels is a process-wide switch: set it before every pass of this model."""
        L.check(self.lib.kzv_ocr_set_precision(1 if self.f32 else 0), "ocr_set_precision")

    # ------------------------------------------------------------------------------------------------ GEMM helpers
    def _gemm_nt(self, A, B16, Cout, M, N, K, epi, bias=None, resid=None, n_valid=0):
        a = L.kzv_gemm_nt_args(A=A.data_ptr(), lda=A.stride(0), B=B16.data_ptr(), ldb=B16.stride(0), C=Cout.data_ptr(), ldc=Cout.stride(0),
                               bias=None if bias is None else bias.data_ptr(), resid=None if resid is None else resid.data_ptr(),
                               ldr=0 if resid is None else resid.stride(0), aux=None, ldaux=0, M=M, N=N, K=K, n_valid=n_valid, drop_p=0.0, drop_key=0)
        L.check((self.lib.kzv_gemm_nt_f32 if self.f32 else self.lib.kzv_gemm_nt)(C.byref(a), epi, L.stream_handle()), "gemm_nt")

    def _gemm_tn(self, P16, Q16, OUT, Mtok, N, K, n_store=0, dbias=None):
        a = L.kzv_gemm_tn_args(P=P16.data_ptr(), ldp=P16.stride(0), Q=Q16.data_ptr(), ldq=Q16.stride(0), OUT=OUT.data_ptr(), ldo=OUT.stride(0),
                               Mtok=Mtok, N=N, K=K, n_store=n_store, dbias=None if dbias is None else dbias.data_ptr())
        L.check((self.lib.kzv_gemm_tn_f32 if self.f32 else self.lib.kzv_gemm_tn)(C.byref(a), L.stream_handle()), "gemm_tn")

    # ------------------------------------------------------------------------------------------------ forward
    def _conv_bn(self, c: _Conv, x16, N, H, W, relu, resid16=None, keep=None):
        """x16 bf16 [N*H*W, cin] -> bf16 [N*Ho*Wo, cout]; `keep` (a dict) receives what the backward needs."""
        import torch
        lib, st, dev = self.lib, L.stream_handle(), self.device
        Ho, Wo = (H + 2 * c.pad - c.k) // c.stride + 1, (W + 2 * c.pad - c.k) // c.stride + 1
        M = N * Ho * Wo
        cols = torch.empty(M, c.Kp, dtype=self.adt, device=dev)
        L.check(lib.kzv_ocr_im2col(x16.data_ptr(), cols.data_ptr(), N, H, W, c.cin, c.k, c.k, c.stride, c.pad, c.Kp, st), "im2col")
        y = torch.empty(M, c.cout, dtype=torch.float32, device=dev)
        self._gemm_nt(cols, self._w16[c.conv_key], y, M, c.cout, c.Kp, L.EPI_F32)
        mean, rstd = torch.empty(c.cout, device=dev), torch.empty(c.cout, device=dev)
        out = torch.empty(M, c.cout, dtype=self.adt, device=dev)
        scratch = torch.empty(lib.kzv_ocr_bn_scratch_floats(M, c.cout), device=dev)
        L.check(lib.kzv_ocr_bn_fwd(y.data_ptr(), M, c.cout, self.param(c.bn_key + ".weight").data_ptr(), self.param(c.bn_key + ".bias").data_ptr(),
                                   self.buffers[c.bn_key + ".running_mean"].data_ptr(), self.buffers[c.bn_key + ".running_var"].data_ptr(),
                                   mean.data_ptr(), rstd.data_ptr(), L.ptr(resid16), out.data_ptr(), int(relu), int(self.training), BN_EPS, BN_MOMENTUM,
                                   scratch.data_ptr(), st), "bn_fwd")
        if self.training:
            self.buffers[c.bn_key + ".num_batches_tracked"] += 1
        if keep is not None:
            keep.update(cols=cols, y=y, mean=mean, rstd=rstd, out=out, geom=(N, H, W, Ho, Wo), relu=relu)
        return out, Ho, Wo

    def _trunk(self, images, tape):
        import torch
        lib, st, dev = self.lib, L.stream_handle(), self.device
        N, Cc, H, W = images.shape
        if Cc != 3:
            raise ValueError(f"images must be [B, 3, H, W], got {tuple(images.shape)}")
        x = torch.empty(N * H * W, 3, dtype=self.adt, device=dev)
        L.check(lib.kzv_ocr_nchw_to_nhwc(images.data_ptr(), x.data_ptr(), N, 3, H, W, st), "nhwc")
        k = {}
        a, H1, W1 = self._conv_bn(self.stem, x, N, H, W, True, keep=k)
        Hp, Wp = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        pooled = torch.empty(N * Hp * Wp, self.stem.cout, dtype=self.adt, device=dev)
        idx = torch.empty(N * Hp * Wp, self.stem.cout, dtype=torch.uint8, device=dev)
        L.check(lib.kzv_ocr_maxpool_fwd(a.data_ptr(), pooled.data_ptr(), idx.data_ptr(), N, H1, W1, self.stem.cout, st), "maxpool")
        tape.append(("stem", k, idx, (H1, W1, Hp, Wp)))
        cur, Hc, Wc = pooled, Hp, Wp
        for stage in self.stages:
            for c1, c2, ds in stage:
                k1, k2, kd = {}, {}, ({} if ds is not None else None)
                a1, Ho, Wo = self._conv_bn(c1, cur, N, Hc, Wc, True, keep=k1)
                idn = cur
                if ds is not None:
                    idn, _, _ = self._conv_bn(ds, cur, N, Hc, Wc, False, keep=kd)
                out, _, _ = self._conv_bn(c2, a1, N, Ho, Wo, True, resid16=idn, keep=k2)
                tape.append(("block", (c1, c2, ds), (k1, k2, kd), (Hc, Wc)))
                cur, Hc, Wc = out, Ho, Wo
        return cur, N, Hc, Wc

    def forward(self, images, _tape=None):
        """model.py:61-88.  images fp32 [B, 3, H, W] (CPU or device)."""
        import torch
        lib, st, dev = self.lib, L.stream_handle(), self.device
        images = images.to(dev, torch.float32).contiguous()
        self._set_precision()
        tape = _tape if _tape is not None else []
        act, N, Hc, Wc = self._trunk(images, tape)
        feat32 = torch.empty(N, self.feat, device=dev)
        feat16 = torch.empty(N, self.feat, dtype=self.adt, device=dev)
        L.check(lib.kzv_ocr_avgpool_fwd(act.data_ptr(), feat32.data_ptr(), feat16.data_ptr(), N, Hc * Wc, self.feat, st), "avgpool")
        mb4 = self.hparams.max_boxes * 4
        boxes = torch.empty(N, mb4, device=dev)
        self._gemm_nt(feat16, self._w16["localization_head.weight"], boxes, N, mb4, self.feat, L.EPI_F32, bias=self.param("localization_head.bias"))
        # 2-layer bidirectional LSTM on the length-1 sequence [B, 1, feat] (model.py:73-75)
        lstm = []
        x16 = feat16
        for layer in range(2):
            K = x16.shape[1]
            h32 = torch.empty(N, 2 * LSTM_HIDDEN, device=dev)
            h16 = torch.empty(N, 2 * LSTM_HIDDEN, dtype=self.adt, device=dev)
            gates_l = []
            for d, sfx in enumerate(("", "_reverse")):
                nm = f"recognition_rnn.weight_ih_l{layer}{sfx}"
                gates = torch.empty(N, 4 * LSTM_HIDDEN, device=dev)
                self._gemm_nt(x16, self._w16[nm], gates, N, 4 * LSTM_HIDDEN, K, L.EPI_F32, bias=self.param(f"recognition_rnn.bias_ih_l{layer}{sfx}"))
                bhh = self.param(f"recognition_rnn.bias_hh_l{layer}{sfx}")
                L.check(lib.kzv_ocr_lstm_cell_fwd(gates.data_ptr(), bhh.data_ptr(), h32[:, d * LSTM_HIDDEN:].data_ptr(), h16[:, d * LSTM_HIDDEN:].data_ptr(),
                                                  2 * LSTM_HIDDEN, N, LSTM_HIDDEN, st), "lstm_cell")
                gates_l.append(gates)
            lstm.append((x16, gates_l, h16))
            x16 = h16
        nc = self.hparams.num_chars
        logits = torch.zeros(N, self.Cp, device=dev)                 # columns >= num_chars stay 0
        self._gemm_nt(x16, self._w16["recognition_fc.weight"], logits, N, self.Cp, 2 * LSTM_HIDDEN, L.EPI_F32, bias=self._bias_padded(), n_valid=nc)
        if _tape is not None:
            _tape.append(("head", dict(feat16=feat16, lstm=lstm, logits=logits, boxes=boxes, geom=(N, Hc, Wc), act=act)))
        return {"pred_boxes": boxes.view(N, self.hparams.max_boxes, 4), "pred_logits": logits[:, :nc].reshape(N, 1, nc)}

    def relu_masks_of_last_step(self):
        """Parity hook: the 0/1 masks of every ReLU of the last _shared_step's forward, in call order, as [N, C, H, W] bool tensors
        (what oracle/ocr_oracle.py::_Act replays)."""
        tape, _, _ = self._pending
        out = []

        def nchw(k):
            N, H, W, Ho, Wo = k["geom"]
            return (k["out"].float() > 0).view(N, Ho, Wo, -1).permute(0, 3, 1, 2).cpu()
        for kind, *rest in tape[:-1]:
            if kind == "stem":
                out.append(nchw(rest[0]))
            else:
                (_, _, _), (k1, k2, _), _ = rest
                out.append(nchw(k1)); out.append(nchw(k2))
        return out

    def _bias_padded(self):
        import torch
        b = getattr(self, "_fcb", None)
        if b is None:
            b = self._fcb = torch.zeros(self.Cp, device=self.device)
        b[:self.hparams.num_chars].copy_(self.param("recognition_fc.bias"))
        return b

    # ------------------------------------------------------------------------------------------------ loss + backward
    CTC_LD = 32          # target columns handed to the CTC kernel: only labels of <= 1 character align with the length-1 sequence,
                         # longer ones are infeasible whatever they spell (loss 0, no gradient) -- their first characters are enough

    def _host_labels(self, label_texts):
        """Labels encoded like model.py:135-139 (unknown characters -> blank): the host half of the recognition loss."""
        hp = self.hparams
        enc = [[hp.char_to_idx.get(ch, hp.blank_char_idx) for ch in text] for text in label_texts]
        lens = [len(e) for e in enc]
        nvalid = sum(1 for n in lens if n > 0)
        return SimpleNamespace(enc=enc, lens=lens, nvalid=nvalid, Lmax=max(lens) if lens else 0,
                               gs=[self.rec_loss_weight / (max(n, 1) * nvalid) if n > 0 else 0.0 for n in lens])

    def _label_tensors(self, lab, B, into=None):
        """Device inputs of the CTC launch (into: the static buffers of a captured step, refilled in place)."""
        import torch
        hp, dev = self.hparams, self.device
        tg = torch.full((B, self.CTC_LD), hp.blank_char_idx, dtype=torch.int64)
        for i, e in enumerate(lab.enc):
            k = min(len(e), self.CTC_LD)
            tg[i, :k] = torch.tensor(e[:k], dtype=torch.int64)
        tl = torch.tensor(lab.lens, dtype=torch.int64)
        gs = torch.tensor(lab.gs, dtype=torch.float32)
        if into is None:
            return SimpleNamespace(tg=tg.to(dev), tl=tl.to(dev), il=torch.ones(B, dtype=torch.int64, device=dev), gs=gs.to(dev))
        into.tg.copy_(tg, non_blocking=True); into.tl.copy_(tl, non_blocking=True); into.gs.copy_(gs, non_blocking=True)
        return into

    def _loss_launch(self, head, gt, counts, lt, has_ctc):
        """The device half of model.py:90-195 (launches only, no host read: capturable): SmoothL1 over each sample's first
        min(count, max_boxes) boxes (mean over the samples that have any) + CTC (blank, zero_infinity, 'mean') over the samples with a
        non-empty label, on the length-1 log-probabilities.  Returns device tensors (loc [1], nll [B], dboxes, dlogits)."""
        import torch
        lib, st, dev, hp = self.lib, L.stream_handle(), self.device, self.hparams
        boxes, logits = head["boxes"], head["logits"]
        B = boxes.shape[0]
        loc = torch.zeros(1, device=dev)
        dboxes = torch.empty(B, hp.max_boxes * 4, device=dev)
        gtb = gt.shape[1] if gt.dim() == 3 else 0
        L.check(lib.kzv_ocr_smooth_l1_boxes(boxes.data_ptr(), hp.max_boxes, gt.data_ptr() if gtb else None, gtb, counts.data_ptr(), B, loc.data_ptr(),
                                            dboxes.data_ptr(), st), "smooth_l1")
        nc = hp.num_chars
        dlogits = torch.zeros(B, self.Cp, device=dev)
        nll = torch.zeros(B, device=dev)
        if has_ctc:
            lp = torch.empty(B, nc, device=dev)
            lg = logits[:, :nc].contiguous()
            L.check(lib.kzv_ocr_log_softmax(lg.data_ptr(), lp.data_ptr(), B, nc, st), "log_softmax")
            scratch = torch.empty(2 * B * 3, device=dev)            # 2 * B * T * (2 * min(L, T) + 1) with T = 1
            dl = torch.empty(B, nc, device=dev)
            L.check(lib.kzv_ocr_ctc(lp.data_ptr(), lt.tg.data_ptr(), self.CTC_LD, lt.il.data_ptr(), lt.tl.data_ptr(), 1, B, nc, hp.blank_char_idx, 1,
                                    self.CTC_LD, scratch.data_ptr(), nll.data_ptr(), lt.gs.data_ptr(), dl.data_ptr(), st), "ctc")
            dlogits[:, :nc].copy_(dl)
        return loc, nll, dboxes, dlogits

    def _loss_values(self, loc, nll, lab, step_name):
        """Host half: the three logged values (one device -> host read each)."""
        rec_val = 0.0
        if lab.nvalid > 0 and lab.Lmax > 0:
            nl = nll.cpu()
            rec_val = float(sum(float(nl[i]) / max(lab.lens[i], 1) for i in range(len(lab.lens)) if lab.lens[i] > 0) / lab.nvalid)
        loc_val = float(loc.item())
        total = self.loc_loss_weight * loc_val + self.rec_loss_weight * rec_val
        self.log(f"{step_name}/loc_loss", loc_val); self.log(f"{step_name}/rec_loss", rec_val); self.log(f"{step_name}/total_loss", total)
        return total

    def _shared_step(self, batch, batch_idx, step_name):
        """model.py:90-195 (see _loss_launch)."""
        import torch
        dev = self.device
        images = batch["images"]
        gt = batch["bounding_boxes_batch"].to(dev, torch.float32).contiguous()
        counts = torch.as_tensor(batch["bbox_counts"], dtype=torch.int32).to(dev)
        B = images.shape[0]
        tape = []
        self.forward(images, _tape=tape)
        lab = self._host_labels(batch["label_texts"])
        has_ctc = lab.nvalid > 0 and lab.Lmax > 0
        lt = self._label_tensors(lab, B) if has_ctc else None
        loc, nll, dboxes, dlogits = self._loss_launch(tape[-1][1], gt, counts, lt, has_ctc)
        total = self._loss_values(loc, nll, lab, step_name)
        self._pending = (tape, dboxes * self.loc_loss_weight if self.loc_loss_weight != 1.0 else dboxes, dlogits)
        return total

    def training_step(self, batch, batch_idx):
        return self._shared_step(batch, batch_idx, "train")

    def validation_step(self, batch, batch_idx):
        return self._shared_step(batch, batch_idx, "val")

    def test_step(self, batch, batch_idx):
        return self._shared_step(batch, batch_idx, "test")

    def zero_grad(self):
        self.flat_grads.zero_()

    def backward(self):
        """Hand-written backward of the last _shared_step (what ``loss.backward()`` does for the reference): fills ``flat_grads``."""
        import torch
        lib, st, dev = self.lib, L.stream_handle(), self.device
        self._set_precision()
        multi = len(self.convs) <= 40
        if multi:
            self._conv_tables()
            torch._foreach_zero_(self._gp)
        tape, dboxes, dlogits = self._pending
        head = tape[-1][1]
        N, Hc, Wc = head["geom"]
        H2 = 2 * LSTM_HIDDEN
        # ---- recognition_fc (gradients accumulate straight into flat_grads: n_store = the un-padded row count)
        dl16 = self._to16(dlogits)                                   # [B, Cp], zero beyond num_chars
        last16 = head["lstm"][1][2]
        nc = self.hparams.num_chars
        self._gemm_tn(dl16, last16, self.grad("recognition_fc.weight"), N, self.Cp, H2, n_store=nc, dbias=self.grad("recognition_fc.bias"))
        dh = torch.empty(N, H2, device=dev)
        self._gemm_nt(dl16, self._w16["recognition_fc.weight.T"], dh, N, H2, self.Cp, L.EPI_F32)
        # ---- LSTM layers, top down
        for layer in (1, 0):
            x16, gates_l, _ = head["lstm"][layer]
            K = x16.shape[1]
            dx = None
            for d, sfx in enumerate(("", "_reverse")):
                dg = torch.empty(N, 4 * LSTM_HIDDEN, dtype=self.adt, device=dev)
                bhh = self.param(f"recognition_rnn.bias_hh_l{layer}{sfx}")
                L.check(lib.kzv_ocr_lstm_cell_bwd(gates_l[d].data_ptr(), bhh.data_ptr(), dh[:, d * LSTM_HIDDEN:].data_ptr(), H2, dg.data_ptr(), N, LSTM_HIDDEN, st), "lstm_bwd")
                nm = f"recognition_rnn.weight_ih_l{layer}{sfx}"
                self._gemm_tn(dg, x16, self.grad(nm), N, 4 * LSTM_HIDDEN, K, dbias=self.grad(f"recognition_rnn.bias_ih_l{layer}{sfx}"))
                nx = torch.empty(N, K, device=dev)
                self._gemm_nt(dg, self._w16[nm + ".T"], nx, N, K, 4 * LSTM_HIDDEN, L.EPI_RESID if dx is not None else L.EPI_F32, resid=dx)
                dx = nx
            dh = dx
        # b_ih and b_hh enter the gates as a sum: equal gradients (W_hh multiplies the zero state: its gradient stays 0)
        for layer in range(2):
            for sfx in ("", "_reverse"):
                self.grad(f"recognition_rnn.bias_hh_l{layer}{sfx}").copy_(self.grad(f"recognition_rnn.bias_ih_l{layer}{sfx}"))
        dfeat = dh                                                     # [B, feat] from the recognition branch
        # ---- localisation head
        db16 = self._to16(dboxes)
        mb4 = self.hparams.max_boxes * 4
        mb4p = _r64(mb4)
        if mb4p != mb4:
            pad = torch.zeros(N, mb4p, dtype=self.adt, device=dev); pad[:, :mb4].copy_(db16); db16 = pad
        self._gemm_tn(db16, head["feat16"], self.grad("localization_head.weight"), N, mb4p, self.feat, n_store=mb4, dbias=self.grad("localization_head.bias"))
        wT = self._w16["localization_head.weight.T"]                   # [feat, mb4]
        if mb4p != mb4:
            wTp = torch.zeros(self.feat, mb4p, dtype=self.adt, device=dev); wTp[:, :mb4].copy_(wT); wT = wTp
        dfeat2 = torch.empty(N, self.feat, device=dev)
        self._gemm_nt(db16, wT, dfeat2, N, self.feat, mb4p, L.EPI_RESID, resid=dfeat)
        # ---- trunk
        da = torch.empty(N * Hc * Wc, self.feat, device=dev)
        L.check(lib.kzv_ocr_avgpool_bwd(dfeat2.data_ptr(), da.data_ptr(), N, Hc * Wc, self.feat, st), "avgpool_bwd")
        for kind, *rest in reversed(tape[:-1]):
            if kind == "block":
                (c1, c2, ds), (k1, k2, kd), (Hin, Win) = rest
                dz2, dy2 = self._bn_bwd(c2, k2, da)
                da1 = self._conv_bwd(c2, k2, dy2, need_dx=True)
                _, dy1 = self._bn_bwd(c1, k1, da1)
                if ds is not None:
                    _, dyd = self._bn_bwd(ds, kd, dz2)
                    dx = self._conv_bwd(ds, kd, dyd, need_dx=True)
                    da = self._conv_bwd(c1, k1, dy1, need_dx=True, accumulate_into=dx)
                else:
                    da = self._conv_bwd(c1, k1, dy1, need_dx=True, accumulate_into=dz2)      # identity shortcut: dx = dz2 + conv path
            else:
                k, idx, (H1, W1, Hp, Wp) = rest
                dpre = torch.empty(N * H1 * W1, self.stem.cout, device=dev)
                L.check(lib.kzv_ocr_maxpool_bwd(da.data_ptr(), idx.data_ptr(), dpre.data_ptr(), N, H1, W1, self.stem.cout, st), "maxpool_bwd")
                _, dy = self._bn_bwd(self.stem, k, dpre)
                self._conv_bwd(self.stem, k, dy, need_dx=False)
        if multi:
            t = self._conv_tables()
            L.check(lib.kzv_ocr_conv_wgrad_unpack_multi(len(self.convs), t.gp, t.g, t.geom, st), "wgrad_unpack_multi")
        self._pending = None

    def _bn_bwd(self, c: _Conv, k, da):
        import torch
        M, dev = k["y"].shape[0], self.device
        dz = torch.empty(M, c.cout, device=dev)
        dy = torch.empty(M, c.cout, dtype=self.adt, device=dev)
        scratch = torch.empty(self.lib.kzv_ocr_bn_scratch_floats(M, c.cout), device=dev)
        L.check(self.lib.kzv_ocr_bn_bwd(da.data_ptr(), k["out"].data_ptr(), k["y"].data_ptr(), M, c.cout, k["mean"].data_ptr(), k["rstd"].data_ptr(),
                                        self.param(c.bn_key + ".weight").data_ptr(), dz.data_ptr(), self.grad(c.bn_key + ".weight").data_ptr(),
                                        self.grad(c.bn_key + ".bias").data_ptr(), dy.data_ptr(), int(k["relu"]), int(self.training), scratch.data_ptr(),
                                        L.stream_handle()), "bn_bwd")
        return dz, dy

    def _to16(self, x32):
        import torch
        out = torch.empty(x32.shape, dtype=self.adt, device=self.device)
        L.check(self.lib.kzv_ocr_cast_bf16(x32.data_ptr(), out.data_ptr(), x32.numel(), L.stream_handle()), "cast")
        return out

    def _conv_bwd(self, c: _Conv, k, dy16, need_dx, accumulate_into=None):
        import torch
        lib, st, dev = self.lib, L.stream_handle(), self.device
        N, H, W, Ho, Wo = k["geom"]
        M = N * Ho * Wo
        if len(self.convs) <= 40:                                # packed gradient buffers zeroed at the start of backward(), unpacked in one launch at its end
            gp = self._gp[self._conv_tables().index[id(c)]]
            self._gemm_tn(dy16, k["cols"], gp, M, c.cout, c.Kp)
        else:
            gp = torch.zeros(c.cout, c.Kp, device=dev)
            self._gemm_tn(dy16, k["cols"], gp, M, c.cout, c.Kp)
            L.check(lib.kzv_ocr_conv_wgrad_unpack(gp.data_ptr(), self.grad(c.conv_key + ".weight").data_ptr(), c.cout, c.cin, c.k, c.k, c.Kp, st), "wgrad_unpack")
        if not need_dx:
            return None
        dcols = torch.empty(M, c.Kp, device=dev)
        self._gemm_nt(dy16, self._w16[c.conv_key + ".T"], dcols, M, c.Kp, c.cout, L.EPI_F32)
        dx = accumulate_into if accumulate_into is not None else torch.empty(N * H * W, c.cin, device=dev)
        L.check(lib.kzv_ocr_col2im(dcols.data_ptr(), dx.data_ptr(), N, H, W, c.cin, c.k, c.k, c.stride, c.pad, c.Kp, int(accumulate_into is not None), st), "col2im")
        return dx

    # ------------------------------------------------------------------------------------------------ optimizer
    def configure_optimizers(self):
        """optim.Adam(self.parameters(), lr) (model.py:196-198): betas (0.9, 0.999), eps 1e-8, no weight decay."""
        import torch
        self._optimizer = SimpleNamespace(m=torch.zeros_like(self.flat_params), v=torch.zeros_like(self.flat_params), step_count=0,
                                          lr=self.hparams.learning_rate, betas=(0.9, 0.999), eps=1e-8)
        return self._optimizer

    def optimizer_state_dict(self):
        """torch.optim.Adam's own ``state_dict()`` layout (what Lightning's ModelCheckpoint stores under ``optimizer_states[0]`` for
        the reference, ocr_lightning/train.py:118-140): per parameter -- in the reference's registration order -- ``step``,
        ``exp_avg``, ``exp_avg_sq``; one param group."""
        import torch
        o = self._optimizer or self.configure_optimizers()
        names = list(self.offsets)
        state = {}
        if o.step_count > 0:
            for i, n in enumerate(names):
                state[i] = {"step": torch.tensor(float(o.step_count)), "exp_avg": self._view(o.m, n).detach().cpu().clone(),
                            "exp_avg_sq": self._view(o.v, n).detach().cpu().clone()}
        group = {"lr": o.lr, "betas": tuple(o.betas), "eps": o.eps, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """The inverse: Adam moments and step count from a checkpoint the reference (or this engine) wrote."""
        import torch
        o = self._optimizer or self.configure_optimizers()
        names = list(self.offsets)
        g = sd["param_groups"][0]
        if len(g["params"]) != len(names):
            raise KeyError(f"optimizer state holds {len(g['params'])} parameters, the model {len(names)}")
        o.lr, o.betas, o.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        self.hparams.learning_rate = o.lr
        steps = set()
        o.m.zero_(); o.v.zero_()
        for i, pid in enumerate(g["params"]):
            st = sd["state"].get(pid)
            if st is None:
                continue
            self._view(o.m, names[i]).copy_(torch.as_tensor(st["exp_avg"]).to(self.device, torch.float32).reshape(self.offsets[names[i]][1]))
            self._view(o.v, names[i]).copy_(torch.as_tensor(st["exp_avg_sq"]).to(self.device, torch.float32).reshape(self.offsets[names[i]][1]))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one flat Adam step serves every parameter here")
        o.step_count = steps.pop() if steps else 0

    def optimizer_step(self, _bc=None):
        """optim.Adam.step + the refresh of the operand copies.  _bc: a device [2] buffer holding this step's bias corrections
        (1 - beta1^t, sqrt(1 - beta2^t)) -- the captured step passes one so that no step-dependent scalar is baked into the graph."""
        o = self._optimizer or self.configure_optimizers()
        if _bc is None:
            o.step_count += 1
            L.check(self.lib.kzv_ocr_adam(self.flat_params.data_ptr(), self.flat_grads.data_ptr(), o.m.data_ptr(), o.v.data_ptr(), self.total, o.lr,
                                          o.betas[0], o.betas[1], o.eps, o.step_count, L.stream_handle()), "adam")
        else:
            L.check(self.lib.kzv_ocr_adam_dev(self.flat_params.data_ptr(), self.flat_grads.data_ptr(), o.m.data_ptr(), o.v.data_ptr(), self.total, o.lr,
                                              o.betas[0], o.betas[1], o.eps, _bc.data_ptr(), L.stream_handle()), "adam_dev")
        self.sync_weights()

    def fit_step(self, batch, batch_idx=0):
        """One optimisation step the way Lightning drives the reference: zero_grad, training_step, backward, [DDP gradient mean when
        a process group is up: pl.Trainer(devices=N) of ocr_lightning/train.py:132-140 -- one all-reduce of the flat gradient buffer,
        BatchNorm statistics stay per rank as in plain DDP], Adam.

        Single process: from the third step of a batch geometry on, the ~700 launches of the step are replayed from ONE captured
        hipGraph (``use_graph``, default on): the eager step is bound by the host issuing its launches from Python, not by the GPU."""
        import torch.distributed as dist
        self.train()
        if self.use_graph and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return self._fit_step_graph(batch)
        return self._fit_step_eager(batch, batch_idx)

    def _fit_step_eager(self, batch, batch_idx=0):
        self.zero_grad()
        loss = self.training_step(batch, batch_idx)
        self.backward()
        self.allreduce_grads()
        self.optimizer_step()
        return loss

    use_graph = True
    GRAPH_WARM_STEPS = 2          # eager steps of a geometry before it is captured (library workspaces reach their final size)
    MAX_GRAPHS = 8                # captured geometries kept (each owns its activations' memory pool); further ones run launch by launch

    def _fit_step_graph(self, batch):
        import torch
        dev = self.device
        images = batch["images"].to(dev, torch.float32).contiguous()
        gt = batch["bounding_boxes_batch"].to(dev, torch.float32).contiguous()
        B = images.shape[0]
        lab = self._host_labels(batch["label_texts"])
        has_ctc = lab.nvalid > 0 and lab.Lmax > 0
        key = (tuple(images.shape), tuple(gt.shape), has_ctc)
        ent = self._graphs.setdefault(key, SimpleNamespace(seen=0, graph=None))
        ent.seen += 1
        if ent.graph is None and (ent.seen <= self.GRAPH_WARM_STEPS or sum(1 for e in self._graphs.values() if e.graph is not None) >= self.MAX_GRAPHS):
            return self._fit_step_eager(batch)
        o = self._optimizer or self.configure_optimizers()
        if ent.graph is None:
            ent.images, ent.gt = torch.empty_like(images), torch.empty_like(gt)
            ent.counts = torch.zeros(B, dtype=torch.int32, device=dev)
            ent.lt = SimpleNamespace(tg=torch.zeros(B, self.CTC_LD, dtype=torch.int64, device=dev), tl=torch.zeros(B, dtype=torch.int64, device=dev),
                                     il=torch.ones(B, dtype=torch.int64, device=dev), gs=torch.zeros(B, device=dev))
            ent.bc = torch.ones(2, device=dev)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.zero_grad()
                tape = []
                self.forward(ent.images, _tape=tape)
                ent.loc, ent.nll, dboxes, dlogits = self._loss_launch(tape[-1][1], ent.gt, ent.counts, ent.lt, has_ctc)
                self._pending = (tape, dboxes * self.loc_loss_weight if self.loc_loss_weight != 1.0 else dboxes, dlogits)
                self.backward()
                self.optimizer_step(_bc=ent.bc)
            ent.graph = g
        ent.images.copy_(images, non_blocking=True); ent.gt.copy_(gt, non_blocking=True)
        ent.counts.copy_(torch.as_tensor(batch["bbox_counts"], dtype=torch.int32), non_blocking=True)
        if has_ctc:
            self._label_tensors(lab, B, into=ent.lt)
        o.step_count += 1
        ent.bc.copy_(torch.tensor([1.0 - o.betas[0] ** o.step_count, math.sqrt(1.0 - o.betas[1] ** o.step_count)], dtype=torch.float32), non_blocking=True)
        ent.graph.replay()
        return self._loss_values(ent.loc, ent.nll, lab, "train")

    def allreduce_grads(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(self.flat_grads, op=dist.ReduceOp.SUM)
            else:                                             # gloo rehearsal: through host memory
                g = self.flat_grads.cpu()
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
                self.flat_grads.copy_(g.to(self.device))
            self.flat_grads.mul_(1.0 / dist.get_world_size())
