"""Host-side mirror of the reference's ``TrOCRModel`` (src/models/trocr_model.py:205-460).

Same constructor, same ``forward`` contract, same step / hook / utility methods, same ``state_dict`` key
names -- but every FLOP runs in libkzv's hand-written HIP kernels through the C ABI of include/kzv.h.
torch is used for device memory, streams and (in kzv.trainer) torch.distributed only; there is no
autograd and no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
import types
from typing import Any

from . import _lib as L
from . import params as P
from .config import ModelConfig, load_decoder_config


class _Cfg(types.SimpleNamespace):
    pass


class TrOCRModel:
    """TrOCR Model with ViT Encoder and RoBERTa Decoder (MI355X engine).

    Mirrors src/models/trocr_model.py:208-217 (constructor), :258-321 (forward), :323-460 (steps, CER,
    optimizer, mode hooks, decode_predictions).
    """

    def __init__(self, encoder_config: dict[str, Any], decoder_path: str, learning_rate: float = 1e-4,
                 beta1: float = 0.9, beta2: float = 0.999, epsilon: float = 1e-8, weight_decay: float = 0,
                 *, device: str = "cuda", init_seed: int = 42, load_tokenizer: bool = True, width_buckets=None,
                 fp8: bool = False):
        import torch
        self.hparams = types.SimpleNamespace(encoder_config=encoder_config, decoder_path=decoder_path,
                                             learning_rate=learning_rate, beta1=beta1, beta2=beta2,
                                             epsilon=epsilon, weight_decay=weight_decay)
        dec_cfg = load_decoder_config(decoder_path)          # FileNotFoundError like scripts/train_trocr.py:88-89
        self.cfg = ModelConfig.from_reference(encoder_config, dec_cfg)
        self.cfg.validate()
        self.tokenizer = None
        if load_tokenizer:
            from transformers import AutoTokenizer                      # trocr_model.py:222
            self.tokenizer = AutoTokenizer.from_pretrained(decoder_path)
        # attribute surface used by scripts/test_trocr_setup.py:118-120
        self.encoder = types.SimpleNamespace(config=_Cfg(hidden_size=self.cfg.enc_hidden), num_patches=self.cfg.num_patches)
        self.decoder = types.SimpleNamespace(config=_Cfg(hidden_size=self.cfg.dec_hidden, vocab_size=self.cfg.vocab))
        self.training = True
        self.device = torch.device(device)
        self.logged: dict[str, list[float]] = {}
        self._optimizer = None
        self._step_seed = 0
        self.trim_padding = True
        self._len_cache = (None, 0)
        # Width buckets (BASELINE.json configs[4]; an extension -- a reference model has one image size): encoder_config's
        # image_size is the WIDEST crop; batches whose width is one of `width_buckets` (multiples of the patch width, <= it)
        # run with fewer patch tokens and the position rows of the same (h, w) cells (include/kzv.h: kzv_set_image_width).
        self.width_buckets = tuple(sorted(int(w) for w in width_buckets)) if width_buckets else None
        if self.width_buckets:
            for w in self.width_buckets:
                if w % self.cfg.patch_w or not (self.cfg.patch_w <= w <= self.cfg.image_w):
                    raise ValueError(f"width bucket {w} must be a multiple of {self.cfg.patch_w} and <= {self.cfg.image_w}")

        lib = L.load()
        c = self.cfg
        self._ccfg = L.kzv_config(
            image_h=c.image_h, image_w=c.image_w, patch_h=c.patch_h, patch_w=c.patch_w, channels=c.channels,
            enc_hidden=c.enc_hidden, enc_layers=c.enc_layers, enc_heads=c.enc_heads, enc_ffn=c.enc_ffn,
            dec_hidden=c.dec_hidden, dec_layers=c.dec_layers, dec_heads=c.dec_heads, dec_ffn=c.dec_ffn,
            vocab=c.vocab, max_pos=c.max_pos, type_vocab=c.type_vocab, pad_id=c.pad_id,
            enc_hidden_dropout=c.enc_hidden_dropout, enc_attn_dropout=c.enc_attn_dropout,
            dec_hidden_dropout=c.dec_hidden_dropout, dec_attn_dropout=c.dec_attn_dropout, ln_eps=c.ln_eps)
        h = C.c_void_p()
        L.check(lib.kzv_model_create(C.byref(self._ccfg), C.byref(h)), "kzv_model_create")
        self._h = h
        # fp8 weight path (BASELINE.json configs[4]; an extension -- the reference trains in bf16 autocast): the encoder's QKV,
        # fc1 and fc2 forward GEMMs read e4m3 weights and activations (include/kzv.h: kzv_set_fp8); backward stays bf16.
        # fp8 = True / 1: forward GEMMs; 2: also the MLP's two input-gradient GEMMs
        self.fp8 = int(fp8)
        if self.fp8:
            L.check(lib.kzv_set_fp8(h, self.fp8), "kzv_set_fp8")
        self._offsets, total = P.param_offsets(c)
        if lib.kzv_param_total(h) != total:
            raise L.KzvError("parameter table mismatch between kzv/params.py and libkzv")
        self.flat_params = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._ws = None
        self._bound = (0, 0)
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        # seeded init (the reference seeds 42: scripts/train_trocr.py:78); weights arrive via load_state_dict
        self.flat_params.copy_(torch.from_numpy(P.recipe_flat(c, init_seed)))
        # decoder weights from decoder_path if present (AutoModelForCausalLM.from_pretrained, trocr_model.py:231)
        self._maybe_load_decoder_weights(decoder_path)

    # ------------------------------------------------------------------ plumbing
    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.load().kzv_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _maybe_load_decoder_weights(self, path: str) -> None:
        f = os.path.join(path, "model.safetensors")
        if not os.path.exists(f):
            return
        from safetensors.torch import load_file
        sd = load_file(f)
        mine = self.state_dict_views()
        for k, v in sd.items():
            name = "decoder." + k
            if name in mine and tuple(mine[name].shape) == tuple(v.shape):
                mine[name].copy_(v.to(self.device, dtype=mine[name].dtype))

    def fp8_act_scales(self):
        """Per-layer multipliers the last forward quantised the encoder's GELU outputs with (parity hook)."""
        import torch
        out = torch.empty(self.cfg.enc_layers, dtype=torch.float32, device=self.device)
        L.check(L.load().kzv_fp8_act_scales(self._h, out.data_ptr(), L.stream_handle()), "kzv_fp8_act_scales")
        torch.cuda.synchronize()
        return out.cpu()

    def _bind(self, batch: int, label_len: int) -> None:
        import torch
        if self._bound == (batch, label_len):
            return
        lib = L.load()
        need = lib.kzv_workspace_bytes(self._h, batch, label_len)
        if need < 0:
            L.check(-1, "kzv_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        L.check(lib.kzv_model_bind(self._h, self.flat_params.data_ptr(), self.flat_grads.data_ptr(),
                                   self._ws.data_ptr(), self._ws.numel(), batch, label_len), "kzv_model_bind")
        self._bound = (batch, label_len)
        self.sync_weights()

    def sync_weights(self) -> None:
        if self._bound != (0, 0):
            L.check(L.load().kzv_model_sync_weights(self._h, L.stream_handle()), "sync_weights")

    def zero_grad(self) -> None:
        self.flat_grads.zero_()

    # ------------------------------------------------------------------ nn.Module-like surface
    def state_dict_views(self):
        """HF-named views into the flat master buffer (writing through them edits the model)."""
        return P.state_dict_from_flat(self.cfg, self.flat_params)

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.state_dict_views().items()}

    def grad_dict(self):
        return {k: v for k, v in P.state_dict_from_flat(self.cfg, self.flat_grads).items()}

    def load_state_dict(self, sd, strict: bool = True):
        import torch
        mine = self.state_dict_views()
        seen = set()
        for k, v in sd.items():
            name = P.canonical_hf_name(k)
            if name not in mine:
                if strict:
                    raise KeyError(f"unexpected key {k}")
                continue
            t = torch.as_tensor(v)
            if tuple(t.shape) != tuple(mine[name].shape):
                raise ValueError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(mine[name].shape)}")
            mine[name].copy_(t.to(self.device, dtype=torch.float32))
            seen.add(name)
        if strict:
            missing = [k for k in mine if k not in seen and k not in P.TIED_ALIASES]
            if missing:
                raise KeyError(f"missing keys: {missing[:4]}...")
        self.sync_weights()
        if self._optimizer is not None:
            self._optimizer.z.copy_(self.flat_params)

    def parameters(self):
        return [self.flat_params]

    def num_parameters(self) -> int:
        return P.num_parameters(self.cfg)

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        return self

    def log(self, name, value, sync_dist: bool = False, **kw):
        """LightningModule.log: the reference logs train_loss / val_loss / val_cer / test_* WITHOUT sync_dist
        (trocr_model.py:331,342,358,370,388), i.e. each rank's own value and rank 0's in the logger -- the default here.
        sync_dist=True is Lightning's mean over ranks: one scalar all-reduce."""
        v = float(value)
        if sync_dist:
            import torch
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                t = torch.tensor([v], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                v = float(t.item()) / dist.get_world_size()
        self.logged.setdefault(name, []).append(v)

    def optimizers(self):
        return self._optimizer

    def __call__(self, pixel_values, labels=None):
        return self.forward(pixel_values, labels)

    # ------------------------------------------------------------------ forward (trocr_model.py:258-321)
    def _check_inputs(self, pixel_values):
        import torch
        c = self.cfg
        if pixel_values.dim() != 4 or pixel_values.shape[1] != c.channels:
            raise ValueError(f"pixel_values must be [B,{c.channels},H,W], got {tuple(pixel_values.shape)}")
        _, _, height, width = pixel_values.shape
        bucket_ok = self.width_buckets is not None and width in self.width_buckets
        if height != c.image_h or (width != c.image_w and not bucket_ok):   # trocr_model.py:83-86
            raise ValueError(f"Input image size ({height}*{width}) doesn't match model ({c.image_h}*{c.image_w}).")
        L.check(L.load().kzv_set_image_width(self._h, int(width)), "set_image_width")
        return pixel_values.to(self.device, dtype=torch.float32).contiguous()

    def forward_loss(self, pixel_values, labels, want_logits: bool = False, seed: int | None = None):
        """Engine call: returns (loss tensor [1] on device, logits or None).  Dropout follows self.training."""
        import torch
        px = self._check_inputs(pixel_values)
        n_max_host = None
        if labels.device.type == "cpu" and labels.dim() == 2:      # dataloader batches: length known without touching the GPU
            n_max_host = int((labels != self.cfg.pad_id).sum(dim=1).max())
            # RoBERTa position ids run to (non-pad decoder inputs) + pad_id and index a [max_pos, H] table
            # (HF modeling_roberta.py:142-155): a longer label raises in the reference ("index out of range in self")
            n_in = int((labels[:, :-1] != self.cfg.pad_id).sum(dim=1).max()) if labels.shape[1] > 1 else 0
            if n_in + self.cfg.pad_id >= self.cfg.max_pos:
                raise IndexError(f"index out of range in self: {n_in} decoder input tokens need position id {n_in + self.cfg.pad_id}, "
                                 f"the position table has {self.cfg.max_pos} rows")
        lab = labels.to(self.device, dtype=torch.int64).contiguous()
        if lab.dim() != 2 or lab.shape[0] != px.shape[0]:
            raise ValueError("labels must be [B, L]")
        B, Lh = lab.shape
        self._bind(B, Lh)
        logits = torch.empty(B, Lh - 1, self.cfg.vocab, dtype=torch.float32, device=self.device) if want_logits else None
        if seed is None:
            self._step_seed += 1
            seed = self._step_seed
        self._keep = (px, lab)   # inputs must outlive the asynchronous kernels (and backward reads labels)
        # trailing all-padding decoder positions change neither loss nor gradients: run the decoder on the prefix that
        # holds characters (one tiny device reduction + sync per step; full length when logits are returned)
        t_act = Lh - 1
        if not want_logits and self.trim_padding:
            key = (lab.data_ptr(), lab._version, tuple(lab.shape))
            if n_max_host is not None:
                n_max = n_max_host
            elif self._len_cache[0] == key:                       # same resident batch again (benchmark loop): no sync
                n_max = self._len_cache[1]
            else:
                n_max = int((lab != self.cfg.pad_id).sum(dim=1).max().item())
                self._len_cache = (key, n_max)
            t_act = max(1, min(Lh - 1, n_max))
        self.last_active_length = t_act
        L.check(L.load().kzv_set_active_length(self._h, t_act), "set_active_length")
        L.check(L.load().kzv_forward_loss(self._h, px.data_ptr(), lab.data_ptr(), self._loss.data_ptr(), L.ptr(logits),
                                          1 if self.training else 0, seed, L.stream_handle()), "kzv_forward_loss")
        return self._loss, logits

    def forward(self, pixel_values, labels=None):
        if labels is not None:
            loss, logits = self.forward_loss(pixel_values, labels, want_logits=True)
            return {"logits": logits, "loss": loss.clone().reshape(())}
        # trocr_model.py:306-316: max_length=128, num_beams=4, early_stopping=True
        return {"generated_ids": self.generate(pixel_values, max_length=128, num_beams=4, early_stopping=True), "logits": None}

    def backward(self) -> None:
        """loss.backward() of the last training-mode forward: fills flat_grads (zeroed first)."""
        lib = L.load()
        st = L.stream_handle()
        L.check(lib.kzv_zero_grads(self._h, st), "zero_grads")
        L.check(lib.kzv_backward(self._h, st), "kzv_backward")

    def generate(self, pixel_values, max_length: int = 128, num_beams: int = 1, early_stopping: bool = True,
                 length_penalty: float = 1.0, use_cache: bool = True):
        """Decode from BOS.  The encoder (and the cross-attention K/V of every decoder layer) runs ONCE.
        use_cache=True (default): each step feeds the newest token through the decoder against cached self-attention keys
        and values (kzv_decode_step); beam steps re-order the cache rows (kzv_decode_reorder).
        use_cache=False: each step is a decoder-only teacher-forced pass over the whole prefix (kzv_decode_logits), whose
        position-t logits only depend on ids[:, :t+1] under the causal mask -- the exact cross-check of the cached path.

        Token selection (greedy for num_beams == 1, else beam search with HF's rules) is kzv/beam.py, pinned on the CPU
        against transformers' own ``generate``; the reference asks for ``num_beams=4, early_stopping=True``
        (trocr_model.py:306-316).  The step logits come from this engine."""
        import torch
        from . import beam as BM
        c = self.cfg
        px = self._check_inputs(pixel_values)
        B = px.shape[0]
        Lh = min(max_length, c.max_pos - c.pad_id - 1)
        lib = L.load()
        was = self.training
        self.training = False
        nb = max(1, int(num_beams))
        BB = B * nb
        # cached decoding runs the encoder and the cross-attention K/V once per IMAGE (kzv_encode_images): beams share them, and no
        # teacher-forced decoder pass is spent on the BOS-only prompt
        share = use_cache
        if nb > 1 and not share:
            px = px.repeat_interleave(nb, dim=0)
        ids0 = torch.full((BB, Lh), c.pad_id, dtype=torch.int64, device=self.device)
        ids0[:, 0] = c.bos_id
        step_logits = torch.empty(BB, c.vocab, dtype=torch.float32, device=self.device)
        if share:
            self._bind(BB, Lh)
            self._keep = (px, ids0)
            L.check(lib.kzv_encode_images(self._h, px.data_ptr(), B, L.stream_handle()), "encode_images")
        else:
            self.forward_loss(px, ids0, want_logits=False, seed=0)          # encoder + cross K/V (and a first decoder pass)
        state = {"valid": torch.zeros(BB, Lh, dtype=torch.uint8, device=self.device)}   # self-attention keys usable (token != pad)
        posids = torch.empty(BB, dtype=torch.int32, device=self.device)
        tok_buf = torch.empty(BB, dtype=torch.int64, device=self.device)
        # the cached step is ~100 small launches: replayed from a hipGraph (kzv_decode_step_graph) unless KZV_DECODE_GRAPH=0;
        # stream capture needs a non-default stream, so the whole decode loop runs on a side stream
        graph = use_cache and os.environ.get("KZV_DECODE_GRAPH", "1") != "0"
        cur = torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(self.device) if graph else cur
        if graph:
            side.wait_stream(cur)

        def step(t, ids):
            if not use_cache:
                ids = ids.contiguous()
                state["ids"] = ids                                        # keep alive until the kernels have run
                L.check(lib.kzv_set_active_length(self._h, t + 1), "set_active_length")   # later positions are not needed
                L.check(lib.kzv_decode_logits(self._h, ids.data_ptr(), t, step_logits.data_ptr(), L.stream_handle()), "decode_logits")
                return step_logits
            valid = state["valid"]
            # newest token, RoBERTa position id (modeling_roberta.py:142-155: cumsum of non-pad tokens + pad_id; prefixes never
            # hold pads) and the key-usable flag of column t, in one launch
            L.check(lib.kzv_decode_prep(ids.data_ptr(), ids.stride(0), t, c.pad_id, BB, tok_buf.data_ptr(), valid.data_ptr(), Lh,
                                        posids.data_ptr(), L.stream_handle()), "decode_prep")
            if graph:
                L.check(lib.kzv_decode_step_graph(self._h, tok_buf.data_ptr(), posids.data_ptr(), valid.data_ptr(), Lh, step_logits.data_ptr(),
                                                  L.stream_handle()), "decode_step_graph")
            else:
                L.check(lib.kzv_decode_step(self._h, tok_buf.data_ptr(), posids.data_ptr(), t, valid.data_ptr(), Lh, step_logits.data_ptr(),
                                            L.stream_handle()), "decode_step")
            return step_logits

        def reorder(rows, n_keys):
            if use_cache:
                rows = rows.contiguous()
                state["valid"].copy_(state["valid"][rows])                # in place: the graph holds this buffer's address
                L.check(lib.kzv_decode_reorder(self._h, rows.data_ptr(), n_keys, L.stream_handle()), "decode_reorder")

        topk = update = None

        try:
            with torch.cuda.stream(side):
                # the hooks allocate and zero their flag / top-k buffers: on the stream whose kernels read them
                if nb > 1 and nb <= 8 and self.device.type == "cuda" and os.environ.get("KZV_BEAM_TOPK", "1") != "0":
                    topk, update = BM.make_device_hooks(B, nb, Lh, c.vocab, c.eos_id, early_stopping, length_penalty, self.device)
                    if os.environ.get("KZV_BEAM_UPDATE", "1") == "0":
                        update = None
                if use_cache:
                    L.check(lib.kzv_set_active_length(self._h, 1), "set_active_length")
                    L.check(lib.kzv_decode_begin(self._h, L.stream_handle()), "decode_begin")
                if nb == 1:
                    gupd = BM.make_greedy_hook(B, c.vocab, c.pad_id, c.eos_id, self.device, Lh) if self.device.type == "cuda" and os.environ.get("KZV_BEAM_TOPK", "1") != "0" else None
                    out = BM.greedy(step, B, Lh, c.pad_id, c.bos_id, c.eos_id, self.device, update=gupd)
                else:
                    out = BM.beam_search(step, reorder, B, nb, Lh, c.vocab, c.pad_id, c.bos_id, c.eos_id, self.device,
                                         early_stopping=early_stopping, length_penalty=length_penalty, topk=topk, update=update)
            if graph:
                cur.wait_stream(side)
                for t_ in (out, step_logits, posids, tok_buf, state["valid"], px):
                    t_.record_stream(cur)
            return out
        finally:
            self.training = was

    # ------------------------------------------------------------------ Lightning-shaped steps (:323-398)
    def training_step(self, batch, batch_idx):
        loss, _ = self.forward_loss(batch["pixel_values"], batch["labels"], want_logits=False)
        self.backward()
        self.last_loss = loss
        return loss

    def validation_step(self, batch, batch_idx):
        was = self.training
        self.training = False
        loss, _ = self.forward_loss(batch["pixel_values"], batch["labels"])
        val = float(loss.item())
        self.log("val_loss", val)
        if batch_idx < 5 and self.tokenizer is not None:
            # trocr_model.py:345-358: self(pixel_values) = beam-4 generation of the whole batch, CER of sample 0
            gen = self(batch["pixel_values"])["generated_ids"]
            pred = self.tokenizer.batch_decode(gen, skip_special_tokens=True)
            tgt = self.tokenizer.batch_decode(batch["labels"], skip_special_tokens=True)
            if len(pred) > 0 and len(tgt) > 0:
                self.log("val_cer", self.calculate_cer(pred[0], tgt[0]))
        self.training = was
        return val

    def test_step(self, batch, batch_idx):
        was = self.training
        self.training = False
        loss, _ = self.forward_loss(batch["pixel_values"], batch["labels"])
        val = float(loss.item())
        self.log("test_loss", val)
        if self.tokenizer is not None:
            gen = self(batch["pixel_values"])["generated_ids"]           # trocr_model.py:373: beam 4, max_length 128
            preds = self.tokenizer.batch_decode(gen, skip_special_tokens=True)
            tgts = self.tokenizer.batch_decode(batch["labels"], skip_special_tokens=True)
            cers = [self.calculate_cer(p, t) for p, t in zip(preds, tgts)]
            self.log("test_cer", sum(cers) / len(cers) if cers else 0.0)
        self.training = was
        return val

    def calculate_cer(self, pred_text: str, target_text: str) -> float:
        """Character Error Rate (trocr_model.py:400-410): Levenshtein / len(target)."""
        if len(target_text) == 0:
            return 1.0 if len(pred_text) > 0 else 0.0
        return _levenshtein(pred_text, target_text) / len(target_text)

    def configure_optimizers(self):
        from .optim import RAdamScheduleFree
        hp = self.hparams
        self._optimizer = RAdamScheduleFree(self, lr=hp.learning_rate, betas=(hp.beta1, hp.beta2), eps=hp.epsilon,
                                            weight_decay=hp.weight_decay)
        return self._optimizer

    # mode hooks (:423-451)
    def on_train_epoch_start(self):
        if self._optimizer is not None:
            self._optimizer.train()

    def on_validation_epoch_start(self):
        if self._optimizer is not None:
            self._optimizer.eval()

    def on_validation_epoch_end(self):
        if self._optimizer is not None:
            self._optimizer.train()

    def on_test_epoch_start(self):
        if self._optimizer is not None:
            self._optimizer.eval()

    def on_predict_epoch_start(self):
        if self._optimizer is not None:
            self._optimizer.eval()

    def decode_predictions(self, pixel_values) -> list[str]:
        self.eval()
        gen = self(pixel_values)["generated_ids"]                         # trocr_model.py:457
        return self.tokenizer.batch_decode(gen, skip_special_tokens=True)


def _levenshtein(a: str, b: str) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]
