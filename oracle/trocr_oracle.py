"""ORACLE (test infrastructure -- never imported by the product path).

CPU restatement, in plain torch ops (fp32 by default, fp64 on request), of the reference's
TrOCR training path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

Parity status: PINNED for forward / loss / gradients by ``tests/golden/*.npz``, which were
generated in the build container by importing the reference ``TrOCRModel``
(/root/reference/src/models/trocr_model.py, with the three import shims of SURVEY.md section
8(c)) -- see ``tools/gen_golden.py``.  UNPINNED: ``radam_schedulefree_step`` (schedulefree==1.4.1
source absent from the container; restated from the published algorithm) and the DDP
mean-of-rank-means semantics (Lightning absent).

Every function cites the reference lines it follows.  "HF:" = transformers
(models/vit/modeling_vit.py, models/roberta/modeling_roberta.py), whose layers the reference
instantiates at src/models/trocr_model.py:147-149 and :231.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _gelu(x):  # erf GELU: ViTConfig.hidden_act="gelu" default; RobertaConfig hidden_act="gelu"
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _heads(x, nh):  # [B,S,H] -> [B,nh,S,d]   (HF modeling_vit.py:212-217)
    B, S, H = x.shape
    return x.view(B, S, nh, H // nh).transpose(1, 2)


def _drop(x, masks, name):
    """nn.Dropout / F.dropout in training mode with an EXPLICIT mask: ``masks[name]`` holds the multiplier of every
    element (0 for dropped, 1/P(keep) for kept), shaped like x.  masks=None or a missing name = dropout off (eval).
    Lets a test replay the exact masks the HIP kernels drew (kzv_debug_dropout_mask) through the reference's arithmetic."""
    if masks is None or name not in masks:
        return x
    return x * masks[name].to(x.dtype).reshape(x.shape)


def _attention(q, k, v, nh, add_mask=None, masks=None, site=None):
    """softmax(q k^T * d^-0.5 + mask) v  -- HF modeling_vit.py:164-189 / modeling_roberta.py:158-183;
    probability dropout at modeling_vit.py:184 / modeling_roberta.py:178 (mask [B, nh, Sq, Sk])."""
    qh, kh, vh = _heads(q, nh), _heads(k, nh), _heads(v, nh)
    s = torch.matmul(qh, kh.transpose(2, 3)) * (qh.shape[-1] ** -0.5)
    if add_mask is not None:
        s = s + add_mask
    p = _drop(torch.softmax(s, dim=-1), masks, site)
    o = torch.matmul(p, vh)  # [B,nh,S,d]
    B, _, S, _ = o.shape
    return o.transpose(1, 2).reshape(B, S, -1)


# ---- fp8 weight path (BASELINE.json configs[4]; NOT in the reference, which has no fp8: this restates the build's own
# quantisation recipe -- include/kzv.h "fp8 weight path" -- so that the HIP kernels can be checked against exact arithmetic) ----
FP8_MAX = 448.0     # largest finite OCP e4m3fn value


def quant_e4m3(x):
    """Nearest OCP e4m3fn value of every element (4 exponent bits, bias 7; 3 mantissa bits; subnormals; round to nearest
    even; saturating at +-448), returned as float32.  Restated from the format: spacing 2^(e-3) in binade e >= -6, 2^-9 below."""
    x = x.detach().to(torch.float32)
    _, e = torch.frexp(x)                                  # x = m * 2^e, m in [0.5, 1)  ->  binade exponent = e - 1
    step = torch.ldexp(torch.ones_like(x), torch.clamp(e - 1, min=-6) - 3)
    return torch.clamp(torch.round(x / step) * step, -FP8_MAX, FP8_MAX)    # torch.round = half to even


def quant_rows_e4m3(x):
    """Per-row quantisation (kzv_quant_rows_fp8; LayerNorm's fp8 output): q = e4m3(x * (448 / amax_row)), scale = amax_row / 448
    (an all-zero row keeps scale 1).  fp32 arithmetic like the kernels.  Returns (q as float32, scale [..., 1])."""
    x32 = x.detach().to(torch.float32)
    amax = x32.abs().amax(dim=-1, keepdim=True)
    nz = amax > 0
    qs = torch.where(nz, torch.tensor(FP8_MAX, dtype=torch.float32) / amax, torch.ones_like(amax))
    scale = torch.where(nz, amax / torch.tensor(FP8_MAX, dtype=torch.float32), torch.ones_like(amax))
    return quant_e4m3(x32 * qs), scale


def next_act_qscale(amax: float, prev: float = 1.0) -> float:
    """Delayed scaling of the per-tensor sites (kzv_fp8_roll): a power of two with one binade of headroom below 448 / amax;
    unchanged while nothing has been seen."""
    return prev if amax <= 0 else 2.0 ** (math.floor(math.log2(FP8_MAX / amax)) - 1)


F8_BOUND = 1.4125          # csrc/layernorm.hip KZV_F8_BOUND: max gelu' (1.13) x 1.25


class _LinearFp8Grad(torch.autograd.Function):
    """fp8 mode 2: the value of _linear_fp8 forward, and an INPUT gradient computed on e4m3 operands like the build's d-fc2 /
    d-fc1 GEMMs (weight and bias gradients stay those of the unquantised linear: the build's weight-gradient GEMMs read bf16).
    kind "fc2": the arriving gradient rows are quantised by their own amax; the row multiplier of the NEXT gradient tensor
    (st["rq"]) comes from the bound ||g row|| * max ||W column|| * F8_BOUND.  kind "fc1": rows quantised with st["rq"]."""

    @staticmethod
    def forward(ctx, h, W, b, val, kind, st):
        ctx.save_for_backward(h, W)
        ctx.kind, ctx.st = kind, st
        return val.detach().clone()

    @staticmethod
    def backward(ctx, g):
        h, W = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        h2 = h.reshape(-1, h.shape[-1])
        wt = W.detach().to(torch.bfloat16).to(torch.float32).t().contiguous()      # rows = W columns, from the bf16 copy
        wq, ws = quant_rows_e4m3(wt)
        g32 = g2.detach().to(torch.float32)
        if ctx.kind == "fc2":
            gq, gs = quant_rows_e4m3(g32)
            norm = torch.sqrt((g32 * g32).sum(dim=-1, keepdim=True))
            cmax = torch.sqrt((wt * wt).sum(dim=-1)).max()
            bound = F8_BOUND * norm * cmax
            ctx.st["rq"] = torch.where(bound > 0, torch.tensor(FP8_MAX, dtype=torch.float32) / bound, torch.ones_like(bound))
        else:
            rq = ctx.st["rq"]
            gq, gs = quant_e4m3(g32 * rq), 1.0 / rq
        dt = g.dtype
        gh = torch.matmul(gq.to(dt), wq.to(dt).t()) * gs.to(dt) * ws.to(dt).reshape(1, -1)
        return gh.reshape(h.shape), torch.matmul(g2.t(), h2), g2.sum(dim=0), None, None, None


def _linear_fp8(h, W, b, act_qscale=None, dgrad=None):
    """F.linear on e4m3 operands: W quantised per output row; h per token row, or per tensor with multiplier ``act_qscale``.
    Forward value = (hq Wq^T) * scales + b in h's dtype; gradients are those of the unquantised linear (the build's backward
    reads the bf16 operands: straight-through) unless ``dgrad`` = (kind, state) asks for the e4m3 input gradient of mode 2."""
    wq, ws = quant_rows_e4m3(W)
    if act_qscale is None:
        hq, hs = quant_rows_e4m3(h)
    else:
        hq = quant_e4m3(h.detach().to(torch.float32) * float(act_qscale))
        hs = torch.full(h.shape[:-1] + (1,), 1.0 / float(act_qscale), dtype=torch.float32)
    dt = h.dtype
    val = torch.matmul(hq.to(dt), wq.to(dt).t()) * hs.to(dt) * ws.to(dt).reshape(1, -1) + b.detach()
    if dgrad is not None:
        return _LinearFp8Grad.apply(h, W, b, val, dgrad[0], dgrad[1])
    plain = F.linear(h, W, b)
    return plain + (val - plain).detach()


def patch_embed(cfg, sd, pixel_values):
    """CustomPatchEmbeddings.forward -- src/models/trocr_model.py:79-92 (Conv2d k=s=patch; h-major patches)."""
    B, C, H, W = pixel_values.shape
    if H != cfg.image_h or W != cfg.image_w:
        raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({cfg.image_h}*{cfg.image_w}).")
    x = F.conv2d(pixel_values, sd["encoder.patch_embeddings.projection.weight"],
                 sd["encoder.patch_embeddings.projection.bias"], stride=(cfg.patch_h, cfg.patch_w))
    return x.flatten(2).transpose(1, 2)


def encoder_forward(cfg, sd, pixel_values, stages=None, masks=None, fp8=None):
    """ViTEncoder.forward -- src/models/trocr_model.py:169-202; layers = HF ViTLayer (modeling_vit.py:257-286).
    Dropout sites (training): embeddings :190 ("enc_emb"), attention probabilities ("enc{i}_attn"), ViTLayer.dropout after
    the attention block (modeling_vit.py:276, "enc{i}_o") and after the MLP (:283, "enc{i}_mlp").
    ``fp8`` (the build's fp8 weight path, not the reference): {"act_qscale": [per-layer multiplier of the GELU output]} runs
    query/key/value, intermediate.dense and output.dense through _linear_fp8; stages["enc{i}_act_amax"] records max |GELU|."""
    B = pixel_values.shape[0]
    x = patch_embed(cfg, sd, pixel_values)
    if stages is not None:
        stages["patch_embed"] = x
    cls = sd["encoder.cls_token"].expand(B, -1, -1)
    x = _drop(torch.cat((cls, x), dim=1) + sd["encoder.position_embeddings"], masks, "enc_emb")   # :183-190
    if stages is not None:
        stages["enc_embed"] = x
    nh = cfg.enc_heads
    for i in range(cfg.enc_layers):
        p = f"encoder.encoder.layer.{i}."
        h = _ln(x, sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"], cfg.ln_eps)
        lin = F.linear if fp8 is None else _linear_fp8
        q = lin(h, sd[p + "attention.attention.query.weight"], sd[p + "attention.attention.query.bias"])
        k = lin(h, sd[p + "attention.attention.key.weight"], sd[p + "attention.attention.key.bias"])
        v = lin(h, sd[p + "attention.attention.value.weight"], sd[p + "attention.attention.value.bias"])
        a = _attention(q, k, v, nh, None, masks, f"enc{i}_attn")
        x = x + _drop(F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]), masks, f"enc{i}_o")
        h = _ln(x, sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"], cfg.ln_eps)
        g8 = {} if (fp8 is not None and fp8.get("dgrad")) else None              # mode 2: the MLP's input gradients on e4m3 operands
        if g8 is None:
            h = _gelu(lin(h, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        else:
            h = _gelu(_linear_fp8(h, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"], None, ("fc1", g8)))
        if fp8 is None:
            y = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        else:
            if stages is not None:
                stages[f"enc{i}_act_amax"] = h.detach().abs().max()
            y = _linear_fp8(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"], fp8["act_qscale"][i],
                            ("fc2", g8) if g8 is not None else None)
        x = x + _drop(y, masks, f"enc{i}_mlp")
        if stages is not None:
            stages[f"enc_layer{i}"] = x
    x = _ln(x, sd["encoder.layernorm.weight"], sd["encoder.layernorm.bias"], cfg.ln_eps)  # :197
    x = x[:, 1:, :]                                                                       # :200
    if stages is not None:
        stages["enc_out"] = x
    return x


def position_ids(ids, pad_id):
    """create_position_ids_from_input_ids -- HF modeling_roberta.py:142-155."""
    m = (ids != pad_id).to(torch.int64)
    return torch.cumsum(m, dim=1) * m + pad_id


def decoder_forward(cfg, sd, input_ids, enc, stages=None, masks=None):
    """RobertaForCausalLM(is_decoder, add_cross_attention) teacher-forced forward.

    embeddings: HF modeling_roberta.py:75-122; layer (post-LN): :421-464 with self-attn :186-250,
    cross-attn :253-326, output blocks :329-340 / :372-398; masks :645-680 (causal AND key != pad,
    no encoder mask); head :877-893.  Attention mask built from ids as in
    src/models/trocr_model.py:278.
    Dropout sites (training): embeddings :120 ("dec_emb"); per layer the self / cross attention probabilities :178
    ("dec{i}_sa", "dec{i}_ca"), RobertaSelfOutput.dropout :338 of both blocks ("dec{i}_sa_o", "dec{i}_ca_o") and
    RobertaOutput.dropout :396 ("dec{i}_ffn").
    """
    r = "decoder.roberta."
    B, T = input_ids.shape
    pos = position_ids(input_ids, cfg.pad_id)
    x = (sd[r + "embeddings.word_embeddings.weight"][input_ids]
         + sd[r + "embeddings.token_type_embeddings.weight"][0]
         + sd[r + "embeddings.position_embeddings.weight"][pos])
    x = _drop(_ln(x, sd[r + "embeddings.LayerNorm.weight"], sd[r + "embeddings.LayerNorm.bias"], cfg.ln_eps), masks, "dec_emb")
    if stages is not None:
        stages["dec_embed"] = x
    neg = torch.finfo(x.dtype).min
    causal = torch.ones(T, T, dtype=torch.bool).tril()
    keep = causal[None, None, :, :] & (input_ids != cfg.pad_id)[:, None, None, :]
    mask = torch.zeros(B, 1, T, T, dtype=x.dtype).masked_fill(~keep, neg)
    nh = cfg.dec_heads
    for i in range(cfg.dec_layers):
        p = r + f"encoder.layer.{i}."
        q = F.linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
        k = F.linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
        v = F.linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
        a = _attention(q, k, v, nh, mask, masks, f"dec{i}_sa")
        a = _drop(F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"]), masks, f"dec{i}_sa_o")
        x = _ln(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        q = F.linear(x, sd[p + "crossattention.self.query.weight"], sd[p + "crossattention.self.query.bias"])
        k = F.linear(enc, sd[p + "crossattention.self.key.weight"], sd[p + "crossattention.self.key.bias"])
        v = F.linear(enc, sd[p + "crossattention.self.value.weight"], sd[p + "crossattention.self.value.bias"])
        a = _attention(q, k, v, nh, None, masks, f"dec{i}_ca")
        a = _drop(F.linear(a, sd[p + "crossattention.output.dense.weight"], sd[p + "crossattention.output.dense.bias"]), masks, f"dec{i}_ca_o")
        x = _ln(a + x, sd[p + "crossattention.output.LayerNorm.weight"], sd[p + "crossattention.output.LayerNorm.bias"], cfg.ln_eps)
        h = _gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        h = _drop(F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"]), masks, f"dec{i}_ffn")
        x = _ln(h + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], cfg.ln_eps)
        if stages is not None:
            stages[f"dec_layer{i}"] = x
    h = _gelu(F.linear(x, sd["decoder.lm_head.dense.weight"], sd["decoder.lm_head.dense.bias"]))
    h = _ln(h, sd["decoder.lm_head.layer_norm.weight"], sd["decoder.lm_head.layer_norm.bias"], cfg.ln_eps)
    # tied: lm_head.decoder.weight IS word_embeddings.weight (modeling_roberta.py:684-687)
    return F.linear(h, sd[r + "embeddings.word_embeddings.weight"], sd["decoder.lm_head.bias"])


def forward(cfg, sd, pixel_values, labels, stages=None, masks=None, fp8=None):
    """TrOCRModel.forward, training branch -- src/models/trocr_model.py:258-297.  Returns (logits, loss).
    ``masks`` (optional): explicit dropout multipliers per site, see _drop; None = eval mode."""
    enc = encoder_forward(cfg, sd, pixel_values, stages, masks, fp8)
    if cfg.has_proj:
        enc = F.linear(enc, sd["encoder_decoder_proj.weight"], sd["encoder_decoder_proj.bias"])  # :269
    if stages is not None:
        stages["proj_out"] = enc
    ids = labels[:, :-1].contiguous()       # :274
    tgt = labels[:, 1:].contiguous()        # :275
    logits = decoder_forward(cfg, sd, ids, enc, stages, masks)
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), tgt.reshape(-1), ignore_index=cfg.pad_id)  # :256,:292
    return logits, loss


def leaf_state_dict(sd_np, dtype=torch.float32, requires_grad=True):
    """numpy HF-named dict -> torch leaves; tied aliases dropped (they share storage)."""
    out = {}
    for k, v in sd_np.items():
        if k.startswith("decoder.lm_head.decoder."):
            continue
        t = torch.tensor(v, dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def forward_backward(cfg, sd_np, pixel_values, labels, dtype=torch.float32, want_stages=False, masks=None, fp8=None):
    """One teacher-forced step; returns dict(logits, loss, grads{hf_name: ndarray}, stages)."""
    sd = leaf_state_dict(sd_np, dtype)
    stages = {} if want_stages else None
    if masks is not None:
        masks = {k: torch.as_tensor(v) for k, v in masks.items()}
    logits, loss = forward(cfg, sd, torch.as_tensor(pixel_values).to(dtype), torch.as_tensor(labels), stages, masks, fp8)
    loss.backward()
    grads = {k: (v.grad.detach().numpy() if v.grad is not None else None) for k, v in sd.items()}
    return {"logits": logits.detach().numpy(), "loss": float(loss.detach()), "grads": grads,
            "stages": {k: v.detach().numpy() for k, v in (stages or {}).items()}}


# ---- runtime policy restated: clip + optimizer --------------------------------
def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ as Lightning applies it (scripts/train_trocr.py:175):
    total L2 norm over all grads; scale = max_norm / (norm + 1e-6), clamped to 1."""
    total = math.sqrt(sum(float((g.astype("float64") ** 2).sum()) for g in grads))
    coef = min(1.0, max_norm / (total + 1e-6))
    return total, coef


class RAdamScheduleFreeState:
    """Host-side scalars of schedulefree.RAdamScheduleFree (parity UNPINNED, see module header).

    Restated from the Schedule-Free paper (Defazio et al. 2024) + the RAdam rectification:
      y (train-mode params) , z (base iterate), v (second moment);  x = eval-mode params.
    Reference call site: src/models/trocr_model.py:412-421 (lr 1e-4, betas (0.9,0.999), eps 1e-8, wd 0),
    mode hooks :423-451.
    """

    def __init__(self, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                 r=0.0, weight_lr_power=2.0, silent_sgd_phase=True):
        self.lr, self.beta1, self.beta2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.r, self.wlp, self.silent = r, weight_lr_power, silent_sgd_phase
        self.k = 0
        self.lr_max = -1.0
        self.weight_sum = 0.0

    def next_scalars(self):
        """Advance one step; returns (lr_t, ckp1, bias_correction2, use_adaptive)."""
        step = self.k + 1
        beta2_t = self.beta2 ** step
        bc2 = 1.0 - beta2_t
        rho_inf = 2.0 / (1.0 - self.beta2) - 1.0
        rho_t = rho_inf - 2.0 * step * beta2_t / bc2
        if rho_t > 4.0:
            rect = math.sqrt((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t))
        else:
            rect = float(not self.silent)
        lr = self.lr * rect
        self.lr_max = max(lr, self.lr_max)
        weight = (step ** self.r) * (self.lr_max ** self.wlp)
        self.weight_sum += weight
        ckp1 = weight / self.weight_sum if self.weight_sum != 0 else 0.0
        self.k = step
        return lr, ckp1, bc2, rho_t > 4.0


def radam_schedulefree_step(state, y, z, v, g):
    """In-place numpy/torch update of (y, z, v) given grad g (already clipped)."""
    lr, ckp1, bc2, adaptive = state.next_scalars()
    v *= state.beta2
    v += (1.0 - state.beta2) * g * g
    if adaptive:
        gn = g / ((v / bc2) ** 0.5 + state.eps)
    else:
        gn = g.copy() if hasattr(g, "copy") else g.clone()
    if state.wd:
        gn = gn + state.wd * y
    y += ckp1 * (z - y)                                  # y.lerp_(z, ckp1)
    y += lr * (state.beta1 * (1.0 - ckp1) - 1.0) * gn    # y.add_(gn, alpha=adaptive_y_lr)
    z -= lr * gn
    return lr, ckp1


def to_eval(y, z, beta1):
    """optimizer.eval(): p <- lerp(p, z, 1 - 1/beta1)   (y -> x)."""
    return y + (1.0 - 1.0 / beta1) * (z - y)


def to_train(x, z, beta1):
    """optimizer.train(): p <- lerp(p, z, 1 - beta1)    (x -> y)."""
    return x + (1.0 - beta1) * (z - x)


# ---- CER: src/models/trocr_model.py:400-410 (editdistance.eval == Levenshtein) ---
def levenshtein(a, b) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def calculate_cer(pred_text: str, target_text: str) -> float:
    if len(target_text) == 0:
        return 1.0 if len(pred_text) > 0 else 0.0
    return levenshtein(pred_text, target_text) / len(target_text)


# ---- greedy decoding by step-wise forward -------------------------
def greedy_stepwise(cfg, sd, pixel_values, max_length):
    """Greedy decoding from BOS through the TRAINING-branch forward, one position per call: token t+1 = argmax of the
    teacher-forced logits at position t over the prefix so far (under the causal AND key-not-pad mask position t only sees
    ids[:, :t+1]; src/models/trocr_model.py:274-287).  This is the golden for the product's KV-cached ``generate`` --
    HF ``generate`` under transformers 5.x is not usable as one.  Finished rows (EOS emitted) are padded.
    Returns (ids [B, max_length] int64, top-2 logit gap of every decision [B, steps])."""
    px = torch.as_tensor(pixel_values)
    B = px.shape[0]
    with torch.no_grad():
        enc = encoder_forward(cfg, sd, px)
        if cfg.has_proj:
            enc = F.linear(enc, sd["encoder_decoder_proj.weight"], sd["encoder_decoder_proj.bias"])
        ids = torch.full((B, max_length), cfg.pad_id, dtype=torch.int64)
        ids[:, 0] = cfg.bos_id
        done = torch.zeros(B, dtype=torch.bool)
        gaps = []
        for t in range(max_length - 1):
            lg = decoder_forward(cfg, sd, ids[:, :-1], enc)[:, t]
            top2 = lg.topk(2, dim=-1).values
            gaps.append((top2[:, 0] - top2[:, 1]).masked_fill(done, float("inf")))
            nxt = lg.argmax(-1)
            nxt = torch.where(done, torch.full_like(nxt, cfg.pad_id), nxt)
            ids[:, t + 1] = nxt
            done |= nxt == cfg.eos_id
            if bool(done.all()):
                break
    return ids.numpy(), torch.stack(gaps, 1).numpy()


def stepwise_logits(cfg, sd, pixel_values, repeat=1):
    """Closure ``step(t, ids) -> logits [B*repeat, V]`` of position t given ids[:, :t+1], through the training-branch
    decoder over the prefix (encoder run once; rows repeated ``repeat`` times for beam search).  Drives kzv/beam.py's
    token selection with ORACLE logits, which is the golden of the engine's ``forward(labels=None)`` (beam 4)."""
    with torch.no_grad():
        enc = encoder_forward(cfg, sd, torch.as_tensor(pixel_values))
        if cfg.has_proj:
            enc = F.linear(enc, sd["encoder_decoder_proj.weight"], sd["encoder_decoder_proj.bias"])
        enc = enc.repeat_interleave(repeat, dim=0)

    def step(t, ids):
        with torch.no_grad():
            return decoder_forward(cfg, sd, ids[:, :t + 1].contiguous(), enc)[:, t]
    return step
