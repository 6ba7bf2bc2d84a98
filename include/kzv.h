/* libkzv -- C ABI of the MI355X-native TrOCR line-OCR training engine.
 *
 * The reference (Kotomiya07/kuzushiji-vision) has no FFI for this path: every FLOP runs inside
 * PyTorch / HF transformers behind the Python class `TrOCRModel`
 * (src/models/trocr_model.py:205-460).  This header is the boundary a maintainer would bind from
 * that class (ctypes stub: INTEGRATION.md).  Each entry point names the reference call it replaces.
 *
 * Conventions: every function returns 0 on success, a negative KZV_E_* code on error and records
 * a message readable through kzv_last_error().  All pointers named d_* are DEVICE pointers owned
 * by the caller (torch tensors in the Python host): parameters, gradients, optimizer state and one workspace
 * sized by kzv_workspace_bytes.  The library itself allocates only small or grow-only scratch on the current device,
 * kept for the life of the process (one device per process): a 4 KiB page of zeros (LDS-DMA loads of out-of-range
 * tile rows are pointed at it), the LayerNorm gamma/beta partial rows (64 regions of 512 KiB: the folds of a backward pass are one
 * launch), the gemm_tn256 partial-tile workspace (6 regions of <= 66 MB: the folds of an encoder layer's four weight gradients are
 * one launch), the embedding backward's partial sums (~5 MB at batch 256) and, per model handle, the generation path's KV cache,
 * beam row tables, decode-layout copy of the cross-attention K/V and the fragment-ordered copy of the decoder weights (9.6 MB;
 * allocated at the first kzv_decode_step / kzv_decode_begin or training forward that needs them, freed by kzv_model_destroy).
 * Other global state: the last-error string, the profiling slots and the CU reserve.  `stream` is a hipStream_t passed as void*.
 * One model handle per process/GPU; a handle is not re-entrant.
 */
#ifndef KZV_H
#define KZV_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZV_OK 0
#define KZV_E_ARG (-1)     /* bad argument / shape (reference: ValueError, trocr_model.py:83-86) */
#define KZV_E_HIP (-2)     /* HIP runtime error */
#define KZV_E_STATE (-3)   /* call order violated (e.g. backward before forward) */

const char* kzv_last_error(void);
int kzv_version(void);

/* ------------------------------------------------------------------ geometry */
/* Mirrors encoder_config (trocr_model.py:234-244) + the decoder's RobertaConfig
 * (train_language_model_scratch.py:406-428). */
typedef struct kzv_config {
    int32_t image_h, image_w, patch_h, patch_w, channels;
    int32_t enc_hidden, enc_layers, enc_heads, enc_ffn;
    int32_t dec_hidden, dec_layers, dec_heads, dec_ffn;
    int32_t vocab, max_pos, type_vocab, pad_id;
    float enc_hidden_dropout, enc_attn_dropout, dec_hidden_dropout, dec_attn_dropout;
    float ln_eps;
} kzv_config;

typedef struct kzv_model kzv_model;

/* TrOCRModel.__init__ (trocr_model.py:208-256): validates geometry, builds the parameter table. */
int kzv_model_create(const kzv_config* cfg, kzv_model** out);
int kzv_model_destroy(kzv_model* m);

/* Parameter table = the reference's state_dict, fused/flattened (kzv/params.py documents the
 * mapping to HF names).  Offsets are in fp32 ELEMENTS into the flat master buffer. */
int kzv_param_count(const kzv_model* m);
int kzv_param_info(const kzv_model* m, int i, const char** name, int64_t* offset, int64_t* rows, int64_t* cols);
int64_t kzv_param_total(const kzv_model* m);                 /* padded element count of the flat buffer */

/* Bytes of caller-provided device memory needed for a per-GPU batch of B crops with labels [B, L]. */
int64_t kzv_workspace_bytes(const kzv_model* m, int batch, int label_len);

/* Bind caller-owned device buffers: fp32 master params, fp32 grads (same layout), workspace.
 * (the reference's nn.Parameter storage + autograd .grad + activation memory) */
int kzv_model_bind(kzv_model* m, float* d_params, float* d_grads, void* d_workspace, int64_t workspace_bytes,
                   int batch, int label_len);

/* Refresh the bf16 compute copies (W and W^T) from the fp32 masters; call after the masters change
 * (load_state_dict / optimizer step).  Stands in for autocast's per-op weight cast. */
int kzv_model_sync_weights(kzv_model* m, void* stream);

/* TrOCRModel.forward(pixel_values, labels) training branch (trocr_model.py:258-297).
 *   d_pixel_values fp32 [B,C,H,W]; d_labels int64 [B,L]; d_loss fp32[1] (mean CE over non-pad targets);
 *   d_logits fp32 [B,L-1,V] or NULL.  train!=0 applies dropout (keyed by `seed`) and keeps what
 *   backward needs.  Image-size mismatch is the caller's ValueError (host checks shapes). */
int kzv_forward_loss(kzv_model* m, const float* d_pixel_values, const int64_t* d_labels,
                     float* d_loss, float* d_logits, int train, uint64_t seed, void* stream);

/* Width buckets (BASELINE.json configs[4]; beyond the reference, whose model instance has ONE image size,
 * src/models/trocr_model.py:83-86,113-115).  The handle is created for the WIDEST crop (cfg.image_w, which sizes the
 * position table and the workspace); a batch of narrower crops [B, C, H, width] runs with (H/ph) * (width/pw) patches:
 * patch (h, w) takes the position row of (h, w) in the widest grid -- for the sin/cos table (trocr_model.py:154-167) that
 * IS the closed-form table of the narrow grid, because an entry depends on (w, h) only.  width must be a multiple of
 * the patch width, <= cfg.image_w.  Applies to the following kzv_forward_loss (default after create: cfg.image_w). */
int kzv_set_image_width(kzv_model* m, int width);

/* Position-id overflow check of the LAST kzv_forward_loss / kzv_decode_logits on this handle: RoBERTa's position ids
 * (count of non-pad decoder inputs + pad_id, HF modeling_roberta.py:142-155) must index the [max_pos, H] table; the reference
 * raises "index out of range" for longer labels, the kernels clamp and raise a device flag.  Synchronises `stream`, reads the
 * flag (cleared by every forward) and returns KZV_E_ARG when it was set.  Optional: the Python host validates labels it can
 * see on the CPU before the call. */
int kzv_check_positions(kzv_model* m, void* stream);

/* Active decoder length.  Labels are [B, L] rows padded to L; when every sample's characters end before column
 * t_active + 1, decoder positions >= t_active hold only padding: they are masked as attention keys, their targets are
 * ignored by the loss and nothing reads their outputs, so the engine may run the decoder on the packed [B, t_active]
 * prefix -- loss and gradients are unchanged (the reference computes and discards those positions).  Default after
 * kzv_model_bind: L - 1 (everything).  kzv_forward_loss with d_logits != NULL requires the full length. */
int kzv_set_active_length(kzv_model* m, int t_active);

/* Decoder-only teacher-forced pass over `d_labels` [B,L] reusing the encoder states (and cross-attention K/V) of the
 * last kzv_forward_loss on this handle; writes the logits of position `pos` of every sample, fp32 [B,V].  Building
 * block of generation (TrOCRModel.forward inference branch, trocr_model.py:298-321): under the causal mask position
 * pos only depends on ids[:, :pos+1].  Invalidates the saved activations (no backward afterwards). */
int kzv_decode_logits(kzv_model* m, const int64_t* d_labels, int pos, float* d_logits, void* stream);

/* Encoder only (ViTEncoder.forward + encoder_decoder_proj + the cross-attention K/V of every decoder layer) for n_images
 * crops [n_images, C, H, W], n_images dividing the bound batch: the bound batch counts DECODER rows, and beam search runs
 * batch / n_images beams per image that all attend to that image's K/V (HF expands encoder_hidden_states per beam,
 * trocr_model.py:306-316 -> generate; the values are identical, so they are computed and stored once).  Only
 * kzv_decode_step* may follow (kzv_decode_logits / kzv_forward_loss need one image per row). */
int kzv_encode_images(kzv_model* m, const float* d_pixel_values, int n_images, void* stream);

/* KV-cached generation step (what `decoder.generate(use_cache=True)` does per token, trocr_model.py:306-316): feeds ONE
 * token per sequence -- d_tokens [B] at decoder index t, with RoBERTa position ids d_posids [B] (t + 1 + pad_id for a live
 * sequence, pad_id for padding) -- through the decoder against the self-attention keys / values cached by steps 0..t-1
 * (this step's are appended) and the cross-attention K/V of the last kzv_forward_loss; d_valid [B, ld_valid] marks the
 * usable self-attention keys (key j usable iff token j is not padding; column t must already be set).  Writes logits
 * [B, V] for the next token.  Steps must be issued in order t = 0, 1, ...; the cache is sized [B, L-1] at the first call.
 * kzv_decode_reorder re-parents the sequences after a beam step: row b continues from former row d_rows[b] (first `len` key
 * positions).  No cache row moves: a row table tells the next steps' attention which ancestor's row holds each cached key. */
int kzv_decode_step(kzv_model* m, const int64_t* d_tokens, const int32_t* d_posids, int t, const uint8_t* d_valid, int64_t ld_valid,
                    float* d_logits, void* stream);
int kzv_decode_reorder(kzv_model* m, const int64_t* d_rows, int len, void* stream);
/* One beam-search step's ranking on the device (what transformers' GenerationMixin._get_top_k_continuations computes with
 * log_softmax + topk): for each of `batch` images, log_softmax of each of its `num_beams` (<= 8) rows of d_logits [batch *
 * num_beams, ld] + that beam's d_beam_scores entry, then the k (<= 16) best of the num_beams * vocab continuations, best first;
 * equal scores rank by the smaller flat index.  d_top_scores [batch, k] fp32, d_top_index [batch, k] int64 = beam * vocab + token. */
int kzv_beam_topk(const float* d_logits, int64_t ld, const float* d_beam_scores, int batch, int num_beams, int vocab, int k,
                  float* d_top_scores, int64_t* d_top_index, void* stream);
/* The inputs of decoder step t from the token table d_ids [batch, ld_ids]: d_tokens [batch] = column t, d_posids [batch] = its
 * RoBERTa position id (t + 1 + pad_id, or pad_id for padding), d_valid[:, t] = token != pad. */
int kzv_decode_prep(const int64_t* d_ids, int64_t ld_ids, int t, int pad_id, int batch, int64_t* d_tokens, uint8_t* d_valid, int64_t ld_valid,
                    int32_t* d_posids, void* stream);
/* Greedy token selection (num_beams = 1): d_ids[:, t + 1] = argmax of the row (first maximum), or pad_id once the sequence has
 * emitted eos_id (d_done [batch] is read and updated); d_running[t] += sequences still running after this step (the caller
 * zeroes d_running [>= t + 1 ints] once per generation and may read it every few steps: steps past the end only write padding). */
int kzv_greedy_update(const float* d_logits, int64_t ld, int vocab, int64_t* d_ids, int64_t ld_ids, int t, uint8_t* d_done, int batch,
                      int pad_id, int eos_id, int32_t* d_running, void* stream);
/* One beam-search step's bookkeeping on the device -- what transformers' GenerationMixin._beam_search does between two decoder
 * steps (running beams of the next step, finished list, early-stop heuristic; kzv/beam.py states it in torch ops and is pinned
 * against HF on the CPU).  State arrays live in caller memory for the whole generation; token rows are double-buffered (the
 * caller swaps run_seq_in/out and fin_seq_in/out after every call).  `cur` = index of the token being chosen (1 = first after
 * BOS).  Outputs: d_rows [batch * num_beams] = the former flat row every running beam continues from (for kzv_decode_reorder),
 * d_flags [5]: [0..2] = {images whose open beams may still improve, images whose finished list is not full, images with a
 * continuation that did not stop}: the loop goes on iff flags[0] > 0 && !(early_stopping && flags[1] == 0) && flags[2] > 0.  That
 * test is also taken on the device: d_flags[3] becomes 1 when the search has ended and d_flags[4] holds the number of updates
 * applied; calls issued after the end change nothing (the caller may read d_flags only every few steps; the valid token rows are
 * then the buffers written by update number d_flags[4]).  The call with cur == 1 resets d_flags[3..4]. */
typedef struct kzv_beam_state {
    int32_t batch, num_beams, max_len, vocab, eos_id;
    const int64_t* run_seq_in; int64_t* run_seq_out;     /* [batch, num_beams, max_len] */
    const int64_t* fin_seq_in; int64_t* fin_seq_out;     /* [batch, num_beams, max_len] */
    float* run_scores; float* fin_scores;                /* [batch, num_beams] */
    uint8_t* fin_done; int64_t* fin_len;                 /* [batch, num_beams] */
    uint8_t* unsatisfied;                                /* [batch] */
} kzv_beam_state;
int kzv_beam_update(const kzv_beam_state* st, const float* d_top_scores, const int64_t* d_top_index, int cur, int early_stopping,
                    float length_penalty, int64_t* d_rows, int32_t* d_flags, void* stream);
/* The same step replayed from a hipGraph (the eager step is ~100 small launches and host-bound at ~0.85 ms per token):
 * the step index lives in device memory -- kzv_decode_begin resets it to 0, every kzv_decode_step_graph runs step t and
 * leaves t + 1 -- so one instantiated graph serves every step of a generation.  The graph is captured on the first call
 * and re-captured whenever a buffer pointer (or the cache copy in use after kzv_decode_reorder) differs from the captured
 * one: keep d_tokens / d_posids / d_valid / d_logits in fixed buffers.  `stream` must not be the default stream.  The
 * active image width / weights must not change between kzv_decode_begin and the last step. */
int kzv_decode_begin(kzv_model* m, void* stream);
/* How a step runs (both entry points).  1 (default; KZV_DECODE_ONE_LAUNCH): for the reference decoder's geometry -- hidden 256, 4
 * heads, FFN 768, <= 128 cached and <= 160 patch keys, 1 / 2 / 4 rows per image -- embeddings, all layers and the LM head's dense
 * layer are ONE launch (csrc/decode_fused.hip: a workgroup per image owns its beams through every layer), followed by the
 * vocabulary GEMM.  0, or any other geometry: one launch per operation (~50 per token).  Same results up to fp32 summation order.
 * -1 returns to the environment's default.  The fragment-ordered weight copies the one-launch step reads (9.6 MB) are refreshed by
 * the first step after kzv_model_sync_weights. */
int kzv_set_decode_one_launch(int on);
/* Training / evaluation forward: the linear chains of a decoder layer -- [output projection + dropout + residual -> LayerNorm ->
 * cross query] and [output projection -> LayerNorm -> fc1 + GELU -> fc2 -> LayerNorm -> the next layer's QKV] -- as TWO launches
 * per layer (csrc/decoder_chain.hip; hidden 256, 4 heads, FFN 768) instead of nine.  0: one launch per operation; 1: the forward
 * chains (+ the decoder's 256 x 256 input-gradient GEMMs on the row-panel kernel); 2 (default; KZV_DEC_CHAIN): also the BACKWARD's three
 * row-local segments per layer (input-gradient GEMM + residual gradient -> LayerNorm backward -> dropout mask -> input-gradient GEMM) as
 * one launch each, 13 -> 6 launches per layer; -1: the environment's default.  Same tensors, same rounding points, same dropout bits:
 * logits bit-identical across the three modes, gradients equal up to float-atomic order. */
int kzv_set_dec_chain(int on);
/* Training / validation forward without returned logits: lm_head.decoder (HF modeling_roberta.py:888-893, tied weight) + log-softmax +
 * NLL (src/models/trocr_model.py:256,292) + the bf16 gradient of the logits as ONE launch; the [B*T, vocab] fp32 logits are never written
 * (csrc/decoder_chain.hip head_ce_kernel: 64 rows per workgroup against the whole vocabulary twice, online softmax statistics; decoder
 * hidden 256).  1 (default; KZV_HEAD_CE), 0: head GEMM + ce_kernel, -1: the environment's default.  kzv_forward_loss with d_logits != NULL
 * always takes the GEMM (it returns them).  Same arithmetic up to fp32 summation order. */
int kzv_set_head_ce(int on);
int kzv_decode_step_graph(kzv_model* m, const int64_t* d_tokens, const int32_t* d_posids, const uint8_t* d_valid, int64_t ld_valid,
                          float* d_logits, void* stream);

/* loss.backward() for the step above: fills the bound fp32 grad buffer (which must be zero on entry;
 * kzv_zero_grads does that).  Backward is split in `kzv_backward_segments()` segments so the host can
 * launch an RCCL all-reduce for a segment's finished gradients while later segments still run
 * (Lightning DDP bucket overlap, scripts/train_trocr.py:166-169).  After segment s returns (work
 * enqueued on `stream`), grads in [lo, hi) of kzv_backward_segment_range(s) are final. */
int kzv_zero_grads(kzv_model* m, void* stream);
int kzv_backward_segments(const kzv_model* m);
int kzv_backward_segment(kzv_model* m, int seg, void* stream);
int kzv_backward_segment_range(const kzv_model* m, int seg, int64_t* lo, int64_t* hi);
int kzv_backward(kzv_model* m, void* stream);               /* all segments */

/* gradient_clip_val (scripts/train_trocr.py:175) + RAdamScheduleFree.step (trocr_model.py:412-421),
 * fused over the flat buffers.  d_z, d_v: optimizer state (z iterate, second moment), same layout.
 * d_scratch: >= 4 KiB fp32 scratch (partial norms).  Host supplies the step scalars (kzv/optim.py). */
typedef struct kzv_opt_step {
    float lr_t;          /* lr * RAdam rectification for this step (0 in the silent phase) */
    float ckp1;          /* schedule-free averaging weight c_{k+1} */
    float beta1, beta2, eps, weight_decay;
    float bias_correction2;
    int32_t adaptive;    /* 1 once rho_t > 4 (use v), else plain (silent) step */
    float max_grad_norm; /* <= 0 disables clipping */
    float grad_scale;    /* multiplied into grads before everything (1/world for DDP mean) */
    float one_minus_beta2; /* 1 - beta2 evaluated in double on the host, as torch's addcmul_(value=1 - beta2) does: in fp32
                            * 1.f - 0.999f is off by 1.3e-5 relative, a bias on every second-moment entry */
} kzv_opt_step;
int kzv_grad_sqnorm(const float* d_grads, int64_t n, float* d_out1, float* d_scratch, void* stream);
int kzv_clip_and_step(float* d_params, float* d_z, float* d_v, const float* d_grads, int64_t n,
                      const float* d_sqnorm, const kzv_opt_step* s, void* stream);
/* The same step with the EMA of the parameters (src/callbacks/ema.py:51-58: shadow = decay * shadow + (1 - decay) * param
 * after every batch) updated in the same pass from the freshly stepped parameters; d_ema NULL = no EMA. */
int kzv_clip_and_step_ema(float* d_params, float* d_z, float* d_v, const float* d_grads, int64_t n,
                          const float* d_sqnorm, const kzv_opt_step* s, float* d_ema, float ema_decay, void* stream);
/* optimizer.eval()/train() parameter swap (trocr_model.py:423-451): p <- p + w*(z - p) */
int kzv_lerp_params(float* d_params, const float* d_z, int64_t n, float w, void* stream);

/* -------------------------------------------------------- per-op entry points (unit parity tests) */
enum { KZV_EPI_BF16 = 0, KZV_EPI_F32 = 1,
       KZV_EPI_GELU = 2,       /* C = gelu_erf(acc + bias) bf16, aux = gelu_erf'(acc + bias) bf16 (saved for backward) */
       KZV_EPI_RESID = 3,      /* C = dropout(acc + bias) + resid, fp32 */
       KZV_EPI_DGELU = 4,      /* C = acc * aux, bf16 (aux = the derivative saved by the forward GELU epilogue) */
       KZV_EPI_GELU_F32 = 5    /* like GELU but C is fp32 (feeds a LayerNorm) */ };

/* C[M,N] = A[M,K] . B[N,K]^T (+bias) with a fused epilogue; bf16 operands, fp32 accumulate (MFMA).
 * Replaces every nn.Linear forward / input-gradient on the path. */
typedef struct kzv_gemm_nt_args {
    const void* A; int64_t lda;       /* bf16 [M,K] */
    const void* B; int64_t ldb;       /* bf16 [n_valid,K] */
    void* C; int64_t ldc;             /* bf16 or fp32 [M,N] by epilogue */
    const float* bias;                /* fp32 [n_valid] or NULL */
    const float* resid; int64_t ldr;  /* KZV_EPI_RESID: fp32 [M,N] */
    void* aux; int64_t ldaux;         /* GELU: bf16 gelu'(pre-activation) OUT (what backward needs); DGELU: the same IN */
    int32_t M, N, K, n_valid;         /* N = columns stored (mult of 4), rows of B >= n_valid read as 0 */
    float drop_p; uint32_t drop_key;  /* KZV_EPI_RESID dropout on (acc+bias); p=0 -> off */
} kzv_gemm_nt_args;
int kzv_gemm_nt(const kzv_gemm_nt_args* a, int epilogue, void* stream);
/* kzv_gemm_nt picks a kernel by shape: the LDS-staged 128x128 / 256x256 kernels, except for M <= rows_max_m (default 0 =
 * never; KZV_ROWS_MAX_M), which takes the few-rows kernel the generation step (kzv_decode_step*) uses internally for its
 * M = batch GEMMs (one wave per 16x64 tile).  The per-op tests raise the threshold to check that kernel through this entry. */
int kzv_set_rows_max_m(int n);
/* K-loop schedule of the persistent 256x256 kernel behind kzv_gemm_nt (large shapes, one-store epilogues): 0 = eight-phase
 * ping-pong (gemm_nt256p.hip), 1 = free-running, two barriers per K-tile (gemm_nt256f.hip).  Same results bit for bit (same
 * per-accumulator summation order).  Default: KZV_NT_FREE (environment) or the library's choice; n < 0 restores the default. */
int kzv_set_nt_schedule(int n);
/* Bit 1 of the schedule (n = 2, 3): the epilogues whose bit is set in `mask` (1 << KZV_EPI_*; default all) run on the four-wave
 * 256x128 kernel, two workgroups per CU out of phase so that one's drain overlaps the other's K loop (gemm_nt256h.hip; shapes
 * with K % 384 == 0, others keep the 256x256 kernels).  Same results bit for bit.  `us`: how late the second workgroup of a CU
 * starts (microseconds, ~; KZV_NTH_STAGGER, default 8). */
int kzv_set_nt_half_epilogues(int mask);
int kzv_set_nt_half_stagger(int us);
/* The same choice for the 256x256 weight-gradient kernel behind kzv_gemm_tn (gemm_tn256.hip; KZV_TN_FREE). */
int kzv_set_tn_schedule(int n);
/* The generation step's GEMMs with the decoder's LayerNorms folded in (hidden size 256): RoBERTa is post-LN, so every sub-layer
 * output s is normalised once and LN(s) feeds one GEMM as A and one later residual add; at M = batch rows the consumers normalise
 * themselves instead of a LayerNorm launch per sub-layer.  ln_a != NULL: A = LN(ln_a [M,256]) (K must be 256; A is ignored);
 * ln_r != NULL: the RESID epilogue's residual = LN(ln_r [M,256]) (N must be 256).  Epilogues: BF16, F32, GELU, GELU_F32 with
 * ln_a; RESID with ln_r.  Same arithmetic as kzv_layernorm_fwd + kzv_gemm_nt up to summation order. */
typedef struct kzv_gemm_rows_ln_args {
    const void* A; int64_t lda;       /* bf16 [M,K] or NULL with ln_a */
    const void* B; int64_t ldb;       /* bf16 [n_valid,K] */
    void* C; int64_t ldc;
    const float* bias;
    void* aux; int64_t ldaux;         /* GELU / GELU_F32 */
    int32_t M, N, K, n_valid;
    const float* ln_a; const float* ln_a_gamma; const float* ln_a_beta;
    const float* ln_r; const float* ln_r_gamma; const float* ln_r_beta;
    float eps;
} kzv_gemm_rows_ln_args;
int kzv_gemm_rows_ln(const kzv_gemm_rows_ln_args* a, int epilogue, void* stream);

/* ---- fp8 weight path (BASELINE.json configs[4]: "fp8 MFMA weight path on CDNA4"; beyond the reference, which has no fp8) ----
 * C[M,N] = (A8[M,K] . B8[N,K]^T) * a_scale[m] * b_scale[n] (+bias) with the BF16 / GELU / RESID / DGELU epilogue of kzv_gemm_nt.
 * A8, B8: OCP e4m3 bytes; row m of A8 holds A[m,:] / a_scale[m] (likewise B8 / b_scale).  fp32 accumulate on the block-scaled
 * MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales): twice the bf16 MFMA rate.  K % 256 == 0.
 * GELU only: when c8 is set, a second copy of C as e4m3, quantised with the per-tensor multiplier *c8_qscale (device scalar),
 * feeds the next GEMM; max |C| is folded into *c8_amax (atomic max), from which the caller derives the next multiplier. */
typedef struct kzv_gemm_nt_fp8_args {
    const void* A; int64_t lda;       /* e4m3 [M,K] */
    const void* B; int64_t ldb;       /* e4m3 [n_valid,K] */
    const float* a_scale;             /* fp32 [M] */
    const float* b_scale;             /* fp32 [n_valid] */
    void* C; int64_t ldc;             /* bf16 (BF16, GELU) or fp32 (RESID) [M,N] */
    const float* bias;                /* fp32 [n_valid] or NULL */
    const float* resid; int64_t ldr;  /* RESID: fp32 [M,N] */
    void* aux; int64_t ldaux;         /* GELU: bf16 gelu'(pre-activation) OUT */
    void* c8; int64_t ldc8;           /* GELU: optional e4m3 [M,N] OUT */
    const float* c8_qscale; float* c8_amax;
    int32_t M, N, K, n_valid;
    float drop_p; uint32_t drop_key;  /* RESID dropout on (acc*scales+bias) */
    const float* c8_rowq;             /* DGELU: fp32 [M], c8[m,:] = e4m3(C[m,:] * c8_rowq[m]) (required with DGELU) */
} kzv_gemm_nt_fp8_args;
int kzv_gemm_nt_fp8(const kzv_gemm_nt_fp8_args* a, int epilogue, void* stream);
/* q[r,:] = e4m3(x[r,:] * 448 / amax_r), scale[r] = amax_r / 448 (1 for an all-zero row); cols % 4 == 0.  The quantiser of the
 * weight copies (per output row) and, fused into LayerNorm, of its output rows. */
int kzv_quant_rows_fp8(const float* x, int64_t rows, int64_t cols, void* q, float* scale, void* stream);
/* kzv_layernorm_fwd that also writes the e4m3 copy of each output row and its scale. */
int kzv_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_bf16, void* y_fp8, float* y_scale,
                          float* stats, int rows, int H, float eps, void* stream);
/* Model switch (call before kzv_model_bind; encoder hidden and ffn must be multiples of 256): 1 = the encoder's QKV, fc1 and fc2
 * forward GEMMs run on e4m3 operands (weights quantised per output row at kzv_model_sync_weights; LayerNorm outputs per token
 * row; GELU outputs per tensor with the previous forward's amax); 2 = also the MLP's two input-gradient GEMMs (transposed e4m3
 * weights; gradient rows quantised by their amax in the LayerNorm backward, and per row by a norm bound in the DGELU epilogue).
 * Everything else -- weight gradients, attention, the decoder -- stays bf16. */
int kzv_set_fp8(kzv_model* m, int mode);
int kzv_get_fp8(const kzv_model* m);
/* Parity hook: the per-tensor multipliers the last forward quantised each encoder layer's GELU output with -> d_out[enc_layers]. */
int kzv_fp8_act_scales(const kzv_model* m, float* d_out, void* stream);

/* OUT[N,K] (+)= P[Mtok,N]^T . Q[Mtok,K]   (weight gradient).  fp32 MFMA accumulation; small outputs add their token splits with
 * fp32 atomics, outputs of >= 9 tiles of 256 x 256 pass each split's partial tile through a workspace rounded to bf16 and fold them
 * in fp32 in a fixed order (bit-identical from run to run; error <= 2^-9 of a split's partial sum -- the reference's autocast GEMM
 * rounds the whole weight gradient to bf16). */
typedef struct kzv_gemm_tn_args {
    const void* P; int64_t ldp;       /* bf16 [Mtok, N] */
    const void* Q; int64_t ldq;       /* bf16 [Mtok, K] */
    float* OUT; int64_t ldo;          /* fp32 [n_store, K] accumulated into */
    int32_t Mtok, N, K, n_store;
    float* dbias;                     /* optional fp32 [n_store]: += column sums of P (bias gradient) */
} kzv_gemm_tn_args;
int kzv_gemm_tn(const kzv_gemm_tn_args* a, void* stream);

/* A Linear's input gradient (kzv_gemm_nt arguments, epilogue BF16 / F32 / RESID / DGELU) and weight gradient (kzv_gemm_tn arguments) from
 * the same dY as ONE launch when both take the 256x256 kernels (>= 256 output tiles of the first, >= 9 of the second, 256 CUs): every
 * workgroup runs its gemm_nt tiles and then a token range of one weight-gradient tile, sized on the host so that all workgroups finish
 * together -- the two launches each end on a partly filled round otherwise.  Falls back to the two separate launches.  The gemm_nt
 * result is bit-identical to kzv_gemm_nt's; the weight gradient differs by the bf16 rounding of its partial tiles (other token splits).
 * model.cpp's encoder backward calls it per nn.Linear (HF vit:192-254).  OFF by default (kzv_set_pair(1) / KZV_PAIR=1 turn it on):
 * measured +0.4 % img/s on the training step, one kernel boundary per pair and nothing from the balancing. */
int kzv_gemm_dgrad_wgrad(const kzv_gemm_nt_args* nt, int epilogue, const kzv_gemm_tn_args* tn, void* stream);
int kzv_set_pair(int on);

/* The same two products on fp32 OPERANDS (A, B, P, Q are float; leading dimensions in elements, multiples of 4; K % 32 == 0 for
 * the NT form), on the f32-input matrix instruction: exact fp32 products, fp32 accumulation (csrc/gemm_f32.hip).  For the model of
 * ocr_lightning/model.py, which the reference trains in fp32 (ocr_lightning/train.py:132-140; its own test wants singles ==
 * batched at 1e-6, ocr_lightning/tests/test_model.py:48-76).  kzv_gemm_nt_f32 takes the epilogues KZV_EPI_F32 and KZV_EPI_RESID
 * (no dropout); aux is ignored.  Launches with few output tiles split the reduction; the partial tiles are summed in a fixed order. */
int kzv_gemm_nt_f32(const kzv_gemm_nt_args* a, int epilogue, void* stream);
int kzv_gemm_tn_f32(const kzv_gemm_tn_args* a, void* stream);

/* LayerNorm over the last dim (fp32 statistics, eps inside rsqrt).  x fp32 [rows, H].
 * Replaces nn.LayerNorm in ViTLayer / RobertaLayer / heads. */
int kzv_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                      float* stats /* [rows,2] mean,rstd */, int rows, int H, float eps, void* stream);
/* dx(+)= LN backward; dy bf16 or fp32; dgamma/dbeta accumulated (atomics). */
int kzv_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, const float* stats, const float* gamma,
                      float* dx, int accumulate_dx, float* dgamma, float* dbeta, int rows, int H, void* stream);

/* Multi-head attention, one workgroup per (batch, head) on the MFMA kernels for head_dim 64 (the benchmark geometry and the
 * decoder); other head dimensions (multiples of 8 up to 128; the reference's CLI default ViT is 768 / 8 heads = 96,
 * scripts/train_trocr.py:41-43) take a plain fp32 kernel -- same results, several times slower (attention_generic.hip).
 * mode 0: no mask (ViT self-attn / decoder cross-attn); mode 1: causal AND key-not-pad (decoder self; head_dim 64 only). */
typedef struct kzv_attn_args {
    const void* Q; const void* K; const void* V;   /* bf16, row strides ldq/ldk/ldv, head h at col h*64 */
    void* O;                                        /* bf16 [B*Sq, ldo] */
    float* LSE;                                     /* fp32 [B, heads, Sq] */
    const void* dO; void* dQ; void* dK; void* dV;   /* backward only (same strides as O/Q/K/V) */
    int64_t ldq, ldk, ldv, ldo;
    const int64_t* ids; int64_t ld_ids; int32_t pad_id;   /* mode 1: decoder input ids [B, ld_ids] */
    int32_t B, heads, Sq, Sk, mode;
    float drop_p; uint32_t drop_key;
    int32_t head_dim;                                     /* 0 = 64; head h sits at column h * head_dim */
} kzv_attn_args;
int kzv_attn_fwd(const kzv_attn_args* a, void* stream);
int kzv_attn_bwd(const kzv_attn_args* a, void* stream);

/* ------------------------------------------------------------- the ResNet / BiLSTM / CTC model of ocr_lightning/model.py
 * (SURVEY.md section 8(f), row N3).  Per-op entry points; the host mirror kzv/ocr_model.py strings them together the way
 * OCRModel.forward / _shared_step do (ocr_lightning/model.py:61-88, 90-195).  Activations are NHWC (rows = pixels, columns =
 * channels) so that a convolution is kzv_ocr_im2col + kzv_gemm_nt, its weight gradient kzv_gemm_tn on the same column matrix and
 * its input gradient kzv_gemm_nt against the transposed packed weight + kzv_ocr_col2im.  bf16 GEMM operands, fp32 everything else --
 * or, after kzv_ocr_set_precision(1), fp32 operands throughout: every buffer documented as "bf16" below is then a float buffer of
 * the same shape and the GEMMs are kzv_gemm_nt_f32 / kzv_gemm_tn_f32 (the reference's own arithmetic; process-wide switch, set by
 * kzv.OCRModel before each of its passes). */
int kzv_ocr_set_precision(int fp32);
/* images fp32 [N, C, H, W] (ocr_collate_fn's stack, dataset.py:100) -> NHWC bf16 */
int kzv_ocr_nchw_to_nhwc(const float* x, void* out_bf16, int N, int C, int H, int W, void* stream);
/* nn.Conv2d forward operand (resnet34's 7x7/2, 3x3/1, 3x3/2 and 1x1/2 convolutions, model.py:31-32): cols bf16 [N*Ho*Wo, Kp],
 * column (kh * KW + kw) * C + c, Kp >= KH*KW*C a multiple of 8 (the GEMM wants a multiple of 64), zero-filled padding. */
int kzv_ocr_im2col(const void* x_bf16, void* cols_bf16, int N, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp, void* stream);
/* nn.Conv2d input gradient from the gradient of the column matrix (fp32 [N*Ho*Wo, Kp]): dx fp32 [N, H, W, C] (+= if accumulate) */
int kzv_ocr_col2im(const float* dcols, float* dx, int N, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp, int accumulate, void* stream);
/* conv weight fp32 [Cout, Cin, KH, KW] (the state_dict layout) -> packed bf16 [Cout, Kp] (+ its transpose [Kp, Cout] if wpT) */
int kzv_ocr_conv_weight(const float* w, void* wp_bf16, void* wpT_bf16, int Cout, int Cin, int KH, int KW, int Kp, void* stream);
/* gradient of the packed weight fp32 [Cout, Kp] accumulated into the state_dict layout [Cout, Cin, KH, KW] */
int kzv_ocr_conv_wgrad_unpack(const float* gp, float* g, int Cout, int Cin, int KH, int KW, int Kp, void* stream);
/* Both for n <= 40 convolutions in ONE launch each (what a ResNet34 step needs after its optimizer step / at the end of its backward):
 * HOST arrays of n device pointers and geom[n][5] = (Cout, Cin, KH, KW, Kp). */
int kzv_ocr_conv_weight_multi(int n, const float* const* w, void* const* wp_bf16, void* const* wpT_bf16, const int32_t* geom, void* stream);
int kzv_ocr_conv_wgrad_unpack_multi(int n, const float* const* gp, float* const* g, const int32_t* geom, void* stream);
/* nn.BatchNorm2d (+ the BasicBlock's residual add and ReLU): out bf16 = [relu](gamma * (y - mean) * rstd + beta [+ resid]); y fp32
 * [M, C].  train: batch statistics (biased variance), running statistics updated with `momentum` and the unbiased variance;
 * eval: the running statistics.  mean / rstd [C] are kept for the backward.  d_scratch: kzv_ocr_bn_scratch_floats(M, C) floats.
 * The statistics (and the backward's dgamma / dbeta) are reduced in a fixed order, without atomics: a one-ulp difference of a mean
 * flips the bf16 rounding of an activation and, through ReLU masks, ~1 % of the gradients downstream. */
int64_t kzv_ocr_bn_scratch_floats(int64_t M, int C);
int kzv_ocr_bn_fwd(const float* y, int64_t M, int C, const float* gamma, const float* beta, float* run_mean, float* run_var, float* mean,
                   float* rstd, const void* resid_bf16, void* out_bf16, int relu, int train, float eps, float momentum, float* d_scratch, void* stream);
/* its backward: da fp32 [M, C] = gradient of the (post-ReLU) output; dz fp32 = da masked by the ReLU (also the gradient of the
 * residual input); dgamma / dbeta are accumulated into (+=); dy bf16 = gradient of y.  d_scratch as for the forward. */
int kzv_ocr_bn_bwd(const float* da, const void* a_bf16, const float* y, int64_t M, int C, const float* mean, const float* rstd, const float* gamma,
                   float* dz, float* dgamma, float* dbeta, void* dy_bf16, int relu, int train, float* d_scratch, void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1) of the ResNet stem, NHWC bf16; idx = winning tap per output (first maximum, like torch) */
int kzv_ocr_maxpool_fwd(const void* x_bf16, void* out_bf16, unsigned char* idx, int N, int H, int W, int C, void* stream);
int kzv_ocr_maxpool_bwd(const float* dout, const unsigned char* idx, float* dx, int N, int H, int W, int C, void* stream);
/* nn.AdaptiveAvgPool2d((1, 1)) + flatten (model.py:34, 66-67): feat fp32 and bf16 [N, C] */
int kzv_ocr_avgpool_fwd(const void* x_bf16, float* feat_f32, void* feat_bf16, int N, int HW, int C, void* stream);
int kzv_ocr_avgpool_bwd(const float* dfeat, float* dx, int N, int HW, int C, void* stream);
/* One nn.LSTM direction on a length-1 sequence with zero initial state (model.py:40-47, 73-75): gates fp32 [B, 4H] = x W_ih^T + b_ih
 * (a kzv_gemm_nt; torch order i, f, g, o), b_hh [4H] added here; h written into columns of the [B, ldh] output (forward | reverse
 * halves).  With h0 = c0 = 0 the recurrent weight W_hh multiplies zeros and the forget gate receives no gradient. */
int kzv_ocr_lstm_cell_fwd(const float* gates, const float* b_hh, float* h_f32, void* h_bf16, int64_t ldh, int B, int H, void* stream);
int kzv_ocr_lstm_cell_bwd(const float* gates, const float* b_hh, const float* dh, int64_t lddh, void* dgates_bf16, int B, int H, void* stream);
/* F.log_softmax(dim = -1) on fp32 rows (model.py:125) */
int kzv_ocr_log_softmax(const float* x, float* lp, int rows, int C, void* stream);
/* nn.CTCLoss(blank, zero_infinity) on lp fp32 [T, B, C] (model.py:51-55, 171-176): d_nll[b] = -log p(target_b) (0 where infinite and
 * zero_infinity); if d_dlogits: the gradient with respect to the logits behind lp, times d_gscale[b] (the caller folds the
 * reduction -- 'mean' = 1 / (max(target_len, 1) * B) -- and the loss weight into it).  d_scratch: 2 * B * T * (2 * min(max_target_len, T) + 1)
 * floats (a label longer than T has no alignment: loss inf -> 0 with zero_infinity, zero gradient, whatever its length; T <= 511).  targets int64 [B, ld_targets]; lengths int64 [B]. */
int kzv_ocr_ctc(const float* lp, const int64_t* targets, int64_t ld_targets, const int64_t* input_lengths, const int64_t* target_lengths, int T,
                int B, int C, int blank, int zero_infinity, int max_target_len, float* d_scratch, float* d_nll, const float* d_gscale,
                float* d_dlogits, void* stream);
/* The localisation loss of _shared_step (model.py:100-122): mean over the samples with n_i = min(count_i, max_boxes) > 0 of the mean
 * SmoothL1 (beta 1) between pred[i, :n_i] and gt[i, :n_i].  *d_loss += loss (zero it first); d_dpred fp32 [B, max_boxes, 4]. */
int kzv_ocr_smooth_l1_boxes(const float* pred, int max_boxes, const float* gt, int gt_boxes, const int32_t* counts, int B, float* d_loss,
                            float* d_dpred, void* stream);
/* torch.optim.Adam (model.py:197; no weight decay, no amsgrad), `step` = 1-based step count */
int kzv_ocr_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step, void* stream);
/* The same step with the bias corrections in device memory: d_bc[0] = 1 - beta1^t, d_bc[1] = sqrt(1 - beta2^t) (fp32).  For a step
 * replayed from a captured hipGraph (kzv.OCRModel.fit_step): no step-dependent scalar among the kernel arguments. */
int kzv_ocr_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, const float* d_bc,
                     void* stream);
int kzv_ocr_cast_bf16(const float* x, void* out_bf16, int64_t n, void* stream);
int kzv_ocr_cast_transpose(const float* x, void* out_bf16, int rows, int cols, void* stream);   /* [rows, cols] fp32 -> bf16 [cols, rows] */

/* ------------------------------------------------------------- measurement hooks (bench.py roofline leg)
 * When enabled, every launch of the hot kernels is bracketed by HIP events on its own stream.
 * kind: 0 gemm_nt, 1 gemm_tn, 2 attn_fwd, 3 attn_bwd, 4 gemm_nt_fp8.  work = algorithmic FLOPs (2*M*N*K; 4*B*h*Sq*Sk*64 /
 * 10*B*h*Sq*Sk*64 for attention fwd / bwd).  Collect after synchronising the stream.
 * kzv_prof_select restricts recording to the kinds in `kind_mask` (bit k = kind k; default all): every bracketed
 * launch costs ~2 us of stream time, so the timed region records only the kernel it reports. */
int kzv_prof_enable(int on, int capacity);
int kzv_prof_select(unsigned kind_mask);
int kzv_prof_collect(int kind, double* total_ms, double* total_flops, int64_t* launches);
/* Sampling: bracket only every `stride`-th launch of each selected kind (default 1 = all).  A bracket costs ~2 us of stream
 * time; with a stride co-prime to the launches per step every launch site is sampled equally often over `stride` steps.
 * kzv_prof_collect then sums the SAMPLED launches; kzv_prof_seen = launches of that kind seen since kzv_prof_enable(1, n). */
int kzv_prof_sample(int stride);
int64_t kzv_prof_seen(int kind);
uint32_t kzv_drop_key(uint64_t seed, uint32_t site);

/* Dropout call sites of the training step (nn.Dropout / F.dropout in the reference).  kzv_forward_loss(train=1, seed) keys
 * site s with kzv_drop_key(seed, s); backward regenerates the same masks from the same keys (nothing is stored).
 *   KZV_SITE_ENC_EMB           ViTEncoder dropout on cat(cls, patches) + pos            (trocr_model.py:190)      [B*Se, He]
 *   KZV_SITE_ENC_LAYER(i, 0)   attention-probability dropout of ViT layer i            (HF modeling_vit.py:184)  [B*heads*Sq, Sk]
 *   KZV_SITE_ENC_LAYER(i, 1)   ViTLayer.dropout on the attention block's output        (HF modeling_vit.py:276)  [B*Se, He]
 *   KZV_SITE_ENC_LAYER(i, 2)   ViTLayer.dropout on the MLP output                      (HF modeling_vit.py:283)  [B*Se, He]
 *   KZV_SITE_DEC_EMB           RobertaEmbeddings.dropout                               (HF modeling_roberta.py:120) [B*T, Hd]
 *   KZV_SITE_DEC_LAYER(i, 0/2) self / cross attention-probability dropout              (HF modeling_roberta.py:178)
 *   KZV_SITE_DEC_LAYER(i, 1/3) RobertaSelfOutput.dropout of the self / cross block     (HF modeling_roberta.py:338)
 *   KZV_SITE_DEC_LAYER(i, 4)   RobertaOutput.dropout (FFN)                             (HF modeling_roberta.py:396)
 * Element index of a [rows, cols] hidden site: row * cols + col (kzv_debug_dropout_mask); an attention site is indexed by
 * (b * heads + h, q, key) (kzv_debug_attn_dropout_mask).  Rows are token rows of the PACKED decoder ([B, t_active],
 * kzv_set_active_length). */
#define KZV_SITE_ENC_EMB 1u
#define KZV_SITE_ENC_LAYER(i, k) (16u + 4u * (uint32_t)(i) + (uint32_t)(k))
#define KZV_SITE_DEC_EMB 1000u
#define KZV_SITE_DEC_LAYER(i, k) (1016u + 8u * (uint32_t)(i) + (uint32_t)(k))

/* Debug / parity entry (mask replay): writes the multiplier (0 or 1/P(keep)) the kernels apply to element
 * index row * ld_index + col under `key` and drop probability p, for row < rows, col < cols: d_out fp32 [rows, cols].
 * Uses the same device hash as every fused dropout epilogue, so a test can hand the exact masks of a training step to the
 * CPU oracle and compare logits, loss and every gradient with dropout ON. */
int kzv_debug_dropout_mask(uint32_t key, float p, int64_t rows, int64_t cols, int64_t ld_index, float* d_out, void* stream);
/* The same for an ATTENTION-PROBABILITY site (KZV_SITE_*_LAYER(i, 0 / 2); F.dropout on the softmax output, HF modeling_vit.py:184,
 * modeling_roberta.py:178): d_out fp32 [pairs * Sq, Sk], pairs = batch * heads.  These sites draw their bits from one 32-bit hash
 * per 4 x 4 block of a (batch, head)'s [Sq, Sk] matrix and a 16-bit multiply per element (kzv_common.h, "attention-probability
 * dropout"; numpy statement: oracle/attn_dropout.py), because the forward kernel holds 4 keys of one query per lane and the
 * backward kernel 4 queries of one key, and this costs the same few instructions in both orientations. */
int kzv_debug_attn_dropout_mask(uint32_t key, float p, int64_t pairs, int32_t Sq, int32_t Sk, float* d_out, void* stream);

/* Data-parallel runs: the GEMM kernels that put exactly one workgroup on every CU (persistent gemm_nt256p, gemm_tn256)
 * double their time when a concurrently running collective holds a few CUs.  With n > 0 the launchers leave n CUs
 * free: gemm_nt falls back to the one-tile-per-workgroup kernel (its workgroups flow to whatever CUs are free) and
 * gemm_tn256 sizes its token splits to (CUs - n) workgroups.  Default 0 (or KZV_CU_RESERVE); kzv/trainer.py sets it
 * when world_size > 1. */
int kzv_set_cu_reserve(int n);

/* ------------------------------------------------------------------ N2: input pipeline on the device
 * Replaces ResizeWithPadding + ToTensor + Normalize(0.5, 0.5) (src/data/trocr_dataset.py:24-53, 97-104) for a batch of
 * decoded uint8 RGB crops of different sizes: Pillow's two-pass LANCZOS resample (bit-exact: fixed-point coefficients,
 * uint8 intermediate), centred paste on a white canvas, [-1, 1] fp32 CHW output.
 *
 * kzv_lanczos_coeffs (HOST, no GPU needed; also usable from DataLoader workers): Pillow's precompute_coeffs +
 * normalize_coeffs_8bpc for resampling `in_size` samples to `out_size`: bounds[out][2] = (first input index, count),
 * kk[out][ksize] = 22-bit fixed-point weights; returns ksize through *ksize (call with kk == NULL to query it).
 * kzv_preprocess_lines (device pointers; `rgb` and `tmp` need 4 bytes of slack after their last byte: pixels are fetched
 * as unaligned 32-bit words): rgb = the crops packed back to back (HWC, 3 bytes per pixel), desc[i] locates
 * crop i, its geometry and its coefficient tables inside `coef` (int32 offsets); tmp = scratch for the horizontal pass
 * (desc[i].tmp_off, in_h * new_w * 3 bytes per crop); max_tmp_pixels = max over crops of in_h * new_w (sizes the launch);
 * lut256 = the 256 possible output values; out = [n, 3, target_h, target_w] fp32. */
typedef struct {
    int64_t src_off;            /* byte offset of the crop in `rgb` */
    int64_t tmp_off;            /* byte offset of its scratch in `tmp` */
    int64_t hb_off, hk_off;     /* int32 offsets in `coef`: horizontal bounds [new_w][2], weights TRANSPOSED [hk_size][new_w] */
    int64_t vb_off, vk_off;     /* vertical bounds [new_h][2], weights [new_h][vk_size] */
    int32_t in_h, in_w, new_h, new_w, paste_x, paste_y, hk_size, vk_size;
} kzv_line_desc;
int kzv_lanczos_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int* ksize);
int kzv_preprocess_lines(const uint8_t* rgb, const kzv_line_desc* desc, const int32_t* coef, int n, int target_h, int target_w,
                         int64_t max_tmp_pixels, const float* lut256, uint8_t* tmp, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KZV_H */
