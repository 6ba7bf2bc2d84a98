// Internal launcher API (C++ linkage) used by model.cpp; the per-op C ABI in include/kzv.h wraps a subset.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kzv_common.h"

// layernorm.hip
int kzv_ln_fwd_ex(const float* x, const float* gamma, const float* beta, void* y16, float* y32, float* stats,
                  int rows, int H, float eps, int seq, int drop_first, float drop_p, uint32_t drop_key, hipStream_t s,
                  void* y8 = nullptr, float* y8_scale = nullptr);    // y8: e4m3 copy with one scale per row (fp8 path)
// fp8 input-gradient path: e4m3 copy of the masked dx rows (per-row amax scale) + the per-row quantisation multiplier (and its
// inverse) of the gradient tensor computed from them next, bounded through *wnorm = the largest row norm of that GEMM's weight
struct KzvLnBwdF8 { void* out8; float* scale; float* rq; float* rqinv; const float* wnorm; };
int kzv_ln_bwd_ex(const void* dy, int dy_is_f32, const float* x, const float* stats, const float* gamma, float* dx,
                  int accumulate_dx, float* dgamma, float* dbeta, int rows, int H, int seq, int drop_first,
                  float drop_p, uint32_t drop_key, hipStream_t s, bf16_t* out16 = nullptr, float out_drop_p = 0.f,
                  uint32_t out_drop_key = 0, const KzvLnBwdF8* f8 = nullptr);

// While one of these is alive, kzv_ln_bwd_ex leaves the fold of its gamma / beta partial sums pending (each call gets a partial
// region of its own); the outermost scope's destructor folds all of them in ONE launch on `stream`.  Without a scope every call
// folds at once (the per-op C ABI).
struct KzvLnDeferScope { explicit KzvLnDeferScope(hipStream_t stream); ~KzvLnDeferScope(); hipStream_t s; };

// for kernels that fuse a LayerNorm backward (decoder_chain.hip): see layernorm.hip
constexpr int KZV_LN_SLOTS = 32;
float* kzv_ln_partial_region(float* dgamma, float* dbeta, int H, hipStream_t s, bool* fold_now);
int kzv_ln_partial_fold(float* partial, float* dgamma, float* dbeta, int H, hipStream_t s);

// elementwise.hip
int kzv_im2row(const float* px, bf16_t* out, int B, int C, int H, int W, int ph, int pw, hipStream_t s);
// gw / gw_max: patches per grid row of this batch / of the position table (width buckets: row of patch p = (p / gw) * gw_max + p % gw)
int kzv_embed_assemble(const float* patch_emb, const float* cls, const float* pos, float* x0, int B, int np, int He,
                       float drop_p, uint32_t key, hipStream_t s, int gw = 0, int gw_max = 0);
int kzv_embed_assemble_bwd(const float* dx0, bf16_t* dpatch, float* dcls, float* dpos, float* dpatch_bias, int B, int np,
                           int He, float drop_p, uint32_t key, hipStream_t s, int gw = 0, int gw_max = 0);
int kzv_cast_drop_colsum(const float* g, bf16_t* out, float* dbias, int M, int N, float drop_p, uint32_t key, hipStream_t s,
                         const bf16_t* gelu_pre = nullptr);
int kzv_colsum_bf16(const bf16_t* g, int64_t ld, float* dbias, int M, int N, hipStream_t s);
int kzv_dec_prepare(const int64_t* labels, int B, int L, int T, int pad, int max_pos, int* posids, float* count, int* err, hipStream_t s);
int kzv_embed_gather(const int64_t* labels, int L, const int* posids, const float* word, const float* type0,
                     const float* postab, float* out, int B, int T, int Hd, hipStream_t s);
int kzv_embed_scatter_bwd(const float* dsum, const int64_t* labels, int L, const int* posids, float* dword, float* dtype0,
                          float* dpostab, int B, int T, int Hd, int pad, hipStream_t s);
int kzv_ce_fwd_bwd(const float* logits, int64_t ldl, const int64_t* labels, int L, int B, int T, int V, int pad,
                   const float* count, float* loss, bf16_t* dlogits, hipStream_t s);
int kzv_copy_logits(const float* logits, int64_t ldl, float* out, int rows, int V, hipStream_t s);

struct KzvCastDesc {         // one 2-D fp32 weight -> bf16 copy (+ optional transposed copy)
    const float* src; bf16_t* dst; bf16_t* dstT;
    int rows, cols; int64_t ldT; int tile0, tiles_c;
};
int kzv_cast_weights(const KzvCastDesc* d_desc, int ndesc, int total_tiles, hipStream_t s);

// fp8 path: fp32 rows -> e4m3 rows quantised by their own largest |x| (dst ~ x * 448 / amax, scale = amax / 448)
struct KzvQuantDesc {
    const float* src; unsigned char* dst; float* scale; int rows, cols, row0;
    const bf16_t* src16; int64_t ld16;      // src == nullptr: bf16 rows with stride ld16 (the transposed weight copies)
    float* normmax;                         // optional: *normmax = max(*normmax, ||row||_2) (atomic; reset by the caller)
};
int kzv_quant_rows(const KzvQuantDesc* d_desc, int ndesc, int total_rows, hipStream_t s);
// fp8 path, once per forward: next multiplier of every per-tensor activation site from the largest |value| the last
// forward saw (power of two, one binade of headroom; unchanged while amax == 0), amax reset, and the dequantisation
// factor 1 / qscale[site] broadcast to the `rows` entries of that site's row-scale array (what the GEMM reads)
int kzv_fp8_roll(float* qscale, float* amax, float* row_scales, int sites, int rows, hipStream_t s);

// gemm.hip: n independent weight gradients in one grid
struct kzv_gemm_tn_args;
int kzv_gemm_tn_group(const kzv_gemm_tn_args* a, int n, hipStream_t s);

// optim.hip
int kzv_sqnorm(const float* g, int64_t n, float* out1, float* scratch, hipStream_t s);

// attention.hip : see include/kzv.h (kzv_attn_fwd / kzv_attn_bwd)

// decode.hip (KV-cached generation step)
int kzv_attn_decode(const bf16_t* q, int64_t ldq, const bf16_t* knew, const bf16_t* vnew, int64_t ldnew, bf16_t* K, bf16_t* V, int64_t kb,
                    int64_t kj, const unsigned char* valid, int64_t ldvalid, bf16_t* out, int64_t ldo, int B, int heads, int nkeys,
                    int append_at, hipStream_t s, const int* tptr = nullptr, int group = 1, int* rows = nullptr, int64_t ldrows = 0, int64_t kh = 64);
int kzv_kv_rows(const int* src, int* dst, const int64_t* parent, int B, int ld, int len, hipStream_t s);
int kzv_step_inc(int* d_t, hipStream_t s);
int kzv_cross_relayout(const bf16_t* src, bf16_t* dst, int images, int keys, int heads, int layers2, hipStream_t s);

// decode_fused.hip: the whole KV-cached decoder step (embeddings .. LM-head dense) of a generation token in one launch
#define KZV_DECODE_FUSED_MAX_LAYERS 12
struct KzvDecodeFusedLayer {
    const bf16_t *wqkv, *wo, *wcq, *wco, *wfc1, *wfc2;          // bf16 copies in MFMA fragment order (kzv_pack_frag)
    const float *bqkv, *bo, *bcq, *bco, *bfc1, *bfc2;
    const float *ln1w, *ln1b, *ln2w, *ln2b, *ln3w, *ln3b;
};
struct KzvDecodeFused {
    KzvDecodeFusedLayer layers[KZV_DECODE_FUSED_MAX_LAYERS]; int nlayers;
    const int64_t* tokens; const int* posids;
    const float *word, *type0, *postab, *elnw, *elnb;
    const bf16_t* whd; const float* bhd; float* hd_out;
    bf16_t* cache; int64_t plane; const bf16_t* ckv; int64_t plane2;
    const unsigned char* valid; int64_t ldvalid;
    const int* tptr; int t, T, npa, B, group;
    int* rows; float eps;
};
int kzv_decode_fused_supported(int Hd, int heads, int Fd, int layers, int group, int T, int npa);
int kzv_decode_fused_launch(const KzvDecodeFused& a, hipStream_t s);
int kzv_pack_frag(const bf16_t* W, bf16_t* out, int N, int K, hipStream_t s);     // [N, K] row-major -> fragment order

// decoder_chain.hip: the linear chains of a decoder layer in the training forward, one launch each (weights in fragment order)
struct KzvDecChainA {          // s1 = drop(ctx Wo^T + b) + xres;  x1 = LN1(s1);  cq = x1 Wcq^T + b
    const bf16_t* ctx; const float* xres; const bf16_t* wo; const float* bo; float drop_p; uint32_t drop_key; const float *g1, *b1;
    const bf16_t* wcq; const float* bcq;
    float *s1, *st1, *x1; bf16_t* x1h; bf16_t* cq; int M; float eps;
    // xres == null: the residual is the PREVIOUS layer's x3 = LN3(s3), recomputed from its sum, statistics and weights; x1 may be null
    // (not written: chain B recomputes it the same way)
    const float *xres_s = nullptr, *xres_st = nullptr, *xres_g = nullptr, *xres_b = nullptr;
};
struct KzvDecChainB {          // s2 = drop(cctx Wco^T + b) + x1; x2 = LN2(s2); act = gelu(x2 Wfc1^T + b); s3 = drop(act Wfc2^T + b) + x2; x3 = LN3(s3); [next qkv]
    const bf16_t* cctx; const float* x1; const bf16_t* wco; const float* bco; float drop_p; uint32_t drop3_key, drop4_key; const float *g2, *b2;
    const bf16_t* wfc1; const float* bfc1; const bf16_t* wfc2; const float* bfc2; const float *g3, *b3;
    const bf16_t* wqkv; const float* bqkv;
    float *s2, *st2, *x2; bf16_t* x2h; bf16_t *pre, *act; float *s3, *st3, *x3; bf16_t* x3h; bf16_t* qkv; int M; float eps;
    // x1 == null: recomputed from (s1, st1, g1, b1); x2 / x3 may be null (not written)
    const float *s1 = nullptr, *st1 = nullptr, *g1 = nullptr, *b1 = nullptr;
};
int kzv_dec_chain_supported(int Hd, int Fd);
int kzv_dec_chain_a(const KzvDecChainA& a, hipStream_t s);
int kzv_dec_chain_b(const KzvDecChainB& a, hipStream_t s);
struct KzvPackJob { const bf16_t* src; bf16_t* dst; int N, K; int n_valid = 0; int ld = 0; int k_valid = 0; };      // rows >= n_valid / columns >= k_valid (> 0) pack as zeros; ld: source row stride (0 = K)
// LM head + cross-entropy in one launch (decoder_chain.hip): x = the head's LayerNorm output [M, 256] bf16, wp = the tied weight in
// fragment order (ceil(V / 256) * 256 rows, zeros beyond V), labels int64 [B, L] (row m = b * T + t scores labels[b][t + 1]), count = the
// number of scored rows (device), loss accumulated (+=), dlogits bf16 [M, Vp] or null
struct KzvHeadCE { const bf16_t* x; const bf16_t* wp; const float* bias; const int64_t* labels; const float* count; float* loss; bf16_t* dlogits; int M, L, T, V, Vp, pad;
                   // optional (with dlogits): dh[M, 256] = dlogits . W, the head's input gradient, from wpt = W^T [256, ceil(V / 256) * 256] in fragment order
                   const bf16_t* wpt = nullptr; bf16_t* dh = nullptr; };
int kzv_head_ce(const KzvHeadCE& a, hipStream_t s);
#define KZV_PACK_MAX_JOBS (12 * KZV_DECODE_FUSED_MAX_LAYERS + 4)
// one input-gradient GEMM of the decoder on the row-panel scheme (decoder_chain.hip): out[M, N] = a[M, K] . (packed W^T)  [+ resid | * aux]
int kzv_dec_lin(const bf16_t* a, const bf16_t* wp, void* out, const float* resid, const bf16_t* aux, int M, int N, int K, int epi, hipStream_t s);
int kzv_pack_frag_multi(const KzvPackJob* jobs, int n, hipStream_t s);
// One row-local segment of a decoder layer's BACKWARD in one launch (decoder_chain.hip, dec_bwd_seg_kernel):
//   d = a[M, K1] . (packed W1^T) (+ resid)          the input gradient of the Linear BELOW a LayerNorm (+ the residual gradient)
//   ds = LayerNorm backward of d (x = the LayerNorm's forward input, st = (mean, rstd), gamma)  -> dsum fp32, dy16 = bf16(dropout mask * ds)
//   out2 = dy16 . (packed W2^T)  [* aux]            the input gradient of the Linear ABOVE it: bf16 [M, 256] or, with aux, [M, 768] (DGELU)
// K1 = 256 or 768; dgamma / dbeta receive the LayerNorm's weight gradients (through the partial-sum regions of layernorm.hip).
struct KzvDecBwdSeg {
    const bf16_t* a; int K1; const bf16_t* wp1; const float* resid;
    const float* x; const float* st; const float* gamma; float* dgamma; float* dbeta;
    float* dsum; bf16_t* dy16; float drop_p; uint32_t drop_key;
    const bf16_t* wp2; const bf16_t* aux; bf16_t* out2; int M;
};
int kzv_dec_bwd_seg(const KzvDecBwdSeg& a, hipStream_t s);       // [N, K] row-major copies -> fragment order, one launch
