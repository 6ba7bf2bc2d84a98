// N1: KV-cached decoder step: one new token per sequence attends over cached keys / values.
//
// Replaces, for generation only, the per-step work of HF RobertaSelfAttention / RobertaCrossAttention with `use_cache=True`
// (modeling_roberta.py:186-326) as driven by `decoder.generate` (src/models/trocr_model.py:306-316).  One wave per
// (sequence, head), fp32 scores and softmax in registers (see attn_decode_kernel); the self-attention variant also appends this
// step's key and value to the cache.  The reference's default 1024x64 columns give 256 cross-attention keys.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "kzv_kernels.h"
#include <cmath>
#include <cstdlib>

namespace {

struct DecAttnP {
    const bf16_t* q; int64_t ldq;                 // [B, ldq], head h at column h*64
    const bf16_t* knew; const bf16_t* vnew; int64_t ldnew;   // this step's key/value rows [B, ldnew] (self-attention) or null
    bf16_t* K; bf16_t* V; int64_t kb, kj, kh;     // key j, head h of sequence b: K + b*kb + h*kh + j*kj (same strides for V)
    const unsigned char* valid; int64_t ldvalid;  // [B, ldvalid]: key j usable (null: all)
    bf16_t* out; int64_t ldo;
    int nkeys, append_at, heads;                  // append_at >= 0: write knew/vnew at key index append_at first
    float scale;
    const int* tptr;                              // non-null: the step index t lives in device memory (graph replay): nkeys = t + 1, append_at = t
    int group;                                    // keys / values of sequence b live at batch index b / group (beams sharing one image's cross-attention K/V)
    int* rows; int64_t ldrows;                    // non-null (beam search): cached key j of sequence b lives in cache row rows[b * ldrows + j]
};

// One WAVE per (G sequences, head), no LDS and no barrier.  Every global access is a wave-instruction over 8 key (or value) rows x
// 128 contiguous bytes: lane = (row r = lane >> 3, 16-byte chunk c = lane & 7), iteration i owns key 8 i + r.  Scores: each
// lane multiplies its chunk of the key by the matching 8 query dimensions (registers), three shuffles sum the 8 chunks, so
// the 8 lanes of a row all hold that key's score; max and sum run across the wave.  Output: the SAME lanes hold the matching
// chunk of the value row, weight it by their key's probability, and three shuffles sum the 8 rows of an iteration.
// The step's own key / value (self-attention) is taken from the projection output directly and stored to the cache on the
// side.  NI = iterations (8 keys each): 24 -> 192 keys, 40 -> 320 keys.
// G > 1 (cross-attention of beam search): the G sequences of a wave are beams of ONE image and share its keys / values, which are
// loaded once and used for G queries (at one wave per beam the four beams re-read the same 41 KB through L2: 31.7 us per call
// at 1,024 rows against 23 us for the self-attention over a cache that does not fit the Infinity Cache).
template <int NI, int G>
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecAttnP p, const int nunits) {
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);          // (G sequences, head) of this wave
    if (unit >= nunits) return;
    const int b0 = (unit / p.heads) * G, h = unit - (unit / p.heads) * p.heads, lane = threadIdx.x & 63;
    const int b = b0;                                              // G == 1: the sequence
    bf16_t* Kb = p.K + (int64_t)(b0 / p.group) * p.kb + h * p.kh;
    bf16_t* Vb = p.V + (int64_t)(b0 / p.group) * p.kb + h * p.kh;
    const int tdev = p.tptr ? *p.tptr : 0;
    const int nkeys = p.tptr ? min(tdev + 1, 8 * NI) : p.nkeys;
    const int append_at = (G > 1) ? -1 : (p.tptr ? min(tdev, 8 * NI - 1) : p.append_at);
    const int r = lane >> 3, c = lane & 7;
    const bf16_t* knew = append_at >= 0 ? p.knew + (int64_t)b * p.ldnew + h * 64 : nullptr;
    const bf16_t* vnew = append_at >= 0 ? p.vnew + (int64_t)b * p.ldnew + h * 64 : nullptr;
    if (append_at >= 0) {                         // lane d copies dimension d of the new key and value into the cache
        Kb[(int64_t)append_at * p.kj + lane] = knew[lane];
        Vb[(int64_t)append_at * p.kj + lane] = vnew[lane];
        // ... and the row table learns where it went (own row): steps that are not followed by a re-parenting leave it complete.
        // No wave reads entry [b][append_at] during this step (that key comes from the projection output).
        if (G == 1 && p.rows && lane == 0 && h == 0) p.rows[(int64_t)b * p.ldrows + append_at] = b;
    }
    float qc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bf16x8 q8 = *(const bf16x8*)(p.q + (int64_t)(b0 + g) * p.ldq + h * 64 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qc[g][e] = bf2f((bf16_t)q8[e]) * p.scale;
    }
    // Beam search re-parents sequences every step; instead of copying cache rows (113 us per step at 1,024 rows), key j of
    // sequence b is read from the row of the ancestor that wrote it (rows[b][j], kzv_decode_reorder keeps the table).  This
    // step's own key is still stored to row b above and read from the projection output.
    int roff[G > 1 ? 1 : NI];                      // element offset of the ancestor's row relative to row b (checked < 2^31 on the host)
    if constexpr (G == 1) {
#pragma unroll
        for (int i = 0; i < NI; ++i) roff[i] = 0;
        if (p.rows) {                              // wave-uniform; the loads themselves are unconditional (clamped index) so that
            const int* tr = p.rows + (int64_t)b * p.ldrows;   // they all go out together instead of one round trip each
            int rj[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) rj[i] = tr[min(8 * i + r, (int)p.ldrows - 1)];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = 8 * i + r;
                roff[i] = (j < nkeys && j != append_at) ? (rj[i] - b) * (int)p.kb : 0;
            }
        }
    }
    auto row = [&](const bf16_t* base, const bf16_t* fresh, int i) -> bf16x8 {
        const int j = 8 * i + r;
        if (j >= nkeys) return (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        return *(const bf16x8*)((j == append_at ? fresh : base + (G > 1 ? 0 : roff[G > 1 ? 0 : i]) + (int64_t)j * p.kj) + c * 8);
    };
    // key-usable flags of this lane's keys: loaded up front and unconditionally (clamped index) -- inside the score loop each one
    // sat in its own branch behind an s_waitcnt vmcnt(0), i.e. 24 serial round trips per self-attention call
    bool kok[G][NI];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < NI; ++i) kok[g][i] = true;
    if (p.valid) {
        unsigned char vb[G][NI];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < NI; ++i) vb[g][i] = p.valid[(int64_t)(b0 + g) * p.ldvalid + min(8 * i + r, (int)p.ldvalid - 1)];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int i = 0; i < NI; ++i) kok[g][i] = vb[g][i] != 0;
    }
    float sc[G][NI];
    float mx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) mx[g] = -INFINITY;
    // rows in flight: 8 wave-instructions = 64 keys.  (Requesting ALL key and value rows before the first score -- 225
    // registers -- changed nothing for 1,024 waves and cost the 4,096-wave beam step 6 ms per generation: occupancy.)
    constexpr int CH = 8;
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += CH) {
        if (i0 * 8 >= nkeys) {                     // wave-uniform: nothing left
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int u = 0; u < CH; ++u) sc[g][i0 + u] = -INFINITY;
            continue;
        }
        bf16x8 kk[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) kk[u] = row(Kb, knew, i0 + u);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = 8 * (i0 + u) + r;
            float kf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) kf[e] = bf2f((bf16_t)kk[u][e]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a += qc[g][e] * kf[e];
                a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
                const bool ok = j < nkeys && kok[g][i0 + u];
                sc[g][i0 + u] = ok ? a : -INFINITY;
                mx[g] = fmaxf(mx[g], sc[g][i0 + u]);
            }
        }
    }
    float inv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        mx[g] = wave_max(mx[g]);
        const bool dead = mx[g] == -INFINITY;      // no usable key (a finished, all-pad row): output zeros
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) { sc[g][i] = dead ? 0.f : __expf(sc[g][i] - mx[g]); sum += sc[g][i]; }
        sum = wave_sum(sum) * 0.125f;              // every key is counted by the 8 lanes of its row
        inv[g] = dead ? 0.f : 1.f / sum;
    }
    float o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += CH) {
        if (i0 * 8 >= nkeys) continue;
        bf16x8 vv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) vv[u] = row(Vb, vnew, i0 + u);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            float vf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vf[e] = bf2f((bf16_t)vv[u][e]);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) o[g][e] += sc[g][i0 + u] * vf[e];
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[g][e] += __shfl_xor(o[g][e], 8, 64); o[g][e] += __shfl_xor(o[g][e], 16, 64); o[g][e] += __shfl_xor(o[g][e], 32, 64); }
        if (r == 0) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            const float iv = inv[g];
            const u32x4 pk = (u32x4){pack_bf2(o[g][0] * iv, o[g][1] * iv), pack_bf2(o[g][2] * iv, o[g][3] * iv), pack_bf2(o[g][4] * iv, o[g][5] * iv), pack_bf2(o[g][6] * iv, o[g][7] * iv)};
            *(u32x4*)(p.out + (int64_t)(b0 + g) * p.ldo + h * 64 + c * 8) = pk;
        }
    }
}

// Cross-attention keys / values for the generation steps: [layer][K|V][image][head][key][64] from the projection output
// [image * keys][layer][K|V][head][64] (one GEMM row per patch token).  There a head's key rows are 128-byte pieces 6 KB
// apart (17 us per call at 256 images: 2.5 TB/s); here each (image, head) streams one contiguous block.  Once per generation.
__global__ __launch_bounds__(256) void cross_relayout_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int images, int keys, int heads,
                                                             int layers2, int64_t total16) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total16) return;
    int64_t r = t;                                  // dst order: [l2][img][h][key][8 chunks]
    const int c = r & 7; r >>= 3;
    const int key = r % keys; r /= keys;
    const int h = r % heads; r /= heads;
    const int img = r % images; r /= images;
    const int l2 = (int)r;
    dst[t] = src[(((int64_t)img * keys + key) * layers2 + l2) * heads * 8 + h * 8 + c];
}

// beam re-parenting without moving the cache: dst[b][j] = src[parent[b]][j] for the keys written before this step, and the key
// the parent wrote at this step (index len - 1) sits in the parent's own row.  src == nullptr: the identity table (first step).
__global__ void kv_rows_kernel(const int* __restrict__ src, int* __restrict__ dst, const int64_t* __restrict__ parent, int B, int ld, int len) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= B * len) return;
    const int b = t / len, j = t - b * len;
    const int p = (int)parent[b];
    dst[(int64_t)b * ld + j] = (j == len - 1 || !src) ? p : src[(int64_t)p * ld + j];
}

// One beam-search step's ranking for one image (HF GenerationMixin._get_top_k_continuations, transformers generation/utils.py:
// log_softmax of each beam's next-token logits + the beam's accumulated score, then the K best of the nb * V continuations in
// descending order; equal scores rank by the smaller flat index beam * V + token).  One workgroup per image.  Every thread
// owns E = ceil(V / 256) tokens of each beam row and reads them as ONE batch of independent loads (a scalar loop here exposed a
// memory round trip per element: 58 us per launch): first for the row's (max, sum of exponentials), combined across the
// workgroup; then (cache hits) for the candidates, kept as a sorted top-KM list per thread in registers.  Each wave then
// extracts its K best by shuffles, and one wave merges the four lists.
// Replaces torch's log_softmax + two radix-select top-k passes + gathers (~190 us per step at 256 images) by one launch.
struct Cand { float v; int i; };
__device__ __forceinline__ bool cand_better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

template <int KM, int E>
__global__ __launch_bounds__(256) void beam_topk_kernel(const float* __restrict__ logits, int64_t ld, const float* __restrict__ run_sc, int nb, int V, int K,
                                                        float* __restrict__ out_lp, int64_t* __restrict__ out_ix) {
    constexpr int NBM = 8;
    __shared__ float s_m[4][NBM], s_s[4][NBM];
    __shared__ float s_cv[4][16]; __shared__ int s_ci[4][16];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* base = logits + (int64_t)img * nb * ld;
    // ---- per-row log-sum-exp: local (max, sum) over this thread's E tokens, then across the workgroup ----
    float lm[NBM], ls[NBM];
#pragma unroll
    for (int k = 0; k < NBM; ++k) {
        lm[k] = -INFINITY; ls[k] = 0.f;
        if (k < nb) {
            float x[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { const int v = tid + e * 256; x[e] = base[(int64_t)k * ld + min(v, V - 1)]; if (v >= V) x[e] = -INFINITY; }
#pragma unroll
            for (int e = 0; e < E; ++e) lm[k] = fmaxf(lm[k], x[e]);
            const float wm = wave_max(lm[k]);          // finite: every row has V >= 1 real entries somewhere in the wave? not per wave:
            lm[k] = wm;                                // a wave past the end of a short row holds -inf and contributes exp(-inf) = 0 below
#pragma unroll
            for (int e = 0; e < E; ++e) ls[k] += (x[e] == -INFINITY) ? 0.f : expf(x[e] - wm);
            ls[k] = wave_sum(ls[k]);
        }
    }
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NBM; ++k) { s_m[w][k] = lm[k]; s_s[w][k] = ls[k]; }
    __syncthreads();
    float rmx[NBM], rlse[NBM];
#pragma unroll
    for (int k = 0; k < NBM; ++k) {
        const float M = fmaxf(fmaxf(s_m[0][k], s_m[1][k]), fmaxf(s_m[2][k], s_m[3][k]));
        float S = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) S += (s_m[q][k] == -INFINITY) ? 0.f : s_s[q][k] * expf(s_m[q][k] - M);
        rmx[k] = M; rlse[k] = logf(S);
    }
    // ---- this thread's best KM continuations, best first (flat indices arrive in increasing order: ties keep the smaller) ----
    float tv[KM]; int ti[KM];
#pragma unroll
    for (int q = 0; q < KM; ++q) { tv[q] = -INFINITY; ti[q] = 0x7fffffff; }
#pragma unroll
    for (int k = 0; k < NBM; ++k) {
        if (k < nb) {
            const float sc = run_sc[img * nb + k];
            float x[E];
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = base[(int64_t)k * ld + min(tid + e * 256, V - 1)];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int v = tid + e * 256;
                const float val = ((x[e] - rmx[k]) - rlse[k]) + sc;
                if (v < V && val > tv[KM - 1]) {
                    tv[KM - 1] = val; ti[KM - 1] = k * V + v;
#pragma unroll
                    for (int q = KM - 1; q > 0; --q)
                        if (tv[q] > tv[q - 1]) { const float a = tv[q]; tv[q] = tv[q - 1]; tv[q - 1] = a; const int b = ti[q]; ti[q] = ti[q - 1]; ti[q - 1] = b; }
                }
            }
        }
    }
    // ---- each wave extracts its K best (no barrier), then wave 0 merges the 4 x K ----
    for (int r = 0; r < K; ++r) {
        float bv = tv[0]; int bi = ti[0]; int bl = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64); const int ol = __shfl_xor(bl, o, 64);
            if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; bl = ol; }
        }
        if (lane == 0) { s_cv[w][r] = bv; s_ci[w][r] = bi; }
        if (lane == bl) {
#pragma unroll
            for (int q = 0; q + 1 < KM; ++q) { tv[q] = tv[q + 1]; ti[q] = ti[q + 1]; }
            tv[KM - 1] = -INFINITY; ti[KM - 1] = 0x7fffffff;
        }
    }
    __syncthreads();
    if (w == 0) {
        float cv = -INFINITY; int ci = 0x7fffffff;                 // lane = list (lane >> 4) entry (lane & 15)
        if ((lane & 15) < K) { cv = s_cv[lane >> 4][lane & 15]; ci = s_ci[lane >> 4][lane & 15]; }
        for (int r = 0; r < K; ++r) {
            float bv = cv; int bi = ci; int bl = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64); const int ol = __shfl_xor(bl, o, 64);
                if (cand_better(ov, oi, bv, bi)) { bv = ov; bi = oi; bl = ol; }
            }
            if (lane == 0) { out_lp[(int64_t)img * K + r] = bv; out_ix[(int64_t)img * K + r] = bi; }
            if (lane == bl) { cv = -INFINITY; ci = 0x7fffffff; }
        }
    }
}

// One beam-search step's bookkeeping for one image: what kzv/beam.py::beam_search does between two decoder steps with ~35 small
// torch kernels (~270 us per step), restated from transformers' GenerationMixin._beam_search (generation/utils.py:3208-3508:
// _get_running_beams_for_next_iteration, _update_finished_beams, _check_early_stop_heuristic) -- kzv/beam.py is the readable
// statement and the CPU path pinned against HF; this kernel is checked against it state for state (tests/test_ops_gpu.py).
// One wave per image: lane 0 does the scalar ranking (K <= 16 continuations, nb <= 8 beams), all lanes copy token rows.
struct BeamUpd {
    const float* top_lp; const int64_t* top_ix;                   // [B, K] from kzv_beam_topk
    const int64_t* run_seq_in; int64_t* run_seq_out;              // [B, nb, L]
    const int64_t* fin_seq_in; int64_t* fin_seq_out;              // [B, nb, L]
    float* run_sc; float* fin_sc; unsigned char* fin_done; int64_t* fin_len; unsigned char* unsat;   // [B, nb] / [B]
    int64_t* rows;                                                // [B * nb]: former flat row each running beam continues from
    int* flags;                                                   // [5]: images still unsatisfied, images whose finished list is not full, images with a continuation that did not stop; [3] = stopped, [4] = updates applied
    int nb, K, L, V, cur, eos, early;
    float div_fin, div_open;                                      // (cur + 1 - prompt) ** length_penalty, (cur + 1 + 1 - prompt - 1) ** length_penalty as fp32
};

__global__ __launch_bounds__(64) void beam_update_kernel(const BeamUpd p) {
    constexpr float NEG = -1.0e9f;
    __shared__ int s_src[16], s_tok[16], s_nxt[8], s_mix[8];
    __shared__ unsigned char s_hit[16];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int nb = p.nb, K = p.K, L = p.L;
    if (p.flags[3]) return;                      // the search has ended (beam_stop_kernel): steps issued before the host noticed change nothing
    // the ~40 state words of this image: one load per lane, together (read by lane 0 alone they were as many serial round trips)
    __shared__ float s_lp[16], s_fsc[8];
    __shared__ int64_t s_ix[16], s_flen[8];
    __shared__ unsigned char s_fdone[8];
    __shared__ unsigned char s_unsat;
    if (lane < K) { s_lp[lane] = p.top_lp[(int64_t)img * K + lane]; s_ix[lane] = p.top_ix[(int64_t)img * K + lane]; }
    if (lane >= 16 && lane < 16 + nb) {
        const int i = lane - 16;
        s_fsc[i] = p.fin_sc[img * nb + i]; s_fdone[i] = p.fin_done[img * nb + i]; s_flen[i] = p.fin_len[img * nb + i];
    }
    if (lane == 32) s_unsat = p.unsat[img];
    __syncthreads();
    if (lane == 0) {
        float s2[16], msc[24];
        unsigned char mdone[24];
        bool all_hits = true, all_done = true;
        for (int i = 0; i < nb; ++i) all_done = all_done && s_fdone[i];
        const float full_neg = (all_done && p.early) ? NEG : 0.f;
        const float unsat_neg = s_unsat ? 0.f : NEG;
        for (int k = 0; k < K; ++k) {
            const int64_t ix = s_ix[k];
            const float lp = s_lp[k];
            const int src = (int)(ix / p.V), tok = (int)(ix - (int64_t)src * p.V);
            const bool hit = tok == p.eos || p.cur + 1 >= L;
            s_src[k] = src; s_tok[k] = tok; s_hit[k] = hit;
            all_hits = all_hits && hit;
            s2[k] = lp + (hit ? NEG : 0.f);
            const bool just = hit && k < nb;
            float f = lp / p.div_fin;
            f = f + full_neg; f = f + unsat_neg; f = f + (just ? 0.f : NEG);
            msc[nb + k] = f; mdone[nb + k] = just;
        }
        // running beams of the next step: the best nb continuations that did not stop (equal scores: the smaller rank first)
        unsigned used = 0;
        float run_new[8];
        for (int i = 0; i < nb; ++i) {
            int best = -1;
            for (int k = 0; k < K; ++k) if (!((used >> k) & 1u) && (best < 0 || s2[k] > s2[best])) best = k;
            used |= 1u << best;
            s_nxt[i] = best; run_new[i] = s2[best];
            p.rows[img * nb + i] = (int64_t)img * nb + s_src[best];
        }
        for (int i = 0; i < nb; ++i) p.run_sc[img * nb + i] = run_new[i];
        // finished list: best nb of (old finished, stopped continuations of rank < nb)
        int64_t mlen[24];
        for (int i = 0; i < nb; ++i) { msc[i] = s_fsc[i]; mdone[i] = s_fdone[i]; mlen[i] = s_flen[i]; }
        for (int k = 0; k < K; ++k) mlen[nb + k] = p.cur + 1;
        used = 0;
        float fsc_new[8]; unsigned char fd_new[8]; int64_t fl_new[8];
        for (int i = 0; i < nb; ++i) {
            int best = -1;
            for (int k = 0; k < nb + K; ++k) if (!((used >> k) & 1u) && (best < 0 || msc[k] > msc[best])) best = k;
            used |= 1u << best;
            s_mix[i] = best; fsc_new[i] = msc[best]; fd_new[i] = mdone[best]; fl_new[i] = mlen[best];
        }
        float fmin = fsc_new[0];
        bool done_new = true;
        for (int i = 0; i < nb; ++i) {
            p.fin_sc[img * nb + i] = fsc_new[i]; p.fin_done[img * nb + i] = fd_new[i]; p.fin_len[img * nb + i] = fl_new[i];
            fmin = fminf(fmin, fsc_new[i]); done_new = done_new && fd_new[i];
        }
        // can the best open beam still beat the worst finished one?
        const float best_open = run_new[0] / p.div_open;
        bool any = false;
        for (int i = 0; i < nb; ++i) any = any || best_open > (fd_new[i] ? fmin : NEG);
        const bool un = s_unsat && any;
        p.unsat[img] = un;
        if (un) atomicAdd(p.flags + 0, 1);
        if (!done_new) atomicAdd(p.flags + 1, 1);
        if (!all_hits) atomicAdd(p.flags + 2, 1);
    }
    __syncthreads();
    // token rows: next running beams = continuation s_nxt[i]; finished rows = old finished row or a continuation
    for (int i = 0; i < nb; ++i) {
        const int k = s_nxt[i];
        const int64_t* src = p.run_seq_in + ((int64_t)img * nb + s_src[k]) * L;
        int64_t* dst = p.run_seq_out + ((int64_t)img * nb + i) * L;
        for (int j = lane; j < L; j += 64) dst[j] = j == p.cur ? (int64_t)s_tok[k] : src[j];
        const int mi = s_mix[i];
        int64_t* fdst = p.fin_seq_out + ((int64_t)img * nb + i) * L;
        if (mi < nb) {
            const int64_t* fsrc = p.fin_seq_in + ((int64_t)img * nb + mi) * L;
            for (int j = lane; j < L; j += 64) fdst[j] = fsrc[j];
        } else {
            const int kk = mi - nb;
            const int64_t* csrc = p.run_seq_in + ((int64_t)img * nb + s_src[kk]) * L;
            for (int j = lane; j < L; j += 64) fdst[j] = j == p.cur ? (int64_t)s_tok[kk] : csrc[j];
        }
    }
}

// inputs of decoder step t from the token table: newest token, its RoBERTa position id (HF modeling_roberta.py:142-155: t + 1 +
// pad_id for a real token, pad_id for padding) and the key-usable flag of column t -- six small torch kernels otherwise
__global__ void decode_prep_kernel(const int64_t* __restrict__ ids, int64_t ld_ids, int t, int pad, int B, int64_t* __restrict__ tok,
                                   unsigned char* __restrict__ valid, int64_t ld_valid, int* __restrict__ posids) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t v = ids[(int64_t)b * ld_ids + t];
    const bool live = v != pad;
    tok[b] = v;
    valid[(int64_t)b * ld_valid + t] = live ? 1 : 0;
    posids[b] = live ? t + 1 + pad : pad;
}

// greedy token selection of one sequence (the num_beams = 1 branch of HF _sample: argmax, padding once the sequence has ended):
// the first index of the row maximum, like torch.argmax
__global__ __launch_bounds__(256) void greedy_update_kernel(const float* __restrict__ logits, int64_t ld, int V, int64_t* __restrict__ ids, int64_t ld_ids,
                                                            int t, unsigned char* __restrict__ done, int pad, int eos, int* __restrict__ flags) {
    __shared__ float s_v[4]; __shared__ int s_i[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* row = logits + (int64_t)b * ld;
    constexpr int E = 8;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int v0 = tid; v0 < V; v0 += 256 * E) {
        float x[E];
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = row[min(v0 + e * 256, V - 1)];
#pragma unroll
        for (int e = 0; e < E; ++e) { const int v = v0 + e * 256; if (v < V && x[e] > bv) { bv = x[e]; bi = v; } }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { s_v[w] = bv; s_i[w] = bi; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 1; q < 4; ++q) if (s_v[q] > bv || (s_v[q] == bv && s_i[q] < bi)) { bv = s_v[q]; bi = s_i[q]; }
        const bool was = done[b] != 0;
        const int nxt = was ? pad : bi;
        ids[(int64_t)b * ld_ids + t + 1] = nxt;
        const bool now = was || nxt == eos;
        done[b] = now ? 1 : 0;
        if (!now) atomicAdd(flags + t, 1);
    }
}

// after beam_update: the loop condition of HF _beam_search, kept on the device so that the host may look only every few steps
__global__ void beam_stop_kernel(int* flags, int early, int cur) {
    if (flags[3]) return;
    flags[4] = cur;                              // updates applied so far (cur counts from 1)
    const bool go_on = flags[0] > 0 && !(early && flags[1] == 0) && flags[2] > 0;
    if (!go_on) flags[3] = 1;
}

__global__ void step_inc_kernel(int* t) { *t += 1; }

}  // namespace

int kzv_step_inc(int* d_t, hipStream_t s) {
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, d_t);
    return kzv_check_launch("step_inc");
}

// tptr != nullptr: self-attention of graph-replayed step `*tptr` (nkeys = the cache capacity, which picks the kernel)
int kzv_attn_decode(const bf16_t* q, int64_t ldq, const bf16_t* knew, const bf16_t* vnew, int64_t ldnew, bf16_t* K, bf16_t* V, int64_t kb,
                    int64_t kj, const unsigned char* valid, int64_t ldvalid, bf16_t* out, int64_t ldo, int B, int heads, int nkeys,
                    int append_at, hipStream_t s, const int* tptr, int group, int* rows, int64_t ldrows, int64_t kh) {
    if (nkeys < 1 || nkeys > 320) return kzv_fail(KZV_E_ARG, "attn_decode: 1..320 keys");
    if (ldq % 8 || ldo % 8 || (knew && ldnew % 8)) return kzv_fail(KZV_E_ARG, "attn_decode: rows must be 16-byte aligned");
    if (group < 1 || (append_at >= 0 && group != 1)) return kzv_fail(KZV_E_ARG, "attn_decode: shared keys cannot be appended to");
    if (kj % 8) return kzv_fail(KZV_E_ARG, "attn_decode: key rows must be 16-byte aligned");
    if (rows && group != 1) return kzv_fail(KZV_E_ARG, "attn_decode: a row table and shared keys exclude each other");
    if (rows && (int64_t)B * kb >= (1ll << 31)) return kzv_fail(KZV_E_ARG, "attn_decode: cache too large for 32-bit row offsets");
    if (kh % 8) return kzv_fail(KZV_E_ARG, "attn_decode: head stride must be 16-byte aligned");
    DecAttnP p{q, ldq, knew, vnew, ldnew, K, V, kb, kj, kh, valid, ldvalid, out, ldo, nkeys, append_at, heads, 0.125f, tptr, group, rows, ldrows};
    // beams of one image (group > 1, shared keys): G of them per wave, G = the largest of 4, 3, 2 dividing the group
    const int G = group % 4 == 0 ? 4 : group % 3 == 0 ? 3 : group % 2 == 0 ? 2 : 1;
    const int nunits = (B / G) * heads;
#define KZV_AD(NI, GG) hipLaunchKernelGGL((attn_decode_kernel<NI, GG>), dim3((nunits + 3) / 4), dim3(256), 0, s, p, nunits)
#define KZV_AD_G(NI) do { if (G == 4) KZV_AD(NI, 4); else if (G == 3) KZV_AD(NI, 3); else if (G == 2) KZV_AD(NI, 2); else KZV_AD(NI, 1); } while (0)
    if (nkeys <= 192) KZV_AD_G(24); else KZV_AD_G(40);
#undef KZV_AD_G
#undef KZV_AD
    return kzv_check_launch("attn_decode");
}

int kzv_kv_rows(const int* src, int* dst, const int64_t* parent, int B, int ld, int len, hipStream_t s) {
    hipLaunchKernelGGL(kv_rows_kernel, dim3((B * len + 255) / 256), dim3(256), 0, s, src, dst, parent, B, ld, len);
    return kzv_check_launch("kv_rows");
}

int kzv_cross_relayout(const bf16_t* src, bf16_t* dst, int images, int keys, int heads, int layers2, hipStream_t s) {
    const int64_t total16 = (int64_t)layers2 * images * heads * keys * 8;
    hipLaunchKernelGGL(cross_relayout_kernel, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, images, keys, heads,
                       layers2, total16);
    return kzv_check_launch("cross_relayout");
}

extern "C" int kzv_beam_topk(const float* d_logits, int64_t ld, const float* d_beam_scores, int batch, int num_beams, int vocab, int k,
                             float* d_top_scores, int64_t* d_top_index, void* stream) {
    if (!d_logits || !d_beam_scores || !d_top_scores || !d_top_index) return kzv_fail(KZV_E_ARG, "beam_topk: null operand");
    if (batch < 1 || num_beams < 1 || num_beams > 8 || vocab < 1 || ld < vocab) return kzv_fail(KZV_E_ARG, "beam_topk: 1..8 beams, ld >= vocab");
    if (k < 1 || k > 16 || (int64_t)k > (int64_t)num_beams * vocab) return kzv_fail(KZV_E_ARG, "beam_topk: 1..16 continuations, at most beams * vocab");
    if ((int64_t)num_beams * vocab >= (1ll << 31)) return kzv_fail(KZV_E_ARG, "beam_topk: beams * vocab beyond 32-bit flat indices");
    if (vocab > 64 * 256) return kzv_fail(KZV_E_ARG, "beam_topk: vocab beyond 16,384");
    hipStream_t s = (hipStream_t)stream;
#define KZV_BT(KM, E) hipLaunchKernelGGL((beam_topk_kernel<KM, E>), dim3(batch), dim3(256), 0, s, d_logits, ld, d_beam_scores, num_beams, vocab, k, d_top_scores, d_top_index)
#define KZV_BT_E(KM) do { if (vocab <= 4 * 256) KZV_BT(KM, 4); else if (vocab <= 20 * 256) KZV_BT(KM, 20); else KZV_BT(KM, 64); } while (0)
    if (k <= 4) KZV_BT_E(4);
    else if (k <= 8) KZV_BT_E(8);
    else KZV_BT_E(16);
#undef KZV_BT_E
#undef KZV_BT
    return kzv_check_launch("beam_topk");
}

extern "C" int kzv_beam_update(const kzv_beam_state* st, const float* d_top_scores, const int64_t* d_top_index, int cur, int early_stopping,
                               float length_penalty, int64_t* d_rows, int* d_flags, void* stream) {
    if (!st || !d_top_scores || !d_top_index || !d_rows || !d_flags) return kzv_fail(KZV_E_ARG, "beam_update: null operand");
    if (!st->run_seq_in || !st->run_seq_out || !st->fin_seq_in || !st->fin_seq_out || !st->run_scores || !st->fin_scores || !st->fin_done ||
        !st->fin_len || !st->unsatisfied) return kzv_fail(KZV_E_ARG, "beam_update: null state array");
    if (st->batch < 1 || st->num_beams < 1 || st->num_beams > 8 || st->max_len < 2 || st->vocab < 1) return kzv_fail(KZV_E_ARG, "beam_update: 1..8 beams");
    if (cur < 1 || cur >= st->max_len) return kzv_fail(KZV_E_ARG, "beam_update: position outside 1..max_len-1");
    if (st->run_seq_in == st->run_seq_out || st->fin_seq_in == st->fin_seq_out) return kzv_fail(KZV_E_ARG, "beam_update: in and out token rows must differ");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_flags, 0, (cur == 1 ? 5 : 3) * sizeof(int), s) != hipSuccess) return kzv_fail(KZV_E_HIP, "beam_update: memset");
    BeamUpd p{d_top_scores, d_top_index, st->run_seq_in, st->run_seq_out, st->fin_seq_in, st->fin_seq_out, st->run_scores, st->fin_scores,
              st->fin_done, st->fin_len, st->unsatisfied, d_rows, d_flags, st->num_beams, 2 * st->num_beams, st->max_len, st->vocab, cur,
              st->eos_id, early_stopping ? 1 : 0,
              (float)pow((double)(cur + 1 - 1), (double)length_penalty), (float)pow((double)(cur + 1 - 1), (double)length_penalty)};
    hipLaunchKernelGGL(beam_update_kernel, dim3(st->batch), dim3(64), 0, s, p);
    hipLaunchKernelGGL(beam_stop_kernel, dim3(1), dim3(1), 0, s, d_flags, early_stopping ? 1 : 0, cur);
    return kzv_check_launch("beam_update");
}

extern "C" int kzv_decode_prep(const int64_t* d_ids, int64_t ld_ids, int t, int pad_id, int batch, int64_t* d_tokens, uint8_t* d_valid, int64_t ld_valid,
                               int32_t* d_posids, void* stream) {
    if (!d_ids || !d_tokens || !d_valid || !d_posids || batch < 1 || t < 0 || t >= ld_ids || t >= ld_valid) return kzv_fail(KZV_E_ARG, "decode_prep: bad arguments");
    hipLaunchKernelGGL(decode_prep_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_ids, ld_ids, t, pad_id, batch, d_tokens, d_valid, ld_valid, d_posids);
    return kzv_check_launch("decode_prep");
}

extern "C" int kzv_greedy_update(const float* d_logits, int64_t ld, int vocab, int64_t* d_ids, int64_t ld_ids, int t, uint8_t* d_done, int batch,
                                 int pad_id, int eos_id, int32_t* d_flags, void* stream) {
    if (!d_logits || !d_ids || !d_done || !d_flags || batch < 1 || vocab < 1 || ld < vocab || t < 0 || t + 1 >= ld_ids) return kzv_fail(KZV_E_ARG, "greedy_update: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(greedy_update_kernel, dim3(batch), dim3(256), 0, s, d_logits, ld, vocab, d_ids, ld_ids, t, d_done, pad_id, eos_id, d_flags);
    return kzv_check_launch("greedy_update");
}
