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

namespace {

struct DecAttnP {
    const bf16_t* q; int64_t ldq;                 // [B, ldq], head h at column h*64
    const bf16_t* knew; const bf16_t* vnew; int64_t ldnew;   // this step's key/value rows [B, ldnew] (self-attention) or null
    bf16_t* K; bf16_t* V; int64_t kb, kj;         // key j of sequence b: K + b*kb + j*kj + h*64 (same strides for V)
    const unsigned char* valid; int64_t ldvalid;  // [B, ldvalid]: key j usable (null: all)
    bf16_t* out; int64_t ldo;
    int nkeys, append_at, heads;                  // append_at >= 0: write knew/vnew at key index append_at first
    float scale;
    const int* tptr;                              // non-null: the step index t lives in device memory (graph replay): nkeys = t + 1, append_at = t
    int group;                                    // keys / values of sequence b live at batch index b / group (beams sharing one image's cross-attention K/V)
};

// One WAVE per (sequence, head), no LDS and no barrier.  Every global access is a wave-instruction over 8 key (or value) rows x
// 128 contiguous bytes: lane = (row r = lane >> 3, 16-byte chunk c = lane & 7), iteration i owns key 8 i + r.  Scores: each
// lane multiplies its chunk of the key by the matching 8 query dimensions (registers), three shuffles sum the 8 chunks, so
// the 8 lanes of a row all hold that key's score; max and sum run across the wave.  Output: the SAME lanes hold the matching
// chunk of the value row, weight it by their key's probability, and three shuffles sum the 8 rows of an iteration.
// The step's own key / value (self-attention) is taken from the projection output directly and stored to the cache on the
// side.  NI = iterations (8 keys each): 24 -> 192 keys, 40 -> 320 keys.
template <int NI>
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecAttnP p, const int npairs) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);          // (sequence, head) of this wave
    if (pair >= npairs) return;
    const int b = pair / p.heads, h = pair - b * p.heads, lane = threadIdx.x & 63;
    bf16_t* Kb = p.K + (int64_t)(b / p.group) * p.kb + h * 64;
    bf16_t* Vb = p.V + (int64_t)(b / p.group) * p.kb + h * 64;
    const int tdev = p.tptr ? *p.tptr : 0;
    const int nkeys = p.tptr ? min(tdev + 1, 8 * NI) : p.nkeys;
    const int append_at = p.tptr ? min(tdev, 8 * NI - 1) : p.append_at;
    const int r = lane >> 3, c = lane & 7;
    const bf16_t* knew = append_at >= 0 ? p.knew + (int64_t)b * p.ldnew + h * 64 : nullptr;
    const bf16_t* vnew = append_at >= 0 ? p.vnew + (int64_t)b * p.ldnew + h * 64 : nullptr;
    if (append_at >= 0) {                         // lane d copies dimension d of the new key and value into the cache
        Kb[(int64_t)append_at * p.kj + lane] = knew[lane];
        Vb[(int64_t)append_at * p.kj + lane] = vnew[lane];
    }
    float qc[8];
    {
        const bf16x8 q8 = *(const bf16x8*)(p.q + (int64_t)b * p.ldq + h * 64 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qc[e] = bf2f((bf16_t)q8[e]) * p.scale;
    }
    auto row = [&](const bf16_t* base, const bf16_t* fresh, int j) -> bf16x8 {
        if (j >= nkeys) return (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        return *(const bf16x8*)((j == append_at ? fresh : base + (int64_t)j * p.kj) + c * 8);
    };
    float sc[NI];
    float mx = -INFINITY;
    constexpr int CH = 8;                          // rows in flight: 8 wave-instructions = 64 keys
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += CH) {
        if (i0 * 8 >= nkeys) {                     // wave-uniform: nothing left
#pragma unroll
            for (int u = 0; u < CH; ++u) sc[i0 + u] = -INFINITY;
            continue;
        }
        bf16x8 kk[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) kk[u] = row(Kb, knew, 8 * (i0 + u) + r);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = 8 * (i0 + u) + r;
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a += qc[e] * bf2f((bf16_t)kk[u][e]);
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
            const bool ok = j < nkeys && (!p.valid || p.valid[(int64_t)b * p.ldvalid + j]);
            sc[i0 + u] = ok ? a : -INFINITY;
            mx = fmaxf(mx, sc[i0 + u]);
        }
    }
    mx = wave_max(mx);
    const bool dead = mx == -INFINITY;             // no usable key (a finished, all-pad row): output zeros
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) { sc[i] = dead ? 0.f : __expf(sc[i] - mx); sum += sc[i]; }
    sum = wave_sum(sum) * 0.125f;                  // every key is counted by the 8 lanes of its row
    const float inv = dead ? 0.f : 1.f / sum;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += CH) {
        if (i0 * 8 >= nkeys) continue;
        bf16x8 vv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) vv[u] = row(Vb, vnew, 8 * (i0 + u) + r);
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += sc[i0 + u] * bf2f((bf16_t)vv[u][e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 8, 64); o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
    if (r == 0) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 pk = (u32x4){pack_bf2(o[0] * inv, o[1] * inv), pack_bf2(o[2] * inv, o[3] * inv), pack_bf2(o[4] * inv, o[5] * inv), pack_bf2(o[6] * inv, o[7] * inv)};
        *(u32x4*)(p.out + (int64_t)b * p.ldo + h * 64 + c * 8) = pk;
    }
}

// beam re-ordering: dst[l][b][j][:] = src[l][idx[b]][j][:] for j < len (16-byte chunks)
__global__ __launch_bounds__(256) void kv_reorder_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int64_t* __restrict__ idx,
                                                         int B, int Tmax, int len, int chunks_per_row, int64_t layer_stride16) {
    const int64_t per_b = (int64_t)len * chunks_per_row;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)B * per_b) return;
    const int b = (int)(t / per_b);
    const int64_t r = t - (int64_t)b * per_b;
    const int64_t lo = (int64_t)blockIdx.y * layer_stride16;
    dst[lo + (int64_t)b * Tmax * chunks_per_row + r] = src[lo + idx[b] * (int64_t)Tmax * chunks_per_row + r];
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
                    int append_at, hipStream_t s, const int* tptr, int group) {
    if (nkeys < 1 || nkeys > 320) return kzv_fail(KZV_E_ARG, "attn_decode: 1..320 keys");
    if (ldq % 8 || ldo % 8 || (knew && ldnew % 8)) return kzv_fail(KZV_E_ARG, "attn_decode: rows must be 16-byte aligned");
    if (group < 1 || (append_at >= 0 && group != 1)) return kzv_fail(KZV_E_ARG, "attn_decode: shared keys cannot be appended to");
    if (kj % 8) return kzv_fail(KZV_E_ARG, "attn_decode: key rows must be 16-byte aligned");
    DecAttnP p{q, ldq, knew, vnew, ldnew, K, V, kb, kj, valid, ldvalid, out, ldo, nkeys, append_at, heads, 0.125f, tptr, group};
    const int npairs = B * heads;
    if (nkeys <= 192) hipLaunchKernelGGL(attn_decode_kernel<24>, dim3((npairs + 3) / 4), dim3(256), 0, s, p, npairs);
    else hipLaunchKernelGGL(attn_decode_kernel<40>, dim3((npairs + 3) / 4), dim3(256), 0, s, p, npairs);
    return kzv_check_launch("attn_decode");
}

int kzv_kv_reorder(const bf16_t* src, bf16_t* dst, const int64_t* idx, int layers2, int B, int Tmax, int len, int Hd, hipStream_t s) {
    const int cpr = Hd / 8;                       // 16-byte chunks per cache row
    const int64_t total = (int64_t)B * len * cpr;
    hipLaunchKernelGGL(kv_reorder_kernel, dim3((unsigned)((total + 255) / 256), layers2), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, idx,
                       B, Tmax, len, cpr, (int64_t)B * Tmax * cpr);
    return kzv_check_launch("kv_reorder");
}
