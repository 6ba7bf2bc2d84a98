// N1: KV-cached decoder step: one new token per sequence attends over cached keys / values.
//
// Replaces, for generation only, the per-step work of HF RobertaSelfAttention / RobertaCrossAttention with `use_cache=True`
// (modeling_roberta.py:186-326) as driven by `decoder.generate` (src/models/trocr_model.py:306-316).  One workgroup per
// (sequence, head), fp32 scores and softmax (see attn_decode_kernel); the self-attention variant also appends this step's key
// and value to the cache.  The reference's default 1024x64 columns give 256 cross-attention keys.
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

// One 256-thread workgroup per (sequence, head).  Every global access is a wave-instruction over 8 key (or value) rows x
// 128 contiguous bytes: lane = (row r = lane >> 3, 16-byte chunk c = lane & 7), wave w of iteration i owns key 32 i + 8 w + r.
// Phase 1: each lane multiplies its chunk of the key by the matching 8 query dimensions (held in registers), three shuffles
// sum the 8 chunks of a key, the scores meet in LDS and are soft-maxed by one thread per key.  Phase 2: the same lanes weight
// their chunk of the value row (requested up front, next to the keys: one memory round trip for both), three shuffles sum
// the 8 rows of a wave-instruction, the four waves' partial outputs meet in LDS.  NU: 256 NU keys at most.
template <int NU>
__global__ __launch_bounds__(256) void attn_decode_kernel(const DecAttnP p) {
    __shared__ float prob[256 * NU];
    __shared__ float red[8];
    __shared__ float part[4][64];
    const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    bf16_t* Kb = p.K + (int64_t)(b / p.group) * p.kb + h * 64;
    bf16_t* Vb = p.V + (int64_t)(b / p.group) * p.kb + h * 64;
    const int tdev = p.tptr ? *p.tptr : 0;
    const int nkeys = p.tptr ? min(tdev + 1, 256 * NU) : p.nkeys;
    const int append_at = p.tptr ? min(tdev, 256 * NU - 1) : p.append_at;
    if (append_at >= 0) {                         // thread d copies dimension d of the new key and value into the cache
        if (tid < 64) {
            Kb[(int64_t)append_at * p.kj + tid] = p.knew[(int64_t)b * p.ldnew + h * 64 + tid];
            Vb[(int64_t)append_at * p.kj + tid] = p.vnew[(int64_t)b * p.ldnew + h * 64 + tid];
        }
        __syncthreads();                          // the appended row is read back below by other waves of this workgroup
    }
    const int r = lane >> 3, c = lane & 7;
    constexpr int NI = 8 * NU;
    bf16x8 kk[NI], vv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = 32 * i + 8 * w + r;
        const bool in = j < nkeys;
        kk[i] = in ? *(const bf16x8*)(Kb + (int64_t)j * p.kj + c * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        vv[i] = in ? *(const bf16x8*)(Vb + (int64_t)j * p.kj + c * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    float qc[8];
    {
        const bf16x8 q8 = *(const bf16x8*)(p.q + (int64_t)b * p.ldq + h * 64 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qc[e] = bf2f((bf16_t)q8[e]) * p.scale;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int j = 32 * i + 8 * w + r;
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a += qc[e] * bf2f((bf16_t)kk[i][e]);
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
        if (c == 0) prob[j] = (j < nkeys && (!p.valid || p.valid[(int64_t)b * p.ldvalid + j])) ? a : -INFINITY;
    }
    __syncthreads();
    float sc[NU];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NU; ++u) { sc[u] = prob[tid + 256 * u]; mx = fmaxf(mx, sc[u]); }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const bool dead = mx == -INFINITY;            // no usable key (a finished, all-pad row): output zeros
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) { sc[u] = dead ? 0.f : __expf(sc[u] - mx); sum += sc[u]; prob[tid + 256 * u] = sc[u]; }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + w] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    const float inv = dead ? 0.f : 1.f / sum;
    // phase 2: out[d] = inv * sum_j prob[j] * V[j][d]
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float pj = prob[32 * i + 8 * w + r];           // 0 for keys beyond nkeys (their scores were -inf)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += pj * bf2f((bf16_t)vv[i][e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 8, 64); o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[w][c * 8 + e] = o[e];
    }
    __syncthreads();
    if (tid < 64) p.out[(int64_t)b * p.ldo + h * 64 + tid] = f2bf((part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid]) * inv);
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
    if (nkeys < 1 || nkeys > 512) return kzv_fail(KZV_E_ARG, "attn_decode: 1..512 keys");
    if (group < 1 || (append_at >= 0 && group != 1)) return kzv_fail(KZV_E_ARG, "attn_decode: shared keys cannot be appended to");
    if (kj % 8) return kzv_fail(KZV_E_ARG, "attn_decode: key rows must be 16-byte aligned");
    DecAttnP p{q, ldq, knew, vnew, ldnew, K, V, kb, kj, valid, ldvalid, out, ldo, nkeys, append_at, heads, 0.125f, tptr, group};
    if (nkeys <= 256) hipLaunchKernelGGL(attn_decode_kernel<1>, dim3(B * heads), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_decode_kernel<2>, dim3(B * heads), dim3(256), 0, s, p);
    return kzv_check_launch("attn_decode");
}

int kzv_kv_reorder(const bf16_t* src, bf16_t* dst, const int64_t* idx, int layers2, int B, int Tmax, int len, int Hd, hipStream_t s) {
    const int cpr = Hd / 8;                       // 16-byte chunks per cache row
    const int64_t total = (int64_t)B * len * cpr;
    hipLaunchKernelGGL(kv_reorder_kernel, dim3((unsigned)((total + 255) / 256), layers2), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, idx,
                       B, Tmax, len, cpr, (int64_t)B * Tmax * cpr);
    return kzv_check_launch("kv_reorder");
}
