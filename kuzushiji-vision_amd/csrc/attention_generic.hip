// Multi-head attention forward / backward for head dimensions OTHER than 64 (8 <= D <= 128, D % 8 == 0), unmasked (ViT).
//
// Why it exists: the reference's CLI defaults give the ViT encoder hidden 768 with 8 heads, i.e. head_dim 96
// (scripts/train_trocr.py:41-43), while the MFMA kernels of attention.hip are built around 64-wide heads (two 32-deep MFMA
// steps, 128-byte LDS rows).  This file is the FUNCTIONAL path for such geometries: plain fp32 VALU arithmetic on LDS-staged
// K / V, same semantics and the same dropout element indexing as attention.hip (HF eager_attention_forward,
// modeling_vit.py:164-189: fp32 softmax of q.k^T * D^-0.5, probability dropout, P.V), several times slower than the MFMA path
// and not part of any benchmark.  The decoder (4 heads of 64) never comes here.
//
//   forward   one workgroup per (batch, head, 16 queries); K and V of the head in LDS (rows padded to D + 2 elements: an odd
//             dword stride, so "lane = key" reads are conflict-free); a wave owns 4 queries, lane = key for the scores and the
//             softmax (wave reductions), lane = output dimension for P.V.
//   backward  kernel A, same decomposition: recomputes P from the saved log-sum-exp, dP = dO.V^T, dS = P (dP keep - delta);
//             writes dQ and parks dS * scale and P * keep (bf16) in a scratch matrix [pair][Sq][Sk_even];
//             kernel B, one workgroup per (batch, head, 16 keys): dK = dS^T.Q, dV = (P keep)^T.dO, lane = dimension.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include <mutex>

namespace {

struct GenP {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; float* LSE;
    const bf16_t* dO; bf16_t* dQ; bf16_t* dK; bf16_t* dV;
    bf16_t* dSs; bf16_t* Pd;                 // backward scratch [B*heads][Sq][Sk_even]
    int64_t ldq, ldk, ldv, ldo;
    int B, heads, Sq, Sk, D;
    float scale; unsigned thr16; float inv_keep; unsigned key;
};

constexpr int QT = 16;       // queries (or keys, kernel B) per workgroup

__device__ __forceinline__ float keep_mul(const GenP& p, int pair, int q, int k) {
    if (!p.thr16) return 1.f;
    // the attention sites' 4 x 4-block generator (kzv_common.h), scalar form: same masks as attention.hip would draw
    const unsigned block = ((unsigned)pair * ((unsigned)(p.Sq + 3) >> 2) + ((unsigned)q >> 2)) * ((unsigned)(p.Sk + 3) >> 2) + ((unsigned)k >> 2);
    return att_keep1(p.key, block, q & 3, k & 3, (int)p.thr16) ? p.inv_keep : 0.f;
}

// stage rows [0, n) of a [n, D] bf16 matrix (row stride ld) into LDS rows of stride DP = D + 2 elements
__device__ __forceinline__ void stage_rows(bf16_t* dst, const bf16_t* src, int64_t ld, int n, int D, int DP) {
    const int c8 = D >> 3;
    for (int i = threadIdx.x; i < n * c8; i += blockDim.x) {
        const int r = i / c8, c = i - r * c8;
        const uint4 v = *(const uint4*)(src + (int64_t)r * ld + c * 8);
        unsigned* d = (unsigned*)(dst + r * DP + c * 8);       // (D + 2) * 2 bytes is a multiple of 4, c * 16 too
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__device__ __forceinline__ float dot_row(const float* q, const bf16_t* row, int D) {     // q in LDS (broadcast), row = this lane's key
    float a = 0.f;
    for (int d = 0; d < D; d += 2) {
        const unsigned u = *(const unsigned*)(row + d);
        a += q[d] * bf2f((bf16_t)(u & 0xffffu)) + q[d + 1] * bf2f((bf16_t)(u >> 16));
    }
    return a;
}

template <bool BWD>
__global__ __launch_bounds__(256) void attn_gen_kernel(const GenP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = p.D, DP = D + 2, Sk = p.Sk;
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + (size_t)Sk * DP;
    float* wbuf = (float*)(Vs + (size_t)Sk * DP + ((size_t)Sk * DP & 1));     // 4-byte aligned: per wave [qf D][of D][pr Sk_pad]
    const int SkP = (Sk + 63) & ~63;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float* qf = wbuf + (size_t)w * (2 * D + SkP);
    float* of = qf + D;                     // backward: dO row
    float* pr = of + D;                     // probabilities (forward) / dS (backward)
    const int pair = blockIdx.x, b = pair / p.heads, h = pair - b * p.heads;
    stage_rows(Ks, p.K + (int64_t)b * Sk * p.ldk + h * D, p.ldk, Sk, D, DP);
    stage_rows(Vs, p.V + (int64_t)b * Sk * p.ldv + h * D, p.ldv, Sk, D, DP);
    __syncthreads();
    for (int i = 0; i < QT / 4; ++i) {
        const int q = blockIdx.y * QT + w * (QT / 4) + i;
        if (q >= p.Sq) break;                                   // wave-uniform
        const int64_t qrow = (int64_t)b * p.Sq + q;
        for (int d = lane; d < D; d += 64) {
            qf[d] = bf2f(p.Q[qrow * p.ldq + h * D + d]) * p.scale;
            if (BWD) of[d] = bf2f(p.dO[qrow * p.ldo + h * D + d]);
        }
        __builtin_amdgcn_wave_barrier();                         // qf / of are read below by every lane of this wave (LDS is in order per wave)
        float delta = 0.f;
        if (BWD) {
            for (int d = lane; d < D; d += 64) delta += bf2f(p.dO[qrow * p.ldo + h * D + d]) * bf2f(p.O[qrow * p.ldo + h * D + d]);
            delta = wave_sum(delta);
        }
        // scores of this lane's keys (k = lane + 64 j)
        float s[8];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = lane + 64 * j;
            s[j] = (k < Sk) ? dot_row(qf, Ks + k * DP, D) : -INFINITY;
            mx = fmaxf(mx, s[j]);
        }
        if (!BWD) {
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[j] = (lane + 64 * j < Sk) ? __expf(s[j] - mx) : 0.f; sum += s[j]; }
            sum = wave_sum(sum);
            const float inv = 1.f / sum;
            if (lane == 0 && p.LSE) p.LSE[(int64_t)pair * p.Sq + q] = mx + __logf(sum);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = lane + 64 * j;
                if (k < SkP) pr[k] = (k < Sk) ? bf2f(f2bf(s[j] * inv * keep_mul(p, pair, q, k))) : 0.f;   // bf16-rounded like the MFMA operand
            }
            __builtin_amdgcn_wave_barrier();
            // O[q][d] = sum_k pr[k] V[k][d]; lane = d (and d + 64)
            for (int d = lane; d < D; d += 64) {
                float a = 0.f;
                for (int k = 0; k < Sk; ++k) a += pr[k] * bf2f(Vs[k * DP + d]);
                p.O[qrow * p.ldo + h * D + d] = f2bf(a);
            }
        } else {
            const float lse = p.LSE[(int64_t)pair * p.Sq + q];
            bf16_t* dsr = p.dSs + ((int64_t)pair * p.Sq + q) * ((Sk + 1) & ~1);
            bf16_t* pdr = p.Pd + ((int64_t)pair * p.Sq + q) * ((Sk + 1) & ~1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = lane + 64 * j;
                float ds = 0.f;
                if (k < Sk) {
                    const float pk = __expf(s[j] - lse);
                    const float kp = keep_mul(p, pair, q, k);
                    const float dP = dot_row(of, Vs + k * DP, D);
                    ds = pk * (dP * kp - delta) * p.scale;
                    dsr[k] = f2bf(ds);
                    pdr[k] = f2bf(pk * kp);
                    ds = bf2f(f2bf(ds));
                }
                if (k < SkP) pr[k] = ds;
            }
            __builtin_amdgcn_wave_barrier();
            // dQ[q][d] = sum_k dS[k] K[k][d]
            for (int d = lane; d < D; d += 64) {
                float a = 0.f;
                for (int k = 0; k < Sk; ++k) a += pr[k] * bf2f(Ks[k * DP + d]);
                p.dQ[qrow * p.ldq + h * D + d] = f2bf(a);
            }
        }
        __builtin_amdgcn_wave_barrier();                         // the next query overwrites qf / of / pr
    }
}

// dK[k][d] = sum_q dSs[q][k] Q[q][d]; dV[k][d] = sum_q Pd[q][k] dO[q][d].  Workgroup = (pair, 16 keys), wave = 4 keys, lane = d.
__global__ __launch_bounds__(256) void attn_gen_dkv_kernel(const GenP p) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int pair = blockIdx.x, b = pair / p.heads, h = pair - b * p.heads;
    const int D = p.D, SkE = (p.Sk + 1) & ~1;
    for (int i = 0; i < QT / 4; ++i) {
        const int k = blockIdx.y * QT + w * (QT / 4) + i;
        if (k >= p.Sk) break;
        float ak[2] = {0.f, 0.f}, av[2] = {0.f, 0.f};
        for (int q = 0; q < p.Sq; ++q) {
            const int64_t qrow = (int64_t)b * p.Sq + q;
            const float ds = bf2f(p.dSs[((int64_t)pair * p.Sq + q) * SkE + k]);
            const float pd = bf2f(p.Pd[((int64_t)pair * p.Sq + q) * SkE + k]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int d = lane + 64 * u;
                if (d < D) {
                    ak[u] += ds * bf2f(p.Q[qrow * p.ldq + h * D + d]);
                    av[u] += pd * bf2f(p.dO[qrow * p.ldo + h * D + d]);
                }
            }
        }
        const int64_t krow = (int64_t)b * p.Sk + k;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int d = lane + 64 * u;
            if (d < D) {
                p.dK[krow * p.ldk + h * D + d] = f2bf(ak[u]);
                p.dV[krow * p.ldv + h * D + d] = f2bf(av[u]);
            }
        }
    }
}

bf16_t* g_scratch = nullptr;
size_t g_scratch_elems = 0;
std::mutex g_mu;

bf16_t* scratch(size_t elems) {          // grow-only, process-global (one device per process; calls are stream-ordered)
    std::lock_guard<std::mutex> lk(g_mu);
    if (elems > g_scratch_elems) {
        if (g_scratch) { (void)hipDeviceSynchronize(); (void)hipFree(g_scratch); g_scratch = nullptr; g_scratch_elems = 0; }
        void* q = nullptr;
        if (hipMalloc(&q, elems * sizeof(bf16_t)) != hipSuccess) return nullptr;
        g_scratch = (bf16_t*)q; g_scratch_elems = elems;
    }
    return g_scratch;
}

int fill(GenP& p, const kzv_attn_args* a, int D, bool bwd) {
    if (a->mode != 0) return kzv_fail(KZV_E_ARG, "attn: the causal / key-padding mode exists for head_dim 64 only");
    if (D < 8 || D > 128 || D % 8) return kzv_fail(KZV_E_ARG, "attn: head_dim must be a multiple of 8 in 8..128 (got %d)", D);
    if (a->Sq <= 0 || a->Sk <= 0 || a->Sk > 512) return kzv_fail(KZV_E_ARG, "attn (generic head_dim): Sk must be in 1..512");
    p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V; p.O = (bf16_t*)a->O; p.LSE = a->LSE;
    p.dO = (const bf16_t*)a->dO; p.dQ = (bf16_t*)a->dQ; p.dK = (bf16_t*)a->dK; p.dV = (bf16_t*)a->dV;
    p.dSs = nullptr; p.Pd = nullptr;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.B = a->B; p.heads = a->heads; p.Sq = a->Sq; p.Sk = a->Sk; p.D = D;
    p.scale = 1.f / sqrtf((float)D);
    kzv_drop_params(a->drop_p, &p.thr16, &p.inv_keep);
    p.key = a->drop_key;
    (void)bwd;
    return KZV_OK;
}

size_t lds_bytes(const GenP& p) {
    const size_t DP = p.D + 2, SkP = (p.Sk + 63) & ~63;
    size_t kv = 2 * (size_t)p.Sk * DP * sizeof(bf16_t);
    kv = (kv + 3) & ~(size_t)3;
    return kv + 4 + 4 * (2 * (size_t)p.D + SkP) * sizeof(float);
}

}  // namespace

int kzv_attn_generic(const kzv_attn_args* a, int D, bool bwd, hipStream_t s) {
    GenP p;
    if (int rc = fill(p, a, D, bwd)) return rc;
    const size_t lds = lds_bytes(p);
    if (lds > 160 * 1024) return kzv_fail(KZV_E_ARG, "attn (generic head_dim): %d keys x head_dim %d do not fit the 160 KiB LDS", p.Sk, D);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_gen_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn_gen_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const dim3 grid(p.B * p.heads, (p.Sq + QT - 1) / QT);
    if (!bwd) {
        hipLaunchKernelGGL(attn_gen_kernel<false>, grid, dim3(256), lds, s, p);
        return kzv_check_launch("attn_fwd (generic head_dim)");
    }
    const size_t per = (size_t)p.B * p.heads * p.Sq * ((p.Sk + 1) & ~1);
    bf16_t* sc = scratch(2 * per);
    if (!sc) return kzv_fail(KZV_E_HIP, "attn_bwd (generic head_dim): %zu bytes of scratch", 2 * per * sizeof(bf16_t));
    p.dSs = sc; p.Pd = sc + per;
    hipLaunchKernelGGL(attn_gen_kernel<true>, grid, dim3(256), lds, s, p);
    hipLaunchKernelGGL(attn_gen_dkv_kernel, dim3(p.B * p.heads, (p.Sk + QT - 1) / QT), dim3(256), 0, s, p);
    return kzv_check_launch("attn_bwd (generic head_dim)");
}
