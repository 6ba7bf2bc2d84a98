// gemm_nt for FEW ROWS (M <= ~1k): the KV-cached generation step runs every nn.Linear of the decoder on one token per
// sequence (M = batch, or batch x beams), K and N in 256..4352 (src/models/trocr_model.py:306-316 -> HF RobertaLayer with
// use_cache).  At M = 256 the 128x128 LDS-staged kernel fills 4..12 of the 256 CUs and costs 7-11 us per call; here one
// workgroup of four waves owns a 16-row x 64-column output tile (grid N/64 x M/16 = 64..1088 workgroups), each wave reads the
// A and B fragments of its quarter of K straight from global memory / L2 in MFMA layout (16 B per lane; the whole operand set
// of a decode step is L2-resident) with every load in flight at once, and the tile finishes with the same fused epilogues
// (gemm_nt.h: bias, GELU + saved pre-activation, residual add, fp32 / bf16).  Latency-bound by design.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "gemm_nt.h"
#include <cstdlib>

namespace {

// One 256-thread workgroup per 16-row x 64-column tile: wave w reduces K-slice w (K/4, rounded to 32-deep steps) with EVERY
// fragment load of its slice in flight before the first MFMA (K <= 1024: <= 8 steps x 5 fragments), the four partial tiles
// meet in LDS and wave 0's ... every wave finishes 16 of the 64 columns.  The residual / pre-activation of the epilogue is
// requested up front as well: the kernel is two dependent memory round trips long, whatever K.
template <int EPI, int STEPS>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const NtParams p) {
    __shared__ float part[4][16][68];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
    const int ksteps = p.K / 32;                                   // K % 64 == 0
    const int per = (ksteps + 3) / 4;                              // 32-deep steps per wave (<= STEPS)
    const int ks0 = w * per, ks1 = min(ksteps, ks0 + per);
    const bf16_t* ap = p.A + (int64_t)min(m0 + l15, p.M - 1) * p.lda + g * 8;
    const bf16_t* bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = p.B + (int64_t)min(n0 + j * 16 + l15, p.n_valid - 1) * p.ldb + g * 8;   // rows >= n_valid: clamped, zeroed below
    // epilogue operands of this thread's output piece (row er, 4 columns at ec), requested before anything else
    const int er = tid >> 4, ec = (tid & 15) * 4;                  // 16 rows x 16 four-column groups = 256 threads
    const int em = m0 + er, en = n0 + ec;
    const bool eok = em < p.M && en < p.N;
    float4 r4 = make_float4(0, 0, 0, 0); uint2 u2 = make_uint2(0, 0);
    if (eok) {
        if (EPI == KZV_EPI_RESID) r4 = *(const float4*)(p.resid + (int64_t)em * p.ldr + en);
        if (EPI == KZV_EPI_DGELU) u2 = *(const uint2*)(p.aux + (int64_t)em * p.ldaux + en);
    }
    bf16x8 fa[STEPS], fb[STEPS][4];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int ks = min(ks0 + i, ksteps - 1);                   // beyond the slice: a valid (unused) address
        fa[i] = *(const bf16x8*)(ap + ks * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[i][j] = *(const bf16x8*)(bp[j] + ks * 32);
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
        if (ks0 + i < ks1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i][j], fa[i], acc[j], 0, 0, 0);   // D[n = 4g + r][m = l15]
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)&part[w][l15][j * 16 + 4 * g] = acc[j];
    __syncthreads();
    if (!eok) return;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float sum = part[0][er][ec + r] + part[1][er][ec + r] + part[2][er][ec + r] + part[3][er][ec + r];
        const bool ok = en + r < p.n_valid;
        v[r] = ok ? sum + ((EPI != KZV_EPI_DGELU && p.bias) ? p.bias[en + r] : 0.f) : 0.f;
    }
    nt_emit<EPI>(p, em, en, v, r4, u2);
}

// ---- the same kernel with the decoder's LayerNorms folded in (generation step, hidden size 256) -------------------------------
// RoBERTa is post-LN: every sub-layer output s is normalised once and the result feeds exactly one GEMM as its A operand and one
// later residual add.  At M = batch rows a LayerNorm launch is 4.7 us of latency for 0.25 MB of work, twenty times per token.
// Here the CONSUMERS normalise instead, keeping the 16 x 64 tiling (and with it the workgroup count -- fusing the other way,
// whole rows per workgroup, lost twice):
//   LNA: the A operand is LN(xa) (K = 256 exactly): wave w reads its K-quarter of the 16 fp32 rows in fragment layout, the
//        row sums meet in LDS (two passes, like ln_fwd_kernel), and the normalised values are packed straight into the fragments.
//   LNR: the residual of the RESID epilogue is LN(xr) (N = 256 exactly): the 16 threads that finish a row each read 16 of its
//        256 values for the statistics (four shuffles), then normalise the 4 columns they own.
// Every workgroup of a row block repeats the same 16-row statistics (4..17 times): 16 KB of L2 reads, no launch.
struct RowsLn {
    const float* xa; const float* ga; const float* ba;       // LNA source [M, 256] + gamma / beta
    const float* xr; const float* gr; const float* br;       // LNR source [M, 256] + gamma / beta
    float eps;
};

template <int EPI, int STEPS, bool LNA, bool LNR>
__global__ __launch_bounds__(256) void gemm_rows_ln_kernel(const NtParams p, const RowsLn q) {
    __shared__ float part[4][16][68];
    __shared__ float red[2][4][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
    const int ksteps = p.K / 32;
    const int per = (ksteps + 3) / 4;
    const int ks0 = w * per, ks1 = min(ksteps, ks0 + per);
    const int arow = min(m0 + l15, p.M - 1);
    const bf16_t* bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = p.B + (int64_t)min(n0 + j * 16 + l15, p.n_valid - 1) * p.ldb + g * 8;
    const int er = tid >> 4, ec = (tid & 15) * 4;
    const int em = m0 + er, en = n0 + ec;
    const bool eok = em < p.M && en < p.N;
    const int emc = min(em, p.M - 1);
    // residual side first (its loads fly with everything else)
    float4 xr16[4]; float4 xrn = make_float4(0, 0, 0, 0), grn = xrn, brn = xrn;
    float4 r4 = make_float4(0, 0, 0, 0);
    if constexpr (LNR) {
        const float4* xrow = (const float4*)(q.xr + (int64_t)emc * 256 + (tid & 15) * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) xr16[c] = xrow[c];
        const int enc = min(en, 252);
        xrn = *(const float4*)(q.xr + (int64_t)emc * 256 + enc);
        grn = *(const float4*)(q.gr + enc); brn = *(const float4*)(q.br + enc);
    } else if (EPI == KZV_EPI_RESID) {
        if (eok) r4 = *(const float4*)(p.resid + (int64_t)em * p.ldr + en);
    }
    // operand loads
    bf16x8 fa[STEPS], fb[STEPS][4];
    float xa[LNA ? 2 : 1][8];
    if constexpr (LNA) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4* src = (const float4*)(q.xa + (int64_t)arow * 256 + (ks0 + i) * 32 + g * 8);
            const float4 a = src[0], b = src[1];
            xa[i][0] = a.x; xa[i][1] = a.y; xa[i][2] = a.z; xa[i][3] = a.w; xa[i][4] = b.x; xa[i][5] = b.y; xa[i][6] = b.z; xa[i][7] = b.w;
        }
    } else {
        const bf16_t* ap = p.A + (int64_t)arow * p.lda + g * 8;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) fa[i] = *(const bf16x8*)(ap + min(ks0 + i, ksteps - 1) * 32);
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const int ks = min(ks0 + i, ksteps - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[i][j] = *(const bf16x8*)(bp[j] + ks * 32);
    }
    if constexpr (LNA) {
        // row statistics over the four waves' K-quarters (K = 256 = 4 x 64), two passes
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xa[i][e];
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        if (g == 0) red[0][w][l15] = s;
        __syncthreads();
        const float mean = (red[0][0][l15] + red[0][1][l15] + red[0][2][l15] + red[0][3][l15]) * (1.f / 256.f);
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = xa[i][e] - mean; v += d * d; }
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (g == 0) red[1][w][l15] = v;
        __syncthreads();
        const float rstd = rsqrtf((red[1][0][l15] + red[1][1][l15] + red[1][2][l15] + red[1][3][l15]) * (1.f / 256.f) + q.eps);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k0 = (ks0 + i) * 32 + g * 8;
            const float4 g0 = *(const float4*)(q.ga + k0), g1 = *(const float4*)(q.ga + k0 + 4);
            const float4 b0 = *(const float4*)(q.ba + k0), b1 = *(const float4*)(q.ba + k0 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (xa[i][e] - mean) * rstd * gg[e] + bb[e];
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            fa[i] = __builtin_bit_cast(bf16x8, (u32x4){pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]), pack_bf2(y[4], y[5]), pack_bf2(y[6], y[7])});
        }
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < STEPS; ++i)
        if (ks0 + i < ks1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i][j], fa[i], acc[j], 0, 0, 0);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)&part[w][l15][j * 16 + 4 * g] = acc[j];
    if constexpr (LNR) {
        // statistics of residual row em: the 16 threads of the row hold 16 values each
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) s += xr16[c].x + xr16[c].y + xr16[c].z + xr16[c].w;
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
        const float mean = s * (1.f / 256.f);
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float a = xr16[c].x - mean, b = xr16[c].y - mean, cc = xr16[c].z - mean, d = xr16[c].w - mean;
            v += a * a + b * b + cc * cc + d * d;
        }
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        const float rstd = rsqrtf(v * (1.f / 256.f) + q.eps);
        r4 = make_float4((xrn.x - mean) * rstd * grn.x + brn.x, (xrn.y - mean) * rstd * grn.y + brn.y,
                         (xrn.z - mean) * rstd * grn.z + brn.z, (xrn.w - mean) * rstd * grn.w + brn.w);
    }
    __syncthreads();
    if (!eok) return;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float sum = part[0][er][ec + r] + part[1][er][ec + r] + part[2][er][ec + r] + part[3][er][ec + r];
        const bool ok = en + r < p.n_valid;
        v[r] = ok ? sum + (p.bias ? p.bias[en + r] : 0.f) : 0.f;
    }
    const uint2 u2 = make_uint2(0, 0);
    nt_emit<EPI>(p, em, en, v, r4, u2);
}

int g_rows_max_m = -1;
int g_rows_scope = 0;

}  // namespace

// Few-rows GEMM whose A operand and / or residual is the LayerNorm of an fp32 [M, 256] tensor (see gemm_rows_ln_kernel).
// ln_a: LN source of the A operand (then p.A is ignored and K must be 256); ln_r: LN source of the residual (RESID, N = 256).
int kzv_rows_ln_launch(const NtParams& p, int epilogue, const float* ln_a, const float* ga, const float* ba, const float* ln_r, const float* gr,
                       const float* br, float eps, hipStream_t s) {
    if (ln_a && p.K != 256) return kzv_fail(KZV_E_ARG, "rows_ln: a normalised A operand needs K = 256");
    if (ln_r && (p.N != 256 || p.n_valid != 256 || epilogue != KZV_EPI_RESID)) return kzv_fail(KZV_E_ARG, "rows_ln: a normalised residual needs RESID and N = 256");
    if (!ln_a && !ln_r) return kzv_fail(KZV_E_ARG, "rows_ln: nothing to normalise");
    if (p.K % 64 || p.K > 1024) return kzv_fail(KZV_E_ARG, "rows_ln: K must be a multiple of 64 up to 1024");
    const RowsLn q{ln_a, ga, ba, ln_r, gr, br, eps};
    const dim3 grid((p.N + 63) / 64, (p.M + 15) / 16);
    const int per = (p.K / 32 + 3) / 4;
#define KZV_RLN(E, ST, LA, LR) hipLaunchKernelGGL((gemm_rows_ln_kernel<E, ST, LA, LR>), grid, dim3(256), 0, s, p, q)
    if (ln_a && !ln_r) {
        switch (epilogue) {
            case KZV_EPI_BF16: KZV_RLN(KZV_EPI_BF16, 2, true, false); break;
            case KZV_EPI_F32: KZV_RLN(KZV_EPI_F32, 2, true, false); break;
            case KZV_EPI_GELU: KZV_RLN(KZV_EPI_GELU, 2, true, false); break;
            case KZV_EPI_GELU_F32: KZV_RLN(KZV_EPI_GELU_F32, 2, true, false); break;
            default: return kzv_fail(KZV_E_ARG, "rows_ln: epilogue not instantiated with a normalised A operand");
        }
    } else if (!ln_a && ln_r) {
        if (per <= 2) KZV_RLN(KZV_EPI_RESID, 2, false, true); else KZV_RLN(KZV_EPI_RESID, 8, false, true);
    } else return kzv_fail(KZV_E_ARG, "rows_ln: A and residual both normalised is not instantiated");
#undef KZV_RLN
    return kzv_check_launch("gemm_rows_ln");
}

KzvRowsScope::KzvRowsScope() { ++g_rows_scope; }
KzvRowsScope::~KzvRowsScope() { --g_rows_scope; }

extern "C" int kzv_set_rows_max_m(int n) {
    if (n < 0) return kzv_fail(KZV_E_ARG, "set_rows_max_m: >= 0");
    g_rows_max_m = n;
    return KZV_OK;
}

// returns 1 when it took the launch
int kzv_rows_launch(const NtParams& p, int epilogue, hipStream_t s) {
    if (g_rows_max_m < 0) { const char* e = getenv("KZV_ROWS_MAX_M"); g_rows_max_m = e ? atoi(e) : 0; }
    if (p.M > (g_rows_scope > 0 ? 4096 : g_rows_max_m)) return 0;      // the generation step (KzvRowsScope) or an explicit threshold
    if (p.K > 4096) return 0;
    const dim3 grid((p.N + 63) / 64, (p.M + 15) / 16);
    const int per = (p.K / 32 + 3) / 4;                              // 32-deep steps per wave
#define KZV_ROWS_CASE(E) case E:                                                                                  \
        if (per <= 2) hipLaunchKernelGGL((gemm_rows_kernel<E, 2>), grid, dim3(256), 0, s, p);                      \
        else if (per <= 8) hipLaunchKernelGGL((gemm_rows_kernel<E, 8>), grid, dim3(256), 0, s, p);                 \
        else hipLaunchKernelGGL((gemm_rows_kernel<E, 32>), grid, dim3(256), 0, s, p);                              \
        break;
    switch (epilogue) {
        KZV_ROWS_CASE(KZV_EPI_BF16) KZV_ROWS_CASE(KZV_EPI_F32) KZV_ROWS_CASE(KZV_EPI_GELU) KZV_ROWS_CASE(KZV_EPI_RESID)
        KZV_ROWS_CASE(KZV_EPI_DGELU) KZV_ROWS_CASE(KZV_EPI_GELU_F32)
        default: return 0;
    }
#undef KZV_ROWS_CASE
    return 1;
}
