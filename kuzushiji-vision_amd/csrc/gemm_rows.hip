// gemm_nt for FEW ROWS (M <= ~1k): the KV-cached generation step runs every nn.Linear of the decoder on one token per
// sequence (M = batch, or batch x beams), K and N in 256..4352 (src/models/trocr_model.py:306-316 -> HF RobertaLayer with
// use_cache).  At M = 256 the 128x128 LDS-staged kernel fills 4..12 of the 256 CUs and costs 7-11 us per call; here one
// WAVE owns a 16-row x 64-column output tile (grid N/64 x M/16 = 64..1088 waves), reads its A and B fragments straight
// from global memory / L2 in MFMA layout (16 B per lane; the whole operand set of a decode step is L2-resident) with every
// load of the K sweep in flight at once, and finishes with the same fused epilogues (gemm_nt.h: bias, GELU + saved
// pre-activation, residual add, fp32 / bf16).  Latency-bound by design: ~one L2 round trip + K/32 x 4 MFMAs.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "gemm_nt.h"
#include <cstdlib>

namespace {

template <int EPI>
__global__ __launch_bounds__(64) void gemm_rows_kernel(const NtParams p) {
    const int lane = threadIdx.x, g = lane >> 4, l15 = lane & 15;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 64;
    const bf16_t* ap = p.A + (int64_t)min(m0 + l15, p.M - 1) * p.lda + g * 8;
    const bf16_t* bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = p.B + (int64_t)min(n0 + j * 16 + l15, p.n_valid - 1) * p.ldb + g * 8;   // rows >= n_valid: clamped, zeroed below
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K is a multiple of 64 (launcher): 64-deep steps, the next step's ten fragments requested before this step's MFMAs
    bf16x8 fa[2][2], fb[2][2][4];
    auto load = [&](int buf, int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fa[buf][h] = *(const bf16x8*)(ap + k0 + 32 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[buf][h][j] = *(const bf16x8*)(bp[j] + k0 + 32 * h);
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[buf][h][j], fa[buf][h], acc[j], 0, 0, 0);     // D[n = 4g + r][m = l15]
    };
    load(0, 0);
    for (int k0 = 0; k0 < p.K; k0 += 128) {
        if (k0 + 64 < p.K) load(1, k0 + 64);
        mma(0);
        if (k0 + 64 >= p.K) break;
        if (k0 + 128 < p.K) load(0, k0 + 128);
        mma(1);
    }
    const int m = m0 + l15;
    if (m >= p.M) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + 4 * g;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = n + r < p.n_valid;
            v[r] = ok ? acc[j][r] + ((EPI != KZV_EPI_DGELU && p.bias) ? p.bias[n + r] : 0.f) : 0.f;
        }
        float4 r4 = make_float4(0, 0, 0, 0); uint2 u2 = make_uint2(0, 0);
        if (EPI == KZV_EPI_RESID) r4 = *(const float4*)(p.resid + (int64_t)m * p.ldr + n);
        if (EPI == KZV_EPI_DGELU) u2 = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n);
        nt_emit<EPI>(p, m, n, v, r4, u2);
    }
}

int g_rows_max_m = -1;
int g_rows_scope = 0;

}  // namespace

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
    const dim3 grid((p.N + 63) / 64, (p.M + 15) / 16);
#define KZV_ROWS_CASE(E) case E: hipLaunchKernelGGL((gemm_rows_kernel<E>), grid, dim3(64), 0, s, p); break;
    switch (epilogue) {
        KZV_ROWS_CASE(KZV_EPI_BF16) KZV_ROWS_CASE(KZV_EPI_F32) KZV_ROWS_CASE(KZV_EPI_GELU) KZV_ROWS_CASE(KZV_EPI_RESID)
        KZV_ROWS_CASE(KZV_EPI_DGELU) KZV_ROWS_CASE(KZV_EPI_GELU_F32)
        default: return 0;
    }
#undef KZV_ROWS_CASE
    return 1;
}
