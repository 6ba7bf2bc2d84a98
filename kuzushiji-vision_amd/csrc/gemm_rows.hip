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
