// fp32-operand GEMMs on the f32-input matrix instruction (v_mfma_f32_32x32x2_f32: exact fp32 products and fp32 accumulation, the
// arithmetic of an fmaf chain; 1/16 of the bf16 MFMA rate = the fp32 vector peak, 157 TFLOP/s).  They exist for the model of
// ocr_lightning/model.py, which the reference trains in fp32 (ocr_lightning/train.py:132-140: pl.Trainer without `precision`) and
// whose own test demands singles == batched at 1e-6 (ocr_lightning/tests/test_model.py:48-76): kzv.OCRModel(precision="fp32") runs
// every nn.Conv2d (im2col + GEMM), nn.Linear and nn.LSTM input projection and all of their gradients through these two kernels.
//
//   kzv_gemm_nt_f32:  C[M,N] = A[M,K] . B[N,K]^T + bias (+ resid)         (forward and input gradients; epilogues F32, RESID)
//   kzv_gemm_tn_f32:  OUT[N,K] += P[Mtok,N]^T . Q[Mtok,K]  (+ dbias)       (weight gradients)
//
// One 256-thread workgroup per 64 x 64 output tile, each of its four waves a 32 x 32 block (16 accumulator registers), reduction
// steps of 32 staged through LDS in the layout the instruction reads (one float per lane: lanes 0..31 = 32 consecutive rows at
// k, lanes 32..63 the same rows at k + 1 -> [k][row] images, conflict-free ds_read_b32), the next step's global loads in flight
// while the current one is multiplied.  The instruction is slow (64 cycles per 32 x 32 x 2), so LDS and memory are nowhere near
// their limits; what matters at this model's sizes is filling the chip: launches with fewer than 256 tiles split the reduction
// over workgroups, whose partial tiles meet in a second, deterministic kernel (fixed summation order: no float atomics).
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int TB = 64;          // tile edge
constexpr int KB = 32;          // reduction step

// partial-tile workspace (grow-only; calls are stream-ordered)
float* g_ws = nullptr;
size_t g_ws_floats = 0;
float* f32_ws(size_t floats) {
    if (floats > g_ws_floats) {
        if (g_ws) { (void)hipDeviceSynchronize(); (void)hipFree(g_ws); g_ws = nullptr; g_ws_floats = 0; }
        void* q = nullptr;
        if (hipMalloc(&q, floats * sizeof(float)) != hipSuccess) return nullptr;
        g_ws = (float*)q; g_ws_floats = floats;
    }
    return g_ws;
}

struct NtF32 {
    const float* A; const float* B; float* C; const float* bias; const float* resid; float* part;
    int64_t lda, ldb, ldc, ldr;
    int M, N, K, n_valid, splits, ksteps;      // ksteps = reduction steps of 32 per split
};

// D = (B tile) x (A tile)^T in the instruction's terms: the lane that ends up with output row m = lane % 32 holds 4 CONSECUTIVE
// columns n per register quad (n = 8 * q + 4 * (lane / 32) + r), i.e. one 16-byte store per quad.
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const NtF32 p) {
    __shared__ float As[2][KB][TB], Bs[2][KB][TB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int tilesN = (p.N + TB - 1) / TB;
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x - tm * tilesN;
    const int split = blockIdx.y;
    const int k_begin = split * p.ksteps * KB;
    const int nsteps = min(p.ksteps, (p.K - k_begin) / KB);
    // loader: thread -> (row = tid & 63, 8 consecutive k at (tid >> 6) * 8); rows beyond M / n_valid are clamped (never stored / stored as 0)
    const int lrow = tid & 63, lk = (tid >> 6) * 8;
    const float* ga = p.A + (int64_t)min(tm * TB + lrow, p.M - 1) * p.lda + k_begin + lk;
    const float* gb = p.B + (int64_t)min(tn * TB + lrow, p.n_valid - 1) * p.ldb + k_begin + lk;
    f32x4 ra[2], rb[2];
    auto gload = [&](int step) {
        ra[0] = *(const f32x4*)(ga + step * KB); ra[1] = *(const f32x4*)(ga + step * KB + 4);
        rb[0] = *(const f32x4*)(gb + step * KB); rb[1] = *(const f32x4*)(gb + step * KB + 4);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) { As[buf][lk + h * 4 + j][lrow] = ra[h][j]; Bs[buf][lk + h * 4 + j][lrow] = rb[h][j]; }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (nsteps > 0) { gload(0); lstore(0); }
    __syncthreads();
    const int kk = lane >> 5, r31 = lane & 31;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
#pragma unroll
        for (int k2 = 0; k2 < KB / 2; ++k2) {
            const float b = Bs[buf][k2 * 2 + kk][wc * 32 + r31];          // instruction's A operand: rows = output columns n
            const float a = As[buf][k2 * 2 + kk][wr * 32 + r31];          // instruction's B operand: columns = output rows m
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    const int m = tm * TB + wr * 32 + r31;
    if (m >= p.M) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n0 = tn * TB + wc * 32 + q * 8 + kk * 4;
        if (n0 >= p.N) continue;
        f32x4 v = (f32x4){acc[q * 4 + 0], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
        if (p.splits > 1) { *(f32x4*)(p.part + ((int64_t)split * p.M + m) * p.N + n0) = v; continue; }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (n0 + r < p.n_valid) ? v[r] + (p.bias ? p.bias[n0 + r] : 0.f) : 0.f;
        if (p.resid) v += *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n0);
        *(f32x4*)(p.C + (int64_t)m * p.ldc + n0) = v;
    }
}
// C = sum over the splits (in order) + bias (+ resid); 4 columns per thread
__global__ void gemm_nt_f32_fold_kernel(const NtF32 p) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n4 = p.N >> 2;
    if (t >= (int64_t)p.M * n4) return;
    const int m = (int)(t / n4), n0 = (int)(t - (int64_t)m * n4) * 4;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splits; ++s) v += *(const f32x4*)(p.part + ((int64_t)s * p.M + m) * p.N + n0);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (n0 + r < p.n_valid) ? v[r] + (p.bias ? p.bias[n0 + r] : 0.f) : 0.f;
    if (p.resid) v += *(const f32x4*)(p.resid + (int64_t)m * p.ldr + n0);
    *(f32x4*)(p.C + (int64_t)m * p.ldc + n0) = v;
}

struct TnF32 {
    const float* P; const float* Q; float* OUT; float* part;
    int64_t ldp, ldq, ldo;
    int Mtok, N, K, n_store, splits, tsteps;   // tsteps = token steps of 32 per split
};
// D rows = output columns k (instruction's A operand = Q), D columns = output rows n (B operand = P): a lane holds n = lane % 32
// and 4 consecutive k per register quad.  Token rows of both operands are the natural [token][column] LDS image.
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const TnF32 p) {
    __shared__ float Ps[2][KB][TB], Qs[2][KB][TB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int tilesK = (p.K + TB - 1) / TB;
    const int tnb = blockIdx.x / tilesK, tkb = blockIdx.x - tnb * tilesK;
    const int split = blockIdx.y;
    const int t_begin = split * p.tsteps * KB;
    const int t_end = min(p.Mtok, t_begin + p.tsteps * KB);
    const int nsteps = (t_end - t_begin + KB - 1) / KB;
    // loader: thread -> (token = tid >> 4 and + 16, 4 consecutive columns at (tid & 15) * 4); columns beyond N / K clamped to the last quad
    const int ltok = tid >> 4, lc = (tid & 15) * 4;
    const int pn = min(tnb * TB + lc, p.N - 4), qk = min(tkb * TB + lc, p.K - 4);
    f32x4 rp[2], rq[2];
    auto gload = [&](int step) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = t_begin + step * KB + ltok + h * 16;
            if (t < t_end) {
                rp[h] = *(const f32x4*)(p.P + (int64_t)t * p.ldp + pn); rq[h] = *(const f32x4*)(p.Q + (int64_t)t * p.ldq + qk);
            } else { rp[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; rq[h] = rp[h]; }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { *(f32x4*)&Ps[buf][ltok + h * 16][lc] = rp[h]; *(f32x4*)&Qs[buf][ltok + h * 16][lc] = rq[h]; }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (nsteps > 0) { gload(0); lstore(0); }
    __syncthreads();
    const int kk = lane >> 5, r31 = lane & 31;
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
#pragma unroll
        for (int t2 = 0; t2 < KB / 2; ++t2) {
            const float q = Qs[buf][t2 * 2 + kk][wc * 32 + r31];
            const float pv = Ps[buf][t2 * 2 + kk][wr * 32 + r31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q, pv, acc, 0, 0, 0);
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    const int n = tnb * TB + wr * 32 + r31;
    if (n >= p.n_store) return;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const int k0 = tkb * TB + wc * 32 + qd * 8 + kk * 4;
        if (k0 >= p.K) continue;
        const f32x4 v = (f32x4){acc[qd * 4 + 0], acc[qd * 4 + 1], acc[qd * 4 + 2], acc[qd * 4 + 3]};
        if (p.splits > 1) *(f32x4*)(p.part + ((int64_t)split * p.n_store + n) * p.K + k0) = v;
        else { f32x4* o = (f32x4*)(p.OUT + (int64_t)n * p.ldo + k0); *o += v; }
    }
}
__global__ void gemm_tn_f32_fold_kernel(const TnF32 p) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k4 = p.K >> 2;
    if (t >= (int64_t)p.n_store * k4) return;
    const int n = (int)(t / k4), k0 = (int)(t - (int64_t)n * k4) * 4;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splits; ++s) v += *(const f32x4*)(p.part + ((int64_t)s * p.n_store + n) * p.K + k0);
    f32x4* o = (f32x4*)(p.OUT + (int64_t)n * p.ldo + k0);
    *o += v;
}
// dbias[n] += sum over the token rows of P[t][n] (the nn.Linear / nn.LSTM bias gradients: a handful of rows), in row order
__global__ void colsum_f32_kernel(const float* __restrict__ P, int64_t ldp, int Mtok, int n_store, float* __restrict__ dbias) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_store) return;
    float s = 0.f;
    for (int t = 0; t < Mtok; ++t) s += P[(int64_t)t * ldp + n];
    dbias[n] += s;
}

int cus() {
    static int v = -1;
    if (v < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        v = n;
    }
    return v;
}

}  // namespace

extern "C" int kzv_gemm_nt_f32(const kzv_gemm_nt_args* a, int epilogue, void* stream) {
    if (!a || !a->A || !a->B || !a->C) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: empty shape");
    if (a->K % KB) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: K must be a multiple of 32");
    if (a->N % 4 || a->ldc % 4 || a->lda % 4 || a->ldb % 4) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: N, lda, ldb, ldc must be multiples of 4 (16-byte rows)");
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: operands must be 16-byte aligned");
    if (epilogue != KZV_EPI_F32 && epilogue != KZV_EPI_RESID) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: epilogue must be F32 or RESID");
    if (epilogue == KZV_EPI_RESID && (!a->resid || a->ldr % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: RESID needs resid, ldr % 4 == 0");
    if (a->drop_p != 0.f) return kzv_fail(KZV_E_ARG, "gemm_nt_f32: no dropout epilogue");
    NtF32 p{};
    p.A = (const float*)a->A; p.B = (const float*)a->B; p.C = (float*)a->C; p.bias = a->bias;
    p.resid = epilogue == KZV_EPI_RESID ? a->resid : nullptr;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr;
    p.M = a->M; p.N = a->N; p.K = a->K; p.n_valid = a->n_valid > 0 ? a->n_valid : a->N;
    const int tiles = ((p.M + TB - 1) / TB) * ((p.N + TB - 1) / TB);
    const int ksteps = p.K / KB;
    int splits = 1;
    if (tiles < cus()) { splits = (2 * cus() + tiles - 1) / tiles; if (splits > ksteps / 4) splits = ksteps / 4; if (splits < 1) splits = 1; if (splits > 16) splits = 16; }
    p.ksteps = (ksteps + splits - 1) / splits;
    splits = (ksteps + p.ksteps - 1) / p.ksteps;
    p.splits = splits;
    hipStream_t s = (hipStream_t)stream;
    if (splits > 1) {
        p.part = f32_ws((size_t)splits * p.M * p.N);
        if (!p.part) return kzv_fail(KZV_E_HIP, "gemm_nt_f32: workspace");
    }
    hipLaunchKernelGGL(gemm_nt_f32_kernel, dim3(tiles, splits), dim3(256), 0, s, p);
    if (splits > 1) hipLaunchKernelGGL(gemm_nt_f32_fold_kernel, dim3((unsigned)(((int64_t)p.M * (p.N / 4) + 255) / 256)), dim3(256), 0, s, p);
    return kzv_check_launch("gemm_nt_f32");
}

extern "C" int kzv_gemm_tn_f32(const kzv_gemm_tn_args* a, void* stream) {
    if (!a || !a->P || !a->Q || !a->OUT) return kzv_fail(KZV_E_ARG, "gemm_tn_f32: null operand");
    if (a->Mtok <= 0 || a->N < 4 || a->K < 4) return kzv_fail(KZV_E_ARG, "gemm_tn_f32: empty shape");
    if (a->N % 4 || a->K % 4 || a->ldp % 4 || a->ldq % 4 || a->ldo % 4) return kzv_fail(KZV_E_ARG, "gemm_tn_f32: N, K and the leading dimensions must be multiples of 4");
    if (((uintptr_t)a->P | (uintptr_t)a->Q | (uintptr_t)a->OUT) & 15) return kzv_fail(KZV_E_ARG, "gemm_tn_f32: operands must be 16-byte aligned");
    TnF32 p{};
    p.P = (const float*)a->P; p.Q = (const float*)a->Q; p.OUT = a->OUT;
    p.ldp = a->ldp; p.ldq = a->ldq; p.ldo = a->ldo;
    p.Mtok = a->Mtok; p.N = a->N; p.K = a->K; p.n_store = a->n_store > 0 ? a->n_store : a->N;
    const int tiles = ((p.n_store + TB - 1) / TB) * ((p.K + TB - 1) / TB);
    const int tsteps = (p.Mtok + KB - 1) / KB;
    int splits = 1;
    if (tiles < 2 * cus()) { splits = (2 * cus() + tiles - 1) / tiles; if (splits > tsteps / 4) splits = tsteps / 4; if (splits < 1) splits = 1; if (splits > 256) splits = 256; }
    p.tsteps = (tsteps + splits - 1) / splits;
    splits = (tsteps + p.tsteps - 1) / p.tsteps;
    p.splits = splits;
    hipStream_t s = (hipStream_t)stream;
    if (splits > 1) {
        p.part = f32_ws((size_t)splits * p.n_store * p.K);
        if (!p.part) return kzv_fail(KZV_E_HIP, "gemm_tn_f32: workspace");
    }
    hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(tiles, splits), dim3(256), 0, s, p);
    if (splits > 1) hipLaunchKernelGGL(gemm_tn_f32_fold_kernel, dim3((unsigned)(((int64_t)p.n_store * (p.K / 4) + 255) / 256)), dim3(256), 0, s, p);
    if (a->dbias) hipLaunchKernelGGL(colsum_f32_kernel, dim3((p.n_store + 255) / 256), dim3(256), 0, s, p.P, p.ldp, p.Mtok, p.n_store, a->dbias);
    return kzv_check_launch("gemm_tn_f32");
}
