// bf16 MFMA GEMMs for gfx950 (MI355X).
//
//  gemm_nt : C[M,N]  = A[M,K] . B[N,K]^T  (+ fused epilogue)   -- every nn.Linear forward and dgrad
//  gemm_tn : O[N,K] += P[Mt,N]^T . Q[Mt,K]                      -- every weight gradient
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4 MFMA
// 16x16x32 tiles), reduction step 64, operands streamed global->LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4) into a 2-stage ring, XOR-swizzled on the SOURCE address so the LDS image
// stays lane-linear while fragment reads are bank-conflict free.
//
// gemm_nt reads both fragments as 16 contiguous bytes (k is the fast axis of A and of B).
// gemm_tn reduces over the ROW index of both operands, so its fragments come out of LDS through the
// gfx950 transposing read ds_read_b64_tr_b16 (no transposed copies in HBM).
//
// MFMA operands are swapped (B fragment first) so every lane ends up with 4 consecutive output
// COLUMNS of one row: epilogue loads/stores are 8/16-byte vectors.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;   // 32 KiB
constexpr int NT_LDS = 2 * STAGE_BYTES;           // 64 KiB -> 2 workgroups / CU

struct NtParams {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; const float* resid; bf16_t* aux;
    const void* zero16;
    int64_t lda, ldb, ldc, ldr, ldaux;
    int M, N, K, n_valid;
    unsigned drop_thr16; float drop_inv_keep; unsigned drop_key;
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = w >> 1, wn = w & 1;
    const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tilesM * tilesN);
    const int tm = id / tilesN, tn = id - tm * tilesN;

    // ---- per-lane source pointers for the 4+4 LDS-DMA pieces this wave issues per stage ----
    // piece j covers tile rows w*32 + j*8 .. +7 (8 rows x 128 B); lane -> row lane>>3, 16-B slot lane&7.
    // slot s of row r holds source chunk s ^ (r & 7)   (involution; the read applies the same XOR)
    const char* pa[4]; const char* pb[4]; int sa[4], sb[4];
    {
        const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = w * 32 + j * 8 + r8;
            const int am = tm * BM + row, bn = tn * BN + row;
            const bool va = am < p.M, vb = bn < p.n_valid;
            pa[j] = va ? (const char*)(p.A + (int64_t)am * p.lda + chunk * 8) : (const char*)p.zero16;
            pb[j] = vb ? (const char*)(p.B + (int64_t)bn * p.ldb + chunk * 8) : (const char*)p.zero16;
            sa[j] = va ? BK * 2 : 0;
            sb[j] = vb ? BK * 2 : 0;
        }
    }
    auto stage = [&](int s) {
        char* dA = smem + s * STAGE_BYTES + (w * 32) * 128;
        char* dB = dA + BM * BK * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { glds16(pa[j], dA + j * 1024); pa[j] += sa[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { glds16(pb[j], dB + j * 1024); pb[j] += sb[j]; }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a stage (row-major 128-B rows, XOR-swizzled 16-B slots)
    const int a_row_off = (wm * 64 + l15) * 128;
    const int b_row_off = BM * BK * 2 + (wn * 64 + l15) * 128;
    const int sw = lane & 7;

    auto compute = [&](int s) {
        const char* base = smem + s * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ((ks * 4 + g) ^ sw) << 4;
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(base + a_row_off + i * 16 * 128 + slot);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(base + b_row_off + j * 16 * 128 + slot);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = p.K / BK;
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nk - 1; ++t) {
        stage(cur ^ 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // ---- epilogue: lane holds C[m][n0..n0+3], m = tile row i*16 + l15, n0 = j*16 + 4g ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tm * BM + wm * 64 + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = tn * BN + wn * 64 + j * 16 + 4 * g;
            if (n0 >= p.N) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (EPI != KZV_EPI_DGELU && p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (n0 + r < p.n_valid) ? p.bias[n0 + r] : 0.f;
            }
            if (EPI == KZV_EPI_BF16) {
                *(uint2*)((bf16_t*)p.C + (int64_t)m * p.ldc + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            } else if (EPI == KZV_EPI_F32) {
                *(float4*)((float*)p.C + (int64_t)m * p.ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (EPI == KZV_EPI_GELU) {
                *(uint2*)(p.aux + (int64_t)m * p.ldaux + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                *(uint2*)((bf16_t*)p.C + (int64_t)m * p.ldc + n0) =
                    make_uint2(pack_bf2(gelu_erf(v[0]), gelu_erf(v[1])), pack_bf2(gelu_erf(v[2]), gelu_erf(v[3])));
            } else if (EPI == KZV_EPI_GELU_F32) {
                *(uint2*)(p.aux + (int64_t)m * p.ldaux + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                *(float4*)((float*)p.C + (int64_t)m * p.ldc + n0) = make_float4(gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3]));
            } else if (EPI == KZV_EPI_RESID) {
                if (p.drop_thr16) {
                    const unsigned e = (unsigned)m * (unsigned)p.N + (unsigned)n0;
                    const unsigned b0 = drop_bits(p.drop_key, e >> 1), b1 = drop_bits(p.drop_key, (e >> 1) + 1);
                    v[0] *= drop_keep(b0, 0, p.drop_thr16, p.drop_inv_keep);
                    v[1] *= drop_keep(b0, 1, p.drop_thr16, p.drop_inv_keep);
                    v[2] *= drop_keep(b1, 0, p.drop_thr16, p.drop_inv_keep);
                    v[3] *= drop_keep(b1, 1, p.drop_thr16, p.drop_inv_keep);
                }
                const float4 r4 = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
                *(float4*)((float*)p.C + (int64_t)m * p.ldc + n0) =
                    make_float4(v[0] + r4.x, v[1] + r4.y, v[2] + r4.z, v[3] + r4.w);
            } else if (EPI == KZV_EPI_DGELU) {
                const uint2 u = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
                v[0] *= gelu_erf_grad(bf2f((bf16_t)(u.x & 0xffff)));
                v[1] *= gelu_erf_grad(bf2f((bf16_t)(u.x >> 16)));
                v[2] *= gelu_erf_grad(bf2f((bf16_t)(u.y & 0xffff)));
                v[3] *= gelu_erf_grad(bf2f((bf16_t)(u.y >> 16)));
                *(uint2*)((bf16_t*)p.C + (int64_t)m * p.ldc + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: OUT[n][k] += sum_t P[t][n] * Q[t][k].   LDS tiles are [64 tokens][128 cols] (256-B rows).
// Reduction index inside a 32-token MFMA step is permuted (element j of lane group g <-> token
// 16*(j>>2) + 4g + (j&3)); both operands use the same permutation so the sum is unchanged, and the two
// groups of a 32-lane half then read 8 consecutive rows -> the XOR swizzle below is conflict-free.
struct TnParams {
    const bf16_t* P; const bf16_t* Q; float* OUT; const void* zero16;
    int64_t ldp, ldq, ldo;
    int Mtok, N, K, n_store, splits, chunk;   // chunk = tokens per split (multiple of 64)
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int wn = w >> 1, wk = w & 1;     // wave tile: n in [wn*64,+64), k in [wk*64,+64)
    const int tilesN = (p.N + 127) / 128, tilesK = (p.K + 127) / 128;
    const int per = tilesN * tilesK;
    const int id = xcd_remap(blockIdx.x, per * p.splits);
    const int split = id / per, rem = id - split * per;
    const int tnb = rem / tilesK, tkb = rem - tnb * tilesK;
    const int t_begin = split * p.chunk;
    const int t_end = min(p.Mtok, t_begin + p.chunk);
    if (t_begin >= t_end) return;

    // staging: piece j of wave w covers tile rows (w*4 + j)*4 .. +3 (4 rows x 256 B); lane -> row lane>>4,
    // slot lane&15; slot s of row r holds source chunk s ^ ((r & 7) << 1).
    const char* pp[4]; const char* pq[4]; int64_t sp[4], sq[4]; int rowj[4];
    {
        const int r4 = lane >> 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (w * 4 + j) * 4 + r4;
            rowj[j] = row;
            const int chunk = (lane & 15) ^ ((row & 7) << 1);
            const int ncol = tnb * 128 + chunk * 8, kcol = tkb * 128 + chunk * 8;
            const bool vp = ncol < p.N, vq = kcol < p.K;
            pp[j] = vp ? (const char*)(p.P + (int64_t)(t_begin + row) * p.ldp + ncol) : nullptr;
            pq[j] = vq ? (const char*)(p.Q + (int64_t)(t_begin + row) * p.ldq + kcol) : nullptr;
            sp[j] = 64 * p.ldp * 2; sq[j] = 64 * p.ldq * 2;
        }
    }
    auto stage = [&](int s, int t0) {
        char* dP = smem + s * STAGE_BYTES + (w * 16) * 256;
        char* dQ = dP + 64 * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = t0 + rowj[j] < t_end;
            glds16((in && pp[j]) ? pp[j] : (const char*)p.zero16, dP + j * 1024);
            if (pp[j]) pp[j] += sp[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool in = t0 + rowj[j] < t_end;
            glds16((in && pq[j]) ? pq[j] : (const char*)p.zero16, dQ + j * 1024);
            if (pq[j]) pq[j] += sq[j];
        }
    };

    f32x4 acc[4][4];   // [k-tile][n-tile]: D rows = k (4g+r), D cols = n (l15)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposing-read addresses: lane supplies row (token) = 4g + (l15>>2) [+16 for the 2nd read],
    // columns c0 + (l15&3)*4 .. +3 of a 16-column block.
    const int trow = 4 * g + (l15 >> 2);
    const int tsw = (trow & 7) << 1;               // (row+16)&7 == row&7: same swizzle for both reads
    const int sub8 = (l15 & 1) * 8;                // which 8-byte half of the 16-B chunk
    const int cq = l15 >> 1 & 1;                   // chunk within the 16-col block (2 chunks of 8 cols)

    auto compute = [&](int s) {
        const char* bP = smem + s * STAGE_BYTES;
        const char* bQ = bP + 64 * 256;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r0 = (ks * 32 + trow) * 256, r1 = r0 + 16 * 256;
            bf16x8 fq[4], fp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int chunk = (wk * 8 + i * 2 + cq) ^ tsw;
                const bf16x4 lo = lds_tr16(bQ + r0 + chunk * 16 + sub8);
                const bf16x4 hi = lds_tr16(bQ + r1 + chunk * 16 + sub8);
                fq[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int chunk = (wn * 8 + j * 2 + cq) ^ tsw;
                const bf16x4 lo = lds_tr16(bP + r0 + chunk * 16 + sub8);
                const bf16x4 hi = lds_tr16(bP + r1 + chunk * 16 + sub8);
                fp[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[i], fp[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nt = (t_end - t_begin + 63) / 64;
    stage(0, t_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nt - 1; ++t) {
        stage(cur ^ 1, t_begin + (t + 1) * 64);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // lane holds OUT[n][k0..k0+3], n = tnb*128 + wn*64 + j*16 + l15, k0 = tkb*128 + wk*64 + i*16 + 4g
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = tnb * 128 + wn * 64 + j * 16 + l15;
        if (n >= p.n_store) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k0 = tkb * 128 + wk * 64 + i * 16 + 4 * g;
            if (k0 >= p.K) continue;
            float* o = p.OUT + (int64_t)n * p.ldo + k0;
            if (p.splits == 1) {
                float4 c = *(float4*)o;
                c.x += acc[i][j][0]; c.y += acc[i][j][1]; c.z += acc[i][j][2]; c.w += acc[i][j][3];
                *(float4*)o = c;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(o + r, acc[i][j][r]);
            }
        }
    }
}

}  // namespace

extern "C" int kzv_gemm_nt(const kzv_gemm_nt_args* a, int epilogue, void* stream) {
    if (!a || !a->A || !a->B || !a->C) return kzv_fail(KZV_E_ARG, "gemm_nt: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return kzv_fail(KZV_E_ARG, "gemm_nt: empty shape");
    if (a->K % BK) return kzv_fail(KZV_E_ARG, "gemm_nt: K must be a multiple of 64");
    if (a->N % 4 || a->ldc % 4) return kzv_fail(KZV_E_ARG, "gemm_nt: N and ldc must be multiples of 4");
    if (a->lda % 8 || a->ldb % 8) return kzv_fail(KZV_E_ARG, "gemm_nt: lda/ldb must be multiples of 8 (16-byte rows)");
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) return kzv_fail(KZV_E_ARG, "gemm_nt: operands must be 16-byte aligned");
    if (epilogue == KZV_EPI_RESID && (!a->resid || a->ldr % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt: RESID needs resid, ldr%4==0");
    if ((epilogue == KZV_EPI_GELU || epilogue == KZV_EPI_DGELU || epilogue == KZV_EPI_GELU_F32) && (!a->aux || a->ldaux % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt: GELU/DGELU need aux");
    NtParams p;
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.C = a->C; p.bias = a->bias; p.resid = a->resid;
    p.aux = (bf16_t*)a->aux; p.zero16 = kzv_zero_page();
    if (!p.zero16) return kzv_fail(KZV_E_HIP, "gemm_nt: zero page unavailable");
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldaux = a->ldaux;
    p.M = a->M; p.N = a->N; p.K = a->K; p.n_valid = a->n_valid > 0 ? a->n_valid : a->N;
    kzv_drop_params(a->drop_p, &p.drop_thr16, &p.drop_inv_keep);
    p.drop_key = a->drop_key;
    const int grid = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
    hipStream_t s = (hipStream_t)stream;
    KzvProfScope prof(0, 2.0 * a->M * p.n_valid * a->K, s);
#define KZV_NT_LAUNCH(E)                                                                                  \
    case E: {                                                                                             \
        static bool attr_done = false;                                                                    \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS); attr_done = true; } \
        hipLaunchKernelGGL(gemm_nt_kernel<E>, dim3(grid), dim3(256), NT_LDS, s, p);                       \
    } break;
    switch (epilogue) {
        KZV_NT_LAUNCH(KZV_EPI_BF16)
        KZV_NT_LAUNCH(KZV_EPI_F32)
        KZV_NT_LAUNCH(KZV_EPI_GELU)
        KZV_NT_LAUNCH(KZV_EPI_RESID)
        KZV_NT_LAUNCH(KZV_EPI_DGELU)
        KZV_NT_LAUNCH(KZV_EPI_GELU_F32)
        default: return kzv_fail(KZV_E_ARG, "gemm_nt: unknown epilogue");
    }
#undef KZV_NT_LAUNCH
    return kzv_check_launch("gemm_nt");
}

extern "C" int kzv_gemm_tn(const kzv_gemm_tn_args* a, void* stream) {
    if (!a || !a->P || !a->Q || !a->OUT) return kzv_fail(KZV_E_ARG, "gemm_tn: null operand");
    if (a->Mtok <= 0 || a->N <= 0 || a->K <= 0) return kzv_fail(KZV_E_ARG, "gemm_tn: empty shape");
    if (a->N % 8 || a->K % 8 || a->ldp % 8 || a->ldq % 8 || a->ldo % 4) return kzv_fail(KZV_E_ARG, "gemm_tn: N,K,ldp,ldq %8, ldo %4");
    if (((uintptr_t)a->P | (uintptr_t)a->Q | (uintptr_t)a->OUT) & 15) return kzv_fail(KZV_E_ARG, "gemm_tn: operands must be 16-byte aligned");
    TnParams p;
    p.P = (const bf16_t*)a->P; p.Q = (const bf16_t*)a->Q; p.OUT = a->OUT; p.zero16 = kzv_zero_page();
    if (!p.zero16) return kzv_fail(KZV_E_HIP, "gemm_tn: zero page unavailable");
    p.ldp = a->ldp; p.ldq = a->ldq; p.ldo = a->ldo;
    p.Mtok = a->Mtok; p.N = a->N; p.K = a->K; p.n_store = a->n_store > 0 ? a->n_store : a->N;
    const int tiles = ((a->N + 127) / 128) * ((a->K + 127) / 128);
    const int tok_tiles = (a->Mtok + 63) / 64;
    int splits = (768 + tiles - 1) / tiles;            // aim for ~3 workgroups per CU
    if (splits > tok_tiles) splits = tok_tiles;
    if (splits < 1) splits = 1;
    int chunk_tiles = (tok_tiles + splits - 1) / splits;
    splits = (tok_tiles + chunk_tiles - 1) / chunk_tiles;
    p.splits = splits; p.chunk = chunk_tiles * 64;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS); attr_done = true; }
    KzvProfScope prof(1, 2.0 * a->Mtok * p.n_store * a->K, (hipStream_t)stream);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles * splits), dim3(256), NT_LDS, (hipStream_t)stream, p);
    return kzv_check_launch("gemm_tn");
}
