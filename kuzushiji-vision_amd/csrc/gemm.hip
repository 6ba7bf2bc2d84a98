// bf16 MFMA GEMMs for gfx950 (MI355X): the C-ABI entry points and the 128x128 kernels.
//
//  gemm_nt : C[M,N]  = A[M,K] . B[N,K]^T  (+ fused epilogue)   -- every nn.Linear forward and dgrad
//  gemm_tn : O[N,K] += P[Mt,N]^T . Q[Mt,K]                      -- every weight gradient
//
// kzv_gemm_nt / kzv_gemm_tn first offer the shape to the 256x256 eight-phase kernels (gemm_nt256p.hip, gemm_nt256.hip,
// gemm_tn256.hip: every large GEMM of a ViT layer); what they decline -- the decoder's small GEMMs, K < 128, odd shapes --
// runs on the kernels in this file: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave = 4x4
// MFMA 16x16x32 tiles), reduction step 64, operands streamed global->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4)
// into a 2-stage ring, XOR-swizzled on the SOURCE address so the LDS image stays lane-linear while fragment reads are
// bank-conflict free; two workgroups per CU.
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
#include "gemm_nt.h"
#include "gemm_tn.h"
#include <cstdlib>

namespace {

constexpr int BK = 64;
constexpr int STAGE_BYTES = (128 + 128) * BK * 2;   // gemm_tn stage: 32 KiB
constexpr int NT_LDS = 2 * STAGE_BYTES;             // gemm_tn: 64 KiB -> 2 workgroups / CU

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}

// Workgroup = WM x WN waves, each wave a 64x64 output tile (4x4 MFMA 16x16x32 tiles); block tile
// BM = 64*WM rows by BN = 64*WN columns; NSTAGE-deep LDS ring of [BM + BN][64] bf16 stages filled by
// LDS-DMA.  NSTAGE-2 stages stay in flight ACROSS the per-step barrier (counted vmcnt + raw s_barrier).
// KB = reduction depth of one stage (64 or 32).  KB = 32 halves the ring (more workgroups per CU); its 64-byte
// rows use the swizzle chunk ^ perm[(row >> 2) & 3], perm = {0,2,3,1} (conflict-free for the ds_read_b128 lane
// groups, derived like the 128-byte-row case).  ASM_DMA issues the LDS-DMA from inline asm (needed for rings
// deeper than 2: hipcc must not see the in-flight DMA or it drains it before every LDS read).
template <int EPI, int WM, int WN, int NSTAGE, int KB, bool ASM_DMA>
__global__ __launch_bounds__(WM * WN * 64) void gemm_nt_kernel(const NtParams p) {
    constexpr int BM = WM * 64, BN = WN * 64, NW = WM * WN;
    constexpr int ROWB = KB * 2, SPR = ROWB / 16, RPP = 1024 / ROWB;   // row bytes, 16-B slots per row, rows per piece
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int PPW = (BM + BN) / RPP / NW;        // 1-KiB LDS-DMA pieces per wave per stage
    static_assert((BM + BN) / RPP % NW == 0, "pieces must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int wm = w / WN, wn = w - wm * WN;
    const int tilesN = (p.N + BN - 1) / BN, tilesM = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, tilesM * tilesN);
    const int tm = id / tilesN, tn = id - tm * tilesN;

    // ---- per-lane source pointers for the LDS-DMA pieces this wave issues per stage ----
    // piece pc covers stage rows pc*8 .. +7 (8 rows x 128 B; rows < BM belong to A, the rest to B);
    // lane -> row lane>>3, 16-B slot lane&7; slot s of row r holds source chunk s ^ (r & 7)
    // (involution; the fragment read applies the same XOR).
    const char* src[PPW]; int inc[PPW];
    {
        const int rl = lane / SPR;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = (w + j * NW) * RPP + rl;
            const int chunk = (lane % SPR) ^ (KB == 64 ? (row & 7) : ((0x1320 >> (((row >> 2) & 3) * 4)) & 3));
            if (row < BM) {
                const int am = tm * BM + row;
                const bool v = am < p.M;
                src[j] = v ? (const char*)(p.A + (int64_t)am * p.lda + chunk * 8) : (const char*)p.zero16;
                inc[j] = v ? ROWB : 0;
            } else {
                const int bn = tn * BN + row - BM;
                const bool v = bn < p.n_valid;
                src[j] = v ? (const char*)(p.B + (int64_t)bn * p.ldb + chunk * 8) : (const char*)p.zero16;
                inc[j] = v ? ROWB : 0;
            }
        }
    }
    auto stage = [&](int s) {
        char* d = smem + s * STAGE + w * 1024;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            if constexpr (ASM_DMA) glds16_asm_m0(src[j], d + j * NW * 1024); else glds16(src[j], d + j * NW * 1024);
            src[j] += inc[j];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a stage (row-major 128-B rows, XOR-swizzled 16-B slots)
    const int a_row_off = (wm * 64 + l15) * ROWB;
    const int b_row_off = BM * ROWB + (wn * 64 + l15) * ROWB;
    const int sw = KB == 64 ? (lane & 7) : ((0x1320 >> (((lane >> 2) & 3) * 4)) & 3);   // rows i*16 + l15: (row>>2)&3 == (l15>>2)

    auto compute = [&](int s) {
        const char* base = smem + s * STAGE;
#pragma unroll
        for (int ks = 0; ks < KB / 32; ++ks) {
            const int slot = ((ks * 4 + g) ^ sw) << 4;
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(base + a_row_off + i * 16 * ROWB + slot);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(base + b_row_off + j * 16 * ROWB + slot);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    // ring schedule: stages t+1 .. t+NSTAGE-2 are in flight while stage t is consumed.  The barrier at the
    // top of step t (a) publishes every wave's stage-t pieces (each wave waited for its own) and (b) proves
    // every wave finished reading buffer (t-1)%NSTAGE, which stage t+NSTAGE-1 then overwrites.
    const int nk = p.K / KB;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s);
    int buf = 0, nxt = NSTAGE - 1;
    for (int t = 0; t < nk; ++t) {
        if (t + NSTAGE - 2 < nk) wait_vmcnt<PPW * (NSTAGE - 2)>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NSTAGE - 1 < nk) stage(nxt);
        compute(buf);
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    // Accumulators (lane: C[m = i*16 + l15][n0 = j*16 + 4g .. +3]) go through LDS (the ring is free now) so
    // that global traffic is row-contiguous: one thread = 4 consecutive columns, 32 threads = one 128-column
    // row (256 B of bf16 / 512 B of fp32 per row per instruction, residual/aux loads included).
    // The tile is staged RC rows at a time (RC = all BM rows when the ring is large enough, else 64).
    constexpr int RC = NSTAGE * STAGE >= BM * BN * 4 ? BM : 64;
    static_assert(NSTAGE * STAGE >= RC * BN * 4, "LDS ring too small to stage 64 rows of the fp32 tile");
    float* tile = (float*)smem;                       // [RC][BN] fp32, 16-B chunks XOR-swizzled by (row & 31)
    constexpr int CPR = BN / 4, RSTEP = NW * 64 / CPR;
    const int c = tid % CPR;
    const int n0 = tn * BN + c * 4;
    const bool interior = tm * BM + BM <= p.M && tn * BN + BN <= p.N;      // wave-uniform
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI != KZV_EPI_DGELU && p.bias && n0 < p.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b4[r] = (n0 + r < p.n_valid) ? p.bias[n0 + r] : 0.f;
    }
#pragma unroll
  for (int r0 = 0; r0 < BM; r0 += RC) {
    __syncthreads();                                  // ring (or the previous chunk) is no longer being read
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = wm * 64 + i * 16 + l15 - r0;
        if (ml >= 0 && ml < RC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = wn * 16 + j * 4 + g;
                *(f32x4*)(tile + ml * BN + ((cc ^ (ml & 31)) << 2)) = acc[i][j];
            }
        }
    }
    __syncthreads();
    // rows are processed 8 at a time with the residual / aux loads of the batch issued up front, so the
    // per-row global-load latency overlaps instead of serialising 16 dependent round trips per tile.
    // Interior tiles are branch-free: per-row guards make hipcc's waitcnt pass lose count at every join and put
    // `s_waitcnt vmcnt(0)` in front of each store (every store then waits for the previous one to complete).
    constexpr int NP = RC / RSTEP;
    static_assert(NP % 8 == 0, "row passes come in batches of 8");
    if (interior) {
#pragma unroll
    for (int pb = 0; pb < NP; pb += 8) {
        float4 r4[8]; uint2 u2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int m = tm * BM + r0 + tid / CPR + (pb + q) * RSTEP;
            if (EPI == KZV_EPI_RESID) r4[q] = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
            if (EPI == KZV_EPI_DGELU) u2[q] = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int rr = tid / CPR + (pb + q) * RSTEP;
            const int m = tm * BM + r0 + rr;
            const f32x4 a4 = *(const f32x4*)(tile + rr * BN + ((c ^ (rr & 31)) << 2));
            float v[4] = {a4[0] + b4[0], a4[1] + b4[1], a4[2] + b4[2], a4[3] + b4[3]};
            nt_emit<EPI>(p, m, n0, v, r4[q], u2[q]);
        }
    }
    } else if (n0 < p.N) {
#pragma unroll 1
    for (int q = 0; q < NP; ++q) {
        const int rr = tid / CPR + q * RSTEP;
        const int m = tm * BM + r0 + rr;
        if (m >= p.M) continue;
        float4 r4 = make_float4(0, 0, 0, 0); uint2 u2 = make_uint2(0, 0);
        if (EPI == KZV_EPI_RESID) r4 = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
        if (EPI == KZV_EPI_DGELU) u2 = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
        const f32x4 a4 = *(const f32x4*)(tile + rr * BN + ((c ^ (rr & 31)) << 2));
        float v[4] = {a4[0] + b4[0], a4[1] + b4[1], a4[2] + b4[2], a4[3] + b4[3]};
        nt_emit<EPI>(p, m, n0, v, r4, u2);
    }
    }
  }     // row chunks
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: OUT[n][k] += sum_t P[t][n] * Q[t][k].   LDS tiles are [64 tokens][128 cols] (256-B rows).
// Reduction index inside a 32-token MFMA step is permuted (element j of lane group g <-> token
// 16*(j>>2) + 4g + (j&3)); both operands use the same permutation so the sum is unchanged, and the two
// groups of a 32-lane half then read 8 consecutive rows -> the XOR swizzle below is conflict-free.
__device__ __forceinline__ void gemm_tn_body(const TnParams& p, const int bid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int wn = w >> 1, wk = w & 1;     // wave tile: n in [wn*64,+64), k in [wk*64,+64)
    const int tilesN = (p.N + 127) / 128, tilesK = (p.K + 127) / 128;
    const int per = tilesN * tilesK;
    const int id = xcd_remap(bid, per * p.splits);
    const int split = id / per, rem = id - split * per;
    const int tnb = rem / tilesK, tkb = rem - tnb * tilesK;
    const int t_begin = split * p.chunk;
    const int t_end = min(p.Mtok, t_begin + p.chunk);
    if (t_begin >= t_end) return;

    // staging: piece j of wave w covers tile rows (w*4 + j)*4 .. +3 (4 rows x 256 B); lane -> row lane>>4,
    // slot lane&15; slot s of row r holds source chunk s ^ ((r & 7) << 1).
    // Columns beyond N / K read the zero page with stride 0, so the steady-state issue is pointer + increment only.
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
            pp[j] = vp ? (const char*)(p.P + (int64_t)(t_begin + row) * p.ldp + ncol) : (const char*)p.zero16;
            pq[j] = vq ? (const char*)(p.Q + (int64_t)(t_begin + row) * p.ldq + kcol) : (const char*)p.zero16;
            sp[j] = vp ? 64 * p.ldp * 2 : 0; sq[j] = vq ? 64 * p.ldq * 2 : 0;
        }
    }
    auto stage = [&](int s) {                       // all 64 token rows of the stage are inside [t_begin, t_end)
        char* dP = smem + s * STAGE_BYTES + (w * 16) * 256;
        char* dQ = dP + 64 * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) { glds16_asm_m0(pp[j], dP + j * 1024); pp[j] += sp[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { glds16_asm_m0(pq[j], dQ + j * 1024); pq[j] += sq[j]; }
    };
    auto stage_tail = [&](int s, int t0) {          // the ragged last stage of the token range: rows >= t_end read zeros
        char* dP = smem + s * STAGE_BYTES + (w * 16) * 256;
        char* dQ = dP + 64 * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_asm_m0(t0 + rowj[j] < t_end ? pp[j] : (const char*)p.zero16, dP + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_asm_m0(t0 + rowj[j] < t_end ? pq[j] : (const char*)p.zero16, dQ + j * 1024);
    };

    f32x4 acc[4][4];   // [k-tile i][n-tile j]: D rows = n (4g+r), D cols = k (l15)
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

    // Bias gradient for free: column sums of P are one more MFMA against a vector of ones.  Every (n-tile, split)
    // is visited by tilesK workgroups (one per k-tile) that stream the same P panel, so workgroup tkb takes the
    // reduction steps t with t % tilesK == tkb: each token is summed exactly once and the extra MFMAs (4 per 32
    // tokens, waves with wk == 0 only) spread evenly over the grid.
    f32x4 bacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = (bf16x8){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bool bias_wave = p.dbias != nullptr && wk == 0;

    auto compute = [&](int s, bool with_bias) {
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp[j], fq[i], acc[i][j], 0, 0, 0);
            if (with_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp[j], ones, bacc[j], 0, 0, 0);
            }
        }
    };

    const int nt = (t_end - t_begin + 63) / 64;
    const int nfull = (t_end - t_begin) / 64;       // stages 0 .. nfull-1 are complete; stage nfull (if any) is ragged
    if (nfull > 0) stage(0); else stage_tail(0, t_begin);
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of stage t have landed
        __builtin_amdgcn_s_barrier();                        // ... and everybody's; buffer cur^1 is free again
        asm volatile("" ::: "memory");
        if (t + 1 < nfull) stage(cur ^ 1);
        else if (t + 1 < nt) stage_tail(cur ^ 1, t_begin + (t + 1) * 64);
        compute(cur, bias_wave && (t % tilesK) == tkb);
        cur ^= 1;
    }
    if (bias_wave && l15 == 0) {     // every column of bacc holds the same sums; column 0 publishes them
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = tnb * 128 + wn * 64 + j * 16 + 4 * g + r;
                if (n < p.n_store) atomicAdd(p.dbias + n, bacc[j][r]);
            }
    }

    // ---- epilogue: accumulators -> LDS (fp32 [128 n][128 k], 16-B chunks XOR-swizzled by row) -> each
    // wave-instruction adds ONE 256-byte row segment (64 consecutive k) to OUT: the shape global float
    // atomics run at full rate with (MI355X_MICROARCH.md, Global float atomics).
    __syncthreads();                                   // every wave is done reading the operand stages
    float* tile = (float*)smem;                        // 128 * 128 * 4 B = 64 KiB = the whole ring
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = wn * 64 + j * 16 + 4 * g + r, k = wk * 64 + i * 16 + l15;
                tile[n * 128 + ((((k >> 2) ^ (n & 31)) << 2) | (k & 3))] = acc[i][j][r];
            }
    __syncthreads();
    // 8 rows per batch: the LDS reads of a batch are issued together (one latency per batch, not per row), and the
    // interior case is branch-free so the adds stream out back to back
    const bool interior = tnb * 128 + 128 <= p.n_store && tkb * 128 + 128 <= p.K;     // wave-uniform
    for (int it0 = 0; it0 < 64; it0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int it = it0 + u;
            const int n = w * 32 + (it >> 1), k = (it & 1) * 64 + lane;
            v[u] = tile[n * 128 + ((((k >> 2) ^ (n & 31)) << 2) | (k & 3))];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int it = it0 + u;
            const int n = w * 32 + (it >> 1), k = (it & 1) * 64 + lane;
            const int gn = tnb * 128 + n, gk = tkb * 128 + k;
            float* o = p.OUT + (int64_t)gn * p.ldo + gk;
            if (interior && p.splits > 1) atomicAdd(o, v[u]);
            else if (gn < p.n_store && gk < p.K) { if (p.splits == 1) *o += v[u]; else atomicAdd(o, v[u]); }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) { gemm_tn_body(p, blockIdx.x); }

// Grouped launch: several independent weight gradients (e.g. the six of one decoder layer, 4..12 tiles of 128x128 each)
// share ONE grid, so the 256 CUs x 2 resident workgroups are filled in a single round instead of six part-filled ones.
constexpr int TN_GROUP_MAX = 40;     // 88-byte descriptors in the kernel arguments (4 KiB): the 36 weight gradients of a six-layer decoder in one grid
struct TnGroup { TnParams p[TN_GROUP_MAX]; int start[TN_GROUP_MAX + 1]; int n; };
__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(const TnGroup g) {
    int i = 0;
#pragma unroll 1
    while (i + 1 < g.n && (int)blockIdx.x >= g.start[i + 1]) ++i;
    gemm_tn_body(g.p[i], (int)blockIdx.x - g.start[i]);
}

}  // namespace

// K-loop schedule of the persistent 256x256 kernel: 0 eight-phase ping-pong (gemm_nt256p.hip), 1 free-running (gemm_nt256f.hip);
// bit 1 (2, 3): the epilogues of g_nt_half_mask go to the four-wave 256x128 kernel, two workgroups per CU (gemm_nt256h.hip)
static int g_nt_schedule = -1;
static int g_nt_half_mask = -1;
static int nt_schedule() {
    if (g_nt_schedule < 0) {
        const char* e = getenv("KZV_NT_FREE"); g_nt_schedule = e ? (atoi(e) != 0) : 0;
        const char* h = getenv("KZV_NT_HALF"); if (h && atoi(h)) g_nt_schedule |= 2;
    }
    return g_nt_schedule;
}
static int nt_half_mask() {
    if (g_nt_half_mask < 0) { const char* e = getenv("KZV_NT_HALF_EPIS"); g_nt_half_mask = e ? atoi(e) : 0x3f; }
    return g_nt_half_mask;
}
extern "C" int kzv_set_nt_schedule(int n) { g_nt_schedule = n < 0 ? -1 : (n & 3); return KZV_OK; }
extern "C" int kzv_set_nt_half_epilogues(int mask) { g_nt_half_mask = mask; return KZV_OK; }

// validation of a kzv_gemm_nt call and its kernel parameter block (shared with the dgrad + wgrad pair launch of gemm_tn256.hip)
int kzv_nt_params(const kzv_gemm_nt_args* a, int epilogue, NtParams* out) {
    if (!a || !a->A || !a->B || !a->C) return kzv_fail(KZV_E_ARG, "gemm_nt: null operand");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return kzv_fail(KZV_E_ARG, "gemm_nt: empty shape");
    if (a->K % 64) return kzv_fail(KZV_E_ARG, "gemm_nt: K must be a multiple of 64");
    if (a->N % 4 || a->ldc % 4) return kzv_fail(KZV_E_ARG, "gemm_nt: N and ldc must be multiples of 4");
    if (a->lda % 8 || a->ldb % 8) return kzv_fail(KZV_E_ARG, "gemm_nt: lda/ldb must be multiples of 8 (16-byte rows)");
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) return kzv_fail(KZV_E_ARG, "gemm_nt: operands must be 16-byte aligned");
    if (epilogue == KZV_EPI_RESID && (!a->resid || a->ldr % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt: RESID needs resid, ldr%4==0");
    if ((epilogue == KZV_EPI_GELU || epilogue == KZV_EPI_DGELU || epilogue == KZV_EPI_GELU_F32) && (!a->aux || a->ldaux % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt: GELU/DGELU need aux");
    NtParams p{};
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.C = a->C; p.bias = a->bias; p.resid = a->resid;
    p.aux = (bf16_t*)a->aux; p.zero16 = kzv_zero_page();
    if (!p.zero16) return kzv_fail(KZV_E_HIP, "gemm_nt: zero page unavailable");
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldaux = a->ldaux;
    p.M = a->M; p.N = a->N; p.K = a->K; p.n_valid = a->n_valid > 0 ? a->n_valid : a->N;
    kzv_drop_params(a->drop_p, &p.drop_thr16, &p.drop_inv_keep);
    p.drop_key = a->drop_key;
    p.strip = kzv_nt_strip() & 0xff;
    *out = p;
    return KZV_OK;
}

extern "C" int kzv_gemm_nt(const kzv_gemm_nt_args* a, int epilogue, void* stream) {
    NtParams p;
    { const int rc = kzv_nt_params(a, epilogue, &p); if (rc != KZV_OK) return rc; }
    hipStream_t s = (hipStream_t)stream;
    KzvProfScope prof(0, 2.0 * a->M * p.n_valid * a->K, s);
    // large shapes: 256x256 eight-phase kernels.  The persistent one wins wherever its per-wave drain is light (one
    // store per element: -2..-10 % vs the one-tile-per-workgroup kernel); the two-store GELU epilogues are faster through
    // the workgroup-wide LDS epilogue of gemm_nt256.hip (512-B / 1-KiB row segments).
    static int use_p = -1;
    if (use_p < 0) { const char* e = getenv("KZV_NT256P"); use_p = e ? atoi(e) : 1; }
    if (kzv_rows_launch(p, epilogue, s)) return kzv_check_launch("gemm_nt");      // M <= 1024: the generation step's GEMMs
    static int p_gelu = -1;      // dev knob: the two-store GELU epilogues on the persistent kernel too (A/B; default off)
    if (p_gelu < 0) { const char* e = getenv("KZV_NT256P_GELU"); p_gelu = e ? atoi(e) : 0; }
    const bool two_store = !p_gelu && (epilogue == KZV_EPI_GELU || epilogue == KZV_EPI_GELU_F32);
    static int half_maxk = -1;   // the four-wave kernel's K loop is latency-bound (three-group ring): it pays only where the drain is a large part of a tile
    if (half_maxk < 0) { const char* e = getenv("KZV_NTH_MAX_K"); half_maxk = e ? atoi(e) : 1 << 30; }
    if ((nt_schedule() & 2) && ((nt_half_mask() >> epilogue) & 1) && p.K <= half_maxk && kzv_cu_reserve() == 0 && kzv_nt256h_launch(p, epilogue, s)) return kzv_check_launch("gemm_nt");
    if ((nt_schedule() & 1) && use_p && !two_store && kzv_cu_reserve() == 0 && kzv_nt256f_launch(p, epilogue, s)) return kzv_check_launch("gemm_nt");
    if (use_p && !two_store && kzv_cu_reserve() == 0 && kzv_nt256p_launch(p, epilogue, s)) return kzv_check_launch("gemm_nt");
    if (kzv_nt256_launch(p, epilogue, s)) return kzv_check_launch("gemm_nt");
#define KZV_NT_CASE(E, WM, WN, NS, KB, AD)                                                                \
    case E: {                                                                                             \
        constexpr int lds = NS * (WM + WN) * 64 * KB * 2;                                                 \
        static bool attr_done = false;                                                                    \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<E, WM, WN, NS, KB, AD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; } \
        const int grid = ((a->M + WM * 64 - 1) / (WM * 64)) * ((a->N + WN * 64 - 1) / (WN * 64));         \
        hipLaunchKernelGGL((gemm_nt_kernel<E, WM, WN, NS, KB, AD>), dim3(grid), dim3(WM * WN * 64), lds, s, p); \
    } break;
#define KZV_NT_VARIANT(WM, WN, NS, KB, AD)                                                                \
    switch (epilogue) {                                                                                   \
        KZV_NT_CASE(KZV_EPI_BF16, WM, WN, NS, KB, AD) KZV_NT_CASE(KZV_EPI_F32, WM, WN, NS, KB, AD)        \
        KZV_NT_CASE(KZV_EPI_GELU, WM, WN, NS, KB, AD) KZV_NT_CASE(KZV_EPI_RESID, WM, WN, NS, KB, AD)      \
        KZV_NT_CASE(KZV_EPI_DGELU, WM, WN, NS, KB, AD) KZV_NT_CASE(KZV_EPI_GELU_F32, WM, WN, NS, KB, AD)  \
        default: return kzv_fail(KZV_E_ARG, "gemm_nt: unknown epilogue");                                 \
    }
    // Measured on MI355X (tools/dev/gemm_bench.py, round 1): 128x128x64 / 2-stage ring / 2 workgroups per CU is the
    // fastest of the variants this template expresses (KB = 32 with 2-4 stages: -5..-15 %; 256x128 tiles at one
    // workgroup per CU: -3 %; a 3-stage ring at one workgroup per CU: -30 %), so only it is instantiated.
    KZV_NT_VARIANT(2, 2, 2, 64, false)
#undef KZV_NT_VARIANT
#undef KZV_NT_CASE
    return kzv_check_launch("gemm_nt");
}

// the generation step's GEMMs with the LayerNorm of their A operand and / or residual folded in (gemm_rows.hip)
extern "C" int kzv_gemm_rows_ln(const kzv_gemm_rows_ln_args* a, int epilogue, void* stream) {
    if (!a || !a->B || !a->C) return kzv_fail(KZV_E_ARG, "gemm_rows_ln: null operand");
    if (!a->ln_a && !a->A) return kzv_fail(KZV_E_ARG, "gemm_rows_ln: needs A or ln_a");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->N % 4 || a->ldc % 4 || a->ldb % 8 || (a->A && a->lda % 8)) return kzv_fail(KZV_E_ARG, "gemm_rows_ln: bad shape");
    if (a->ln_a && (!a->ln_a_gamma || !a->ln_a_beta)) return kzv_fail(KZV_E_ARG, "gemm_rows_ln: ln_a needs gamma and beta");
    if (a->ln_r && (!a->ln_r_gamma || !a->ln_r_beta)) return kzv_fail(KZV_E_ARG, "gemm_rows_ln: ln_r needs gamma and beta");
    if ((epilogue == KZV_EPI_GELU || epilogue == KZV_EPI_GELU_F32) && (!a->aux || a->ldaux % 4)) return kzv_fail(KZV_E_ARG, "gemm_rows_ln: GELU needs aux");
    NtParams p{};
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.C = a->C; p.bias = a->bias; p.aux = (bf16_t*)a->aux;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldc; p.ldaux = a->ldaux;
    p.M = a->M; p.N = a->N; p.K = a->K; p.n_valid = a->n_valid > 0 ? a->n_valid : a->N;
    p.drop_inv_keep = 1.f;
    return kzv_rows_ln_launch(p, epilogue, a->ln_a, a->ln_a_gamma, a->ln_a_beta, a->ln_r, a->ln_r_gamma, a->ln_r_beta, a->eps, (hipStream_t)stream);
}

// fp8 (e4m3) operands with per-row scales, block-scaled MFMA at twice the bf16 rate (gemm_nt256p.hip).
extern "C" int kzv_gemm_nt_fp8(const kzv_gemm_nt_fp8_args* a, int epilogue, void* stream) {
    if (!a || !a->A || !a->B || !a->C || !a->a_scale || !a->b_scale) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: null operand or scale");
    if (a->M <= 0 || a->N <= 0 || a->K <= 0) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: empty shape");
    if (a->K % 256) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: K must be a multiple of 256");
    if (a->N % 4 || a->ldc % 4) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: N and ldc must be multiples of 4");
    if (a->lda % 16 || a->ldb % 16) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: lda/ldb must be multiples of 16 (16-byte rows)");
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: operands must be 16-byte aligned");
    if (epilogue != KZV_EPI_BF16 && epilogue != KZV_EPI_GELU && epilogue != KZV_EPI_RESID && epilogue != KZV_EPI_DGELU)
        return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: epilogue must be BF16, GELU, RESID or DGELU");
    if (epilogue == KZV_EPI_DGELU && (!a->aux || a->ldaux % 4 || !a->c8 || !a->c8_rowq || a->ldc8 % 4))
        return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: DGELU needs aux, c8 and c8_rowq");
    if (epilogue == KZV_EPI_RESID && (!a->resid || a->ldr % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: RESID needs resid, ldr%4==0");
    if (epilogue == KZV_EPI_GELU && (!a->aux || a->ldaux % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: GELU needs aux");
    if (epilogue == KZV_EPI_GELU && a->c8 && (!a->c8_qscale || !a->c8_amax || a->ldc8 % 4)) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: c8 needs c8_qscale, c8_amax, ldc8%4==0");
    NtParams p{};
    p.A = (const bf16_t*)a->A; p.B = (const bf16_t*)a->B; p.C = a->C; p.bias = a->bias; p.resid = a->resid;
    p.aux = (bf16_t*)a->aux; p.zero16 = kzv_zero_page();
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldaux = a->ldaux;
    p.M = a->M; p.N = a->N; p.K = a->K; p.n_valid = a->n_valid > 0 ? a->n_valid : a->N;
    kzv_drop_params(a->drop_p, &p.drop_thr16, &p.drop_inv_keep);
    p.drop_key = a->drop_key;
    p.strip = kzv_nt_strip() & 0xff;
    p.a_scale = a->a_scale; p.b_scale = a->b_scale;
    p.c8 = (unsigned char*)a->c8; p.ldc8 = a->ldc8; p.c8_qscale = a->c8_qscale; p.c8_amax = a->c8_amax; p.c8_rowq = a->c8_rowq;
    hipStream_t s = (hipStream_t)stream;
    KzvProfScope prof(4, 2.0 * a->M * p.n_valid * a->K, s);
    const int rc = kzv_nt256p_fp8_launch(p, epilogue, s);
    if (rc != KZV_OK) return rc;
    return kzv_check_launch("gemm_nt_fp8");
}

static int tn_fill(const kzv_gemm_tn_args* a, TnParams& p) {
    if (!a || !a->P || !a->Q || !a->OUT) return kzv_fail(KZV_E_ARG, "gemm_tn: null operand");
    if (a->Mtok <= 0 || a->N <= 0 || a->K <= 0) return kzv_fail(KZV_E_ARG, "gemm_tn: empty shape");
    if (a->N % 8 || a->K % 8 || a->ldp % 8 || a->ldq % 8 || a->ldo % 4) return kzv_fail(KZV_E_ARG, "gemm_tn: N,K,ldp,ldq %8, ldo %4");
    if (((uintptr_t)a->P | (uintptr_t)a->Q | (uintptr_t)a->OUT) & 15) return kzv_fail(KZV_E_ARG, "gemm_tn: operands must be 16-byte aligned");
    p.P = (const bf16_t*)a->P; p.Q = (const bf16_t*)a->Q; p.OUT = a->OUT; p.zero16 = kzv_zero_page();
    if (!p.zero16) return kzv_fail(KZV_E_HIP, "gemm_tn: zero page unavailable");
    p.ldp = a->ldp; p.ldq = a->ldq; p.ldo = a->ldo;
    p.Mtok = a->Mtok; p.N = a->N; p.K = a->K; p.n_store = a->n_store > 0 ? a->n_store : a->N;
    p.dbias = a->dbias;
    return KZV_OK;
}
// Token splits of one problem given the workgroup budget `target` it may fill: ONE resident round without overshooting
// it, and >= min_steps reduction steps per workgroup so the 64 float atomics per lane of the epilogue (issue-bound, and
// contended when many splits hit one small output) stay a small share.  Returns the workgroup count.
static int tn_plan(TnParams& p, int target) {
    static int min_steps = -1;
    if (min_steps < 0) { const char* e = getenv("KZV_TN_MINSTEPS"); min_steps = e ? atoi(e) : 16; }
    const int tiles = ((p.N + 127) / 128) * ((p.K + 127) / 128);
    const int tok_tiles = (p.Mtok + 63) / 64;
    int splits = target / tiles;
    if (splits > tok_tiles / min_steps) splits = tok_tiles / min_steps;
    if (splits > tok_tiles) splits = tok_tiles;
    if (splits < 1) splits = 1;
    const int chunk_tiles = (tok_tiles + splits - 1) / splits;
    splits = (tok_tiles + chunk_tiles - 1) / chunk_tiles;
    p.splits = splits; p.chunk = chunk_tiles * 64;
    return tiles * splits;
}
static int tn_target() {
    static int target = -1;
    if (target < 0) { const char* e = getenv("KZV_TN_BLOCKS"); target = e ? atoi(e) : 512; }
    return target;
}

extern "C" int kzv_gemm_tn(const kzv_gemm_tn_args* a, void* stream) {
    TnParams p;
    const int rc = tn_fill(a, p);
    if (rc != KZV_OK) return rc;
    KzvProfScope prof(1, 2.0 * a->Mtok * p.n_store * a->K, (hipStream_t)stream);
    if (kzv_tn256_launch(p, (hipStream_t)stream)) return kzv_check_launch("gemm_tn");     // >= 9 tiles of 256x256: eight-phase kernel
    const int blocks = tn_plan(p, tn_target());
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS); attr_done = true; }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(blocks), dim3(256), NT_LDS, (hipStream_t)stream, p);
    return kzv_check_launch("gemm_tn");
}

// n independent weight gradients in one grid (model.cpp: the six of a decoder layer; head; cross-K/V + projection).
// Problems the 256x256 kernel would take are launched on their own.  The workgroup budget (one resident round) is shared
// in proportion to the tile counts.
int kzv_gemm_tn_group(const kzv_gemm_tn_args* a, int n, hipStream_t s) {
    if (n <= 0) return KZV_OK;
    if (n == 1 || n > TN_GROUP_MAX) {
        for (int i = 0; i < n; ++i) { const int rc = kzv_gemm_tn(a + i, s); if (rc != KZV_OK) return rc; }
        return KZV_OK;
    }
    TnGroup g;
    g.n = 0;
    int tiles_total = 0;
    double work = 0;
    for (int i = 0; i < n; ++i) {
        TnParams p;
        const int rc = tn_fill(a + i, p);
        if (rc != KZV_OK) return rc;
        const int t256 = ((p.n_store + 255) / 256) * ((p.K + 255) / 256);
        if (t256 >= 24) { const int r2 = kzv_gemm_tn(a + i, s); if (r2 != KZV_OK) return r2; continue; }   // the eight-phase kernel's shapes
        g.p[g.n++] = p;
        tiles_total += ((p.N + 127) / 128) * ((p.K + 127) / 128);
        work += 2.0 * p.Mtok * p.n_store * p.K;
    }
    if (g.n == 0) return KZV_OK;
    KzvProfScope prof(1, work, s);
    int blocks = 0;
    for (int i = 0; i < g.n; ++i) {
        const int tiles = ((g.p[i].N + 127) / 128) * ((g.p[i].K + 127) / 128);
        // one resident round (512 workgroups) for the small groups; a group with more tiles than that round's half (the 36 weight
        // gradients of the decoder's layers: 288 tiles) takes two: 3 token splits of 80 stages, 261 us against 335 with one split per
        // tile and 348 for six launches of ten splits (tools/dev/r5_tnblocks.sh)
        const int target = tiles_total >= 256 ? 2 * tn_target() : tn_target();
        int share = (int)((int64_t)target * tiles / tiles_total);
        if (share < tiles) share = tiles;
        g.start[i] = blocks;
        blocks += tn_plan(g.p[i], share);
    }
    g.start[g.n] = blocks;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_tn_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS); attr_done = true; }
    hipLaunchKernelGGL(gemm_tn_group_kernel, dim3(blocks), dim3(256), NT_LDS, s, g);
    return kzv_check_launch("gemm_tn_group");
}
