// gemm_nt, 256x256 tile, eight-phase ping-pong schedule for gfx950 (MI355X).
//
//  C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue), A and B bf16 with k contiguous.
//
// One 512-thread workgroup per CU (8 waves, 2 per SIMD) owns a 256x256 output tile; wave (wr, wc) owns rows
// wr*128..+127, columns wc*64..+63 (8x4 MFMA 16x16x32 accumulators = 128 registers).  Reduction step 64.
//
// LDS (128 KiB): two K-tile buffers, each four 16-KiB HALF-TILES of 128 rows x 128 B:
//     A-h0 / A-h1 : local row r -> tile row (r>>6)*128 + h*64 + (r&63)   (the h-th 64-row half of BOTH wave rows)
//     B-h0 / B-h1 : local row r -> tile col (r>>5)*64  + h*32 + (r&31)   (the h-th 32-col half of ALL four wave cols)
// so a K-tile is consumed in four PHASES, one accumulator quadrant (64 rows x 32 cols, 16 MFMAs) each:
//     p1: read A-h0,B-h0 -> (A0,B0)   p2: read B-h1 -> (A0,B1)   p3: read A-h1 -> (A1,B1)   p4: (A1,B0)
// and every half-tile has ONE last-read phase, after which its slot can be refilled for K-tile t+2.  Each phase
// also issues one half-tile of LDS-DMA (2 x global_load_lds_dwordx4 per lane):
//     p1: B-h1(t+1)   p2: A-h1(t+1)   p3: A-h0(t+2)   p4: B-h0(t+2)
// i.e. a slot is refilled >= 2 phases after its last read (WAR, also across the stagger below) and every half-tile
// is issued 4..5 phases (one whole K-tile of MFMA time) before the `s_waitcnt vmcnt(8)` that retires it; the wait
// sits BEFORE the phase's first barrier and the first read of that data is in the NEXT phase (RAW through a
// barrier every wave has passed after its own wait).  vmcnt is never 0 inside the steady loop.
//
// Ping-pong: waves 4..7 execute one extra s_barrier up front, so they run one barrier interval behind waves 0..3;
// each phase is [reads + DMA issue] barrier [16 MFMA] barrier, hence while one wave of a SIMD runs its MFMAs
// the other issues its LDS reads and loads (cdna_hip_programming.md "256^2 8-phase template").
//
// Swizzle: slot s of local row r holds source chunk s ^ (r & 7) (applied to the SOURCE address of the DMA and to
// the fragment read; the LDS image stays lane-linear).  Epilogue: as gemm.hip (accumulators -> LDS -> row
// contiguous global traffic), two chunks of 128 rows.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "gemm_nt.h"
#include <cstdlib>

namespace {

constexpr int HT_BYTES = 128 * 128;        // half-tile: 128 rows x 64 bf16
constexpr int BUF_BYTES = 4 * HT_BYTES;    // A-h0, A-h1, B-h0, B-h1
constexpr int LDS_BYTES = 2 * BUF_BYTES;   // 128 KiB
constexpr int KA0 = 0, KA1 = 1, KB0 = 2, KB1 = 3;

// LDS-DMA, 16 B per lane: source = sbase (wave-uniform) + voff (per lane), destination = lds_dst + 16*lane.
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(const NtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wr = w >> 2, wc = w & 3;
    const int tilesN = (p.N + 255) / 256, tilesM = (p.M + 255) / 256;
    const int id = xcd_remap(blockIdx.x, tilesM * tilesN);
    int tm, tn;
    nt_tile_coords(id, tilesM, tilesN, p.strip, tm, tn);

    // ---- LDS-DMA sources: wave w fills 1-KiB pieces w and w+8 of every half-tile (8 rows x 128 B each) ----
    // Rows beyond M / columns beyond n_valid are clamped to a valid row: their products are never stored.
    const char* baseA = (const char*)(p.A + (int64_t)tm * 256 * p.lda);
    const char* baseB = (const char*)p.B;          // B offsets are absolute: a tile may start beyond n_valid
    unsigned voff[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = j * 64 + w * 8 + (lane >> 3);
        const unsigned cb = (unsigned)(((lane & 7) ^ (r & 7)) * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int arow = (r >> 6) * 128 + h * 64 + (r & 63);
            arow = max(0, min(tm * 256 + arow, p.M - 1) - tm * 256);
            voff[KA0 + h][j] = (unsigned)arow * (unsigned)(p.lda * 2) + cb;
            int bcol = (r >> 5) * 64 + h * 32 + (r & 31);
            bcol = min(tn * 256 + bcol, p.n_valid - 1);
            voff[KB0 + h][j] = (unsigned)bcol * (unsigned)(p.ldb * 2) + cb;
        }
    }
    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)smem) + (unsigned)w * 1024u);
    auto stage = [&](int buf, int kind, int kt) {
        const char* sb = (kind < 2 ? baseA : baseB) + (int64_t)kt * 128;
        const unsigned d = ldsw + (unsigned)(buf * BUF_BYTES + kind * HT_BYTES);
        glds16_s(voff[kind][0], sb, d);
        glds16_s(voff[kind][1], sb, d + 8192u);
    };

    const int nk = p.K / 64;                       // >= 2 (checked by the launcher)
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment reads: lane supplies row l15 of a 16-row fragment, k = 8*g + 32*kh .. +7 (one 16-B slot)
    const int sw = l15 & 7;
    const int slot0 = (g ^ sw) << 4, slot1 = ((4 + g) ^ sw) << 4;
    const int a_off = (wr * 64 + l15) * 128, b_off = (wc * 32 + l15) * 128;
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    auto readA = [&](int buf, int mh) {
        const char* b = smem + buf * BUF_BYTES + (KA0 + mh) * HT_BYTES + a_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *(const bf16x8*)(b + i * 2048 + slot0);
            fa[i][1] = *(const bf16x8*)(b + i * 2048 + slot1);
        }
    };
    auto readB = [&](int buf, int nh, bf16x8 (&fb)[2][2]) {
        const char* b = smem + buf * BUF_BYTES + (KB0 + nh) * HT_BYTES + b_off;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = *(const bf16x8*)(b + j * 2048 + slot0);
            fb[j][1] = *(const bf16x8*)(b + j * 2048 + slot1);
        }
    };
    // B fragment first: every lane ends up with 4 consecutive output COLUMNS of one row (as gemm.hip)
    auto mma = [&](int mh, int nh, const bf16x8 (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kh], fa[i][kh], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // One K-tile = four phases.  s1 / s2: K-tiles t+1 / t+2 exist (wave-uniform); the tail tiles skip the refills
    // and tighten the waits to what is still outstanding.  One loop body for steady state and tail keeps the
    // accumulators in place (separate tail copies made hipcc shuffle and spill them).
    auto ktile = [&](auto bufc, int t) {
        constexpr int BUF = decltype(bufc)::value;
        const bool s1 = t + 1 < nk, s2 = t + 2 < nk;
        // p1
        readA(BUF, 0); readB(BUF, 0, fb0);
        if (s1) { stage(BUF ^ 1, KB1, t + 1); vmcnt<8>(); } else vmcnt<2>();      // retires B-h1(t), read in p2
        __builtin_amdgcn_s_barrier();
        mma(0, 0, fb0);
        __builtin_amdgcn_s_barrier();
        // p2
        readB(BUF, 1, fb1);
        if (s1) { stage(BUF ^ 1, KA1, t + 1); vmcnt<8>(); } else vmcnt<0>();      // retires A-h1(t), read in p3
        __builtin_amdgcn_s_barrier();
        mma(0, 1, fb1);
        __builtin_amdgcn_s_barrier();
        // p3
        readA(BUF, 1);
        if (s2) stage(BUF, KA0, t + 2);
        __builtin_amdgcn_s_barrier();
        mma(1, 1, fb1);
        __builtin_amdgcn_s_barrier();
        // p4
        if (s2) { stage(BUF, KB0, t + 2); vmcnt<8>(); } else if (s1) vmcnt<4>();  // retires A-h0(t+1), B-h0(t+1)
        __builtin_amdgcn_s_barrier();
        mma(1, 0, fb0);
        __builtin_amdgcn_s_barrier();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    stage(0, KA0, 0); stage(0, KB0, 0); stage(0, KB1, 0); stage(0, KA1, 0); stage(1, KA0, 1); stage(1, KB0, 1);
    vmcnt<8>();                                    // A-h0(0), B-h0(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();     // waves 4..7 run one barrier interval behind
    for (int t = 0; t < nk; t += 2) {
        ktile(I0{}, t);
        if (t + 1 < nk) ktile(I1{}, t + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();     // balance the barrier count

    // ---- epilogue: two chunks of 128 rows (chunk c = accumulator rows mh == c of both wave rows) ----
    float* tile = (float*)smem;                    // [128][256] fp32, 16-B chunks XOR-swizzled by (row & 31)
    const int c4 = tid & 63;
    const int n0 = tn * 256 + c4 * 4;
    const bool interior = tm * 256 + 256 <= p.M && tn * 256 + 256 <= p.n_valid;    // wave-uniform (n_valid <= N)
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    bool nv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        nv[r] = n0 + r < p.n_valid;
        if (EPI != KZV_EPI_DGELU && p.bias && nv[r]) b4[r] = p.bias[n0 + r];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        __syncthreads();                           // the ring / the previous chunk is no longer being read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lr = wr * 64 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = wc * 16 + j * 4 + g;
                *(f32x4*)(tile + lr * 256 + ((cc ^ (lr & 31)) << 2)) = acc[c * 4 + i][j];
            }
        }
        __syncthreads();
        // Interior tiles (all of them when M, N are multiples of 256) take a branch-free path: with per-row guards
        // hipcc's waitcnt pass loses count at every join and puts `s_waitcnt vmcnt(0)` in front of each store, i.e.
        // every store waits for the previous one to complete.
        if (interior) {
#pragma unroll
            for (int pb = 0; pb < 16; pb += 8) {
                float4 r4[8]; uint2 u2[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int lr = (pb + q) * 8 + w;
                    const int m = tm * 256 + (lr >> 6) * 128 + c * 64 + (lr & 63);
                    if (EPI == KZV_EPI_RESID) r4[q] = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
                    if (EPI == KZV_EPI_DGELU) u2[q] = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int lr = (pb + q) * 8 + w;
                    const int m = tm * 256 + (lr >> 6) * 128 + c * 64 + (lr & 63);
                    const f32x4 a4 = *(const f32x4*)(tile + lr * 256 + ((c4 ^ (lr & 31)) << 2));
                    float v[4] = {a4[0] + b4[0], a4[1] + b4[1], a4[2] + b4[2], a4[3] + b4[3]};
                    nt_emit<EPI>(p, m, n0, v, r4[q], u2[q]);
                }
            }
        } else if (n0 < p.N) {
#pragma unroll 1
            for (int q = 0; q < 16; ++q) {
                const int lr = q * 8 + w;
                const int m = tm * 256 + (lr >> 6) * 128 + c * 64 + (lr & 63);
                if (m >= p.M) continue;
                float4 r4 = make_float4(0, 0, 0, 0); uint2 u2 = make_uint2(0, 0);
                if (EPI == KZV_EPI_RESID) r4 = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
                if (EPI == KZV_EPI_DGELU) u2 = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
                const f32x4 a4 = *(const f32x4*)(tile + lr * 256 + ((c4 ^ (lr & 31)) << 2));
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = nv[r] ? a4[r] + b4[r] : 0.f;
                nt_emit<EPI>(p, m, n0, v, r4, u2);
            }
        }
    }
}

int nt256_min_tiles() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("KZV_NT256_MIN_TILES"); v = e ? atoi(e) : 384; }
    return v;
}

}  // namespace

int kzv_nt256_launch(const NtParams& p, int epilogue, hipStream_t s) {
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    // below ~1.5 rounds of the 256 CUs the 128x128 kernel (4x the tiles, 2 workgroups per CU) fills the chip better
    if (p.K < 128 || tiles < nt256_min_tiles()) return 0;
    if ((uint64_t)256 * (uint64_t)p.lda * 2 > 0xffffffffull || (uint64_t)p.n_valid * (uint64_t)p.ldb * 2 > 0xffffffffull) return 0;   // 32-bit DMA offsets
#define KZV_NT256_CASE(E)                                                                                           \
    case E: {                                                                                                       \
        static bool attr_done = false;                                                                              \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done = true; } \
        hipLaunchKernelGGL((gemm_nt256_kernel<E>), dim3(tiles), dim3(512), LDS_BYTES, s, p);                        \
    } break;
    switch (epilogue) {
        KZV_NT256_CASE(KZV_EPI_BF16) KZV_NT256_CASE(KZV_EPI_F32) KZV_NT256_CASE(KZV_EPI_GELU)
        KZV_NT256_CASE(KZV_EPI_RESID) KZV_NT256_CASE(KZV_EPI_DGELU) KZV_NT256_CASE(KZV_EPI_GELU_F32)
        default: return 0;
    }
#undef KZV_NT256_CASE
    return 1;
}
