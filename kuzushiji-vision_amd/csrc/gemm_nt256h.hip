// gemm_nt, persistent 256x128 kernel of FOUR waves for gfx950 (MI355X): two workgroups per CU, out of phase.
//
// VERDICT r03 item 3 (the fc1 + GELU drain): a 256x256 tile's accumulators fill the register file of its eight waves, so a
// workgroup's drain (13 us of erf / exp / stores for the GELU epilogue against an 18-us K loop) cannot overlap its own next
// K loop, and one workgroup is all a CU holds.  Here a workgroup is HALF of that -- 4 waves, a 256x128 tile, one wave per SIMD
// with the same 128x64 wave tile, 80 KiB of LDS -- so two live on a CU, and while one drains the other owns the matrix pipe.
// The K loop is gemm_nt256f.hip's free-running schedule (every wave hides its own fragment reads and LDS-DMA issues under
// its own MFMAs: the only kind of schedule that works with one wave per SIMD); read that file's header first.  Differences:
//
// * LDS ring of THREE half-tile groups instead of two K-tiles (80 KiB has no room for two): group g(2u) = h0(u) =
//   {A-h0 (128 rows), B-h0 (64 columns)} of K-tile u, g(2u + 1) = h1(u); group n lives in slot n % 3 (A slots 3 x 16 KiB, then
//   B slots 3 x 8 KiB).  h0(u) is dead behind the barrier in p1(u) -> its slot takes h1(u + 1); h1(u) is dead behind the barrier
//   in p3(u) -> its slot takes h0(u + 2): every DMA flies four phases (64 MFMAs of its wave) and each wait is vmcnt(6) (one
//   younger group of 4 A + 2 B pieces per wave).  Slots repeat every 3 K-tiles and the two B fragment sets swap every K-tile:
//   the body is unrolled by 6, K % 384 == 0 (768, 2304, 3072: every encoder GEMM of ViT-B).
// * The second workgroup of a CU starts `stagger` microseconds late (blockIdx >> 3 >= grid / 16: the dispatcher fills the 32 CUs
//   of an XCD once before it doubles up), which is what de-phases the pair; lockstep pairs drain together and gain nothing.
// * Drain through a 2-KiB patch per wave: half a row block ([16 rows][32 columns] fp32) at a time, 128-byte row segments.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "gemm_nt.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int A_HT = 128 * 128;            // A half-tile: 128 rows x 64 bf16
constexpr int B_HT = 64 * 128;             // B half-tile: 64 columns x 64 bf16
constexpr int B_BASE = 3 * A_HT;           // B slots behind the three A slots
constexpr int RING_BYTES = 3 * (A_HT + B_HT);      // 72 KiB
constexpr int LDS_BYTES = RING_BYTES + 4 * 2048;   // + one 2-KiB drain patch per wave = 80 KiB: two workgroups per CU

__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 38) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
    else if constexpr (N == 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}
constexpr int cmin(int a, int b) { return a < b ? a : b; }
#define KZV_SB() __builtin_amdgcn_sched_barrier(0)

// VMEM operations one wave issues while draining an interior tile (32 four-column groups per lane)
template <int EPI> constexpr int drain_ops() { return (EPI == KZV_EPI_BF16 || EPI == KZV_EPI_F32) ? 32 : 64; }

struct TileSrc {            // where the next half-tile group of one half index comes from: wave-uniform (SGPRs) throughout
    const char* a; const char* b;      // tile row panel of A, tile column panel of B, both at the stream's current K-tile
    unsigned limA, limB;               // largest byte offset a lane may read in the panel (rows beyond M / n_valid are clamped onto it)
    int kt, seq;
};

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt256h_kernel(const NtParams p, const int tiles, const int tilesN, const int strip, const int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wr = w >> 1, wc = w & 1;
    const int G = gridDim.x;
    // blocks land on XCD (blockIdx % 8): give each XCD a contiguous run of every step's tiles (shared A row panels)
    const int vblk = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int nk = p.K / 64;                       // K-tiles: a multiple of 6 (checked by the launcher)
    const unsigned lda2 = (unsigned)p.lda * 2u, ldb2 = (unsigned)p.ldb * 2u;

    // LDS-DMA sources.  Wave w fills the 1-KiB pieces w + 4t of every half-tile (A: t = 0..3, B: t = 0..1): piece w + 4t, lane l
    // -> local row r = 32t + q, q = w*8 + (l >> 3), 16-byte chunk (l & 7) ^ (r & 7) of the 128-byte K-tile row.  r maps to tile
    // row (t >> 1)*128 + h*64 + (t & 1)*32 + q of A and to tile column t*64 + h*32 + q of B (local rows = [wave row / column][half's
    // rows]), so a lane's byte offset is [per-lane, fixed for the kernel] qA / qB + [wave-uniform] a row count * lda2 / ldb2.
    unsigned qA, qB;
    {
        const int q = w * 8 + (lane >> 3);
        const unsigned cb = (unsigned)(((lane & 7) ^ (q & 7)) * 16);
        qA = (unsigned)q * lda2 + cb;
        qB = (unsigned)q * ldb2 + cb;
    }
    auto set_tile = [&](TileSrc& s, int seq) {
        s.seq = seq; s.kt = 0;
        const int id = seq * G + vblk;
        const int idc = id < tiles ? id : 0;       // a stream past its last tile re-reads tile 0 into slots nobody reads
        int tm, tn;
        nt_tile_coords(idc, tiles / tilesN, tilesN, strip, tm, tn);
        tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
        s.a = (const char*)p.A + (int64_t)tm * 256 * p.lda * 2;
        s.b = (const char*)p.B + (int64_t)tn * 128 * p.ldb * 2;
        s.limA = (unsigned)min(p.M - 1 - tm * 256, 255) * lda2 + 112u;
        s.limB = (unsigned)min(p.n_valid - 1 - tn * 128, 127) * ldb2 + 112u;       // the launcher keeps every tile's first column < n_valid
    };
    auto advance = [&](TileSrc& s) {
        s.a += 128; s.b += 128;
        if (++s.kt == nk) set_tile(s, s.seq + 1);
    };
    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)smem) + (unsigned)w * 1024u);
    auto stageA = [&](const TileSrc& s, int slot, int h, int t) {
        const unsigned v = min(qA + (unsigned)((t >> 1) * 128 + h * 64 + (t & 1) * 32) * lda2, s.limA);
        glds16_s(v, s.a, ldsw + (unsigned)(slot * A_HT + t * 4096));
    };
    auto stageB = [&](const TileSrc& s, int slot, int h, int t) {
        const unsigned v = min(qB + (unsigned)(t * 64 + h * 32) * ldb2, s.limB);
        glds16_s(v, s.b, ldsw + (unsigned)(B_BASE + slot * B_HT + t * 4096));
    };

    f32x4 acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // fragment reads: lane supplies row l15 of a 16-row fragment, 16-byte slots (g ^ (l15 & 7)) and ((4 + g) ^ (l15 & 7))
    const KZV_LDS char *pA0, *pA1, *pB0, *pB1;        // LDS address space: 32-bit, ds_read with immediate offsets
    {
        const int sw = l15 & 7;
        const int slot0 = (g ^ sw) << 4, slot1 = ((4 + g) ^ sw) << 4;
        const int a_off = (wr * 64 + l15) * 128, b_off = B_BASE + (wc * 32 + l15) * 128;
        const KZV_LDS char* sm = (const KZV_LDS char*)smem;
        pA0 = sm + a_off + slot0; pA1 = sm + a_off + slot1; pB0 = sm + b_off + slot0; pB1 = sm + b_off + slot1;
        asm volatile("" : "+v"(pA0), "+v"(pA1), "+v"(pB0), "+v"(pB1));      // four base registers; everything else is an immediate
    }
    struct Frag { bf16x8 k[2]; };          // the two 16-byte K-chunks of one 16-row fragment row: two MFMAs
    Frag fa[4], fbX[2], fbY[2];
    auto rdA = [&](int slot, int i) {
        const int o = slot * A_HT + i * 2048;
        fa[i].k[0] = *(const KZV_LDS bf16x8*)(pA0 + o); fa[i].k[1] = *(const KZV_LDS bf16x8*)(pA1 + o);
    };
    auto rdB = [&](int slot, int j, Frag (&fb)[2]) {
        const int o = slot * B_HT + j * 2048;
        fb[j].k[0] = *(const KZV_LDS bf16x8*)(pB0 + o); fb[j].k[1] = *(const KZV_LDS bf16x8*)(pB1 + o);
    };
    // two MFMAs: accumulator row block i of half mh against both column blocks of half nh, K-chunk kh
    auto mm2 = [&](int mh, int nh, int i, int kh, const Frag (&fb)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j].k[kh], fa[i].k[kh], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
    };
    auto mm4 = [&](int mh, int nh, int i, const Frag (&fb)[2]) { mm2(mh, nh, i, 0, fb); mm2(mh, nh, i, 1, fb); };

    TileSrc s0, s1;                    // s0 feeds the h0 groups (stands at K-tile u + 2 when K-tile u starts), s1 the h1 groups (u + 1)
    constexpr int D = drain_ops<EPI>();
    constexpr int W6 = cmin(63, 6 + D);

    // One K-tile of the stream (header).  U = u % 3 fixes the slots; Bf: the B set holding B0(u) (p1, p4), Bs: the set B1(u) is
    // read into (p2, p3) and, in p4, B0(u+1).  wide: the waits follow a credited drain.  last: last K-tile of an output tile.
    auto ktile = [&](auto uc, Frag (&Bf)[2], Frag (&Bs)[2], bool wide, bool last) {
        constexpr int U = decltype(uc)::value;
        constexpr int S0 = (2 * U) % 3, S1 = (2 * U + 1) % 3, S0N = (2 * U + 2) % 3;
        // ---- p1 (A0, B0) ----
        mm4(0, 0, 0, Bf); mm4(0, 0, 1, Bf);
        KZV_SB();
        if (wide) vmcnt<W6>(); else vmcnt<6>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        KZV_SB();
        rdB(S1, 0, Bs); rdB(S1, 1, Bs);
        KZV_SB();
        mm2(0, 0, 2, 0, Bf); KZV_SB(); stageA(s1, S0, 1, 0); KZV_SB();
        mm2(0, 0, 2, 1, Bf); KZV_SB(); stageA(s1, S0, 1, 1); KZV_SB();
        mm2(0, 0, 3, 0, Bf); KZV_SB(); stageA(s1, S0, 1, 2); KZV_SB();
        mm2(0, 0, 3, 1, Bf); KZV_SB(); stageA(s1, S0, 1, 3); KZV_SB();
        // ---- p2 (A0, B1) ----
        mm4(0, 1, 0, Bs); KZV_SB(); stageB(s1, S0, 1, 0); rdA(S1, 0); KZV_SB();
        mm4(0, 1, 1, Bs); KZV_SB(); stageB(s1, S0, 1, 1); rdA(S1, 1); KZV_SB();
        mm4(0, 1, 2, Bs); KZV_SB(); rdA(S1, 2); KZV_SB();
        mm4(0, 1, 3, Bs); KZV_SB(); rdA(S1, 3); KZV_SB();
        advance(s1);
        // ---- p3 (A1, B1) ----
        mm4(1, 1, 0, Bs); mm4(1, 1, 1, Bs);
        KZV_SB();
        if (wide) vmcnt<W6>(); else vmcnt<6>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        KZV_SB();
        mm2(1, 1, 2, 0, Bs); KZV_SB(); stageA(s0, S1, 0, 0); KZV_SB();
        mm2(1, 1, 2, 1, Bs); KZV_SB(); stageA(s0, S1, 0, 1); KZV_SB();
        mm2(1, 1, 3, 0, Bs); KZV_SB(); stageA(s0, S1, 0, 2); KZV_SB();
        mm2(1, 1, 3, 1, Bs); KZV_SB(); stageA(s0, S1, 0, 3); KZV_SB();
        // ---- p4 (A1, B0) ----
        mm4(1, 0, 0, Bf); KZV_SB(); stageB(s0, S1, 0, 0); if (!last) { rdA(S0N, 0); rdB(S0N, 0, Bs); } KZV_SB();
        mm4(1, 0, 1, Bf); KZV_SB(); stageB(s0, S1, 0, 1); if (!last) { rdA(S0N, 1); rdB(S0N, 1, Bs); } KZV_SB();
        mm4(1, 0, 2, Bf); KZV_SB(); if (!last) rdA(S0N, 2); KZV_SB();
        mm4(1, 0, 3, Bf); KZV_SB(); if (!last) rdA(S0N, 3); KZV_SB();
        advance(s0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;

    // ---- drain: this wave's 128x64 accumulators -> global, through its private 2-KiB LDS patch, half a row block at a time ----
    auto drain = [&](int tm, int tn, auto interiorc) {
        constexpr bool interior = decltype(interiorc)::value;
        int ln = lane;
        asm volatile("" : "+v"(ln));       // keep the drain's address terms out of the K loop's live set
        // patch = half an accumulator row block: [16 rows][32 cols] fp32 (128-B rows), 16-B chunks XOR (row & 7)
        float* patch = (float*)(smem + RING_BYTES + w * 2048);
        const int prow = ln >> 3, pchunk = ln & 7;            // read-back: 8 rows x 128 B per wave-instruction
        const int l15 = ln & 15, g = ln >> 4;
        const int nb = tn * 128 + wc * 64 + pchunk * 4;       // + ch*32: accumulator column blocks 2ch, 2ch + 1
        float b4[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bool nv[2][4];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int n0 = nb + ch * 32;
#pragma unroll
            for (int r = 0; r < 4; ++r) nv[ch][r] = interior || n0 + r < p.n_valid;
            if (EPI != KZV_EPI_DGELU && p.bias) {             // before any store (a later load could only be waited for with them)
                if constexpr (interior) { const float4 t = *(const float4*)(p.bias + n0); b4[ch][0] = t.x; b4[ch][1] = t.y; b4[ch][2] = t.z; b4[ch][3] = t.w; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (nv[ch][r]) b4[ch][r] = p.bias[n0 + r];
                }
            }
        }
        auto block_loads = [&](int b, float4 (&r4)[2], uint2 (&u2)[2]) {
            const int ch = b >> 3, i = b & 7;
            const int m0 = tm * 256 + wr * 128 + i * 16, n0 = nb + ch * 32;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + q * 8 + prow;
                if (EPI == KZV_EPI_RESID) r4[q] = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
                if (EPI == KZV_EPI_DGELU) u2[q] = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
            }
        };
        // interior tiles: branch-free, the residual / derivative loads running LOOK half-blocks ahead of their use; edge tiles:
        // guarded, row by row
        constexpr int LOOK = 6;
        float4 r4[16][2]; uint2 u2[16][2];
        if constexpr (interior) {
#pragma unroll
            for (int b = 0; b < LOOK; ++b) block_loads(b, r4[b], u2[b]);
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) {                        // half-block b = (column half ch, row block i): tile rows wr*128 + i*16 .. +15
            const int ch = b >> 3, i = b & 7;
            const int m0 = tm * 256 + wr * 128 + i * 16, n0 = nb + ch * 32;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int chunk = jj * 4 + g;
                *(f32x4*)(patch + l15 * 32 + ((chunk ^ (l15 & 7)) << 2)) = acc[i][ch * 2 + jj];
            }
            if constexpr (interior) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = q * 8 + prow;
                    const f32x4 a4 = *(const f32x4*)(patch + row * 32 + ((pchunk ^ prow) << 2));
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = a4[r] + b4[ch][r];
                    nt_emit<EPI>(p, m0 + row, n0, v, r4[b][q], u2[b][q]);
                }
                if (b + LOOK < 16) block_loads(b + LOOK, r4[b + LOOK], u2[b + LOOK]);
            } else {
#pragma unroll 1
                for (int q = 0; q < 2; ++q) {
                    const int row = q * 8 + prow;
                    const int m = m0 + row;
                    const f32x4 a4 = *(const f32x4*)(patch + row * 32 + ((pchunk ^ prow) << 2));
                    if (m < p.M && n0 < p.N) {
                        float4 e4 = make_float4(0, 0, 0, 0); uint2 eu = make_uint2(0, 0);
                        if (EPI == KZV_EPI_RESID) e4 = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
                        if (EPI == KZV_EPI_DGELU) eu = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = nv[ch][r] ? a4[r] + b4[ch][r] : 0.f;
                        nt_emit<EPI>(p, m, n0, v, e4, eu);
                    }
                }
            }
        }
    };

    // ---- the stream ----
    set_tile(s0, 0); set_tile(s1, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) stageA(s0, 0, 0, t);            // g(0) = h0(0) -> slot 0
    stageB(s0, 0, 0, 0); stageB(s0, 0, 0, 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) stageA(s1, 1, 1, t);            // g(1) = h1(0) -> slot 1
    stageB(s1, 1, 1, 0); stageB(s1, 1, 1, 1);
    advance(s0); advance(s1);
#pragma unroll
    for (int t = 0; t < 4; ++t) stageA(s0, 2, 0, t);            // g(2) = h0(1) -> slot 2
    stageB(s0, 2, 0, 0); stageB(s0, 2, 0, 1);
    advance(s0);
    // the second workgroup of a CU starts late: the pair runs out of phase from then on (header)
    if (stagger > 0 && (int)(blockIdx.x >> 3) >= (G >> 4)) {
#pragma unroll 1
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    vmcnt<12>();                                    // g(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    auto first_frags = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) rdA(0, i);
        rdB(0, 0, fbX); rdB(0, 1, fbX);
    };
    first_frags();
    bool credit = false;                            // previous drain was of an interior tile
#ifdef KZV_STAMPS
    // per-block stamps (dev, tools/dev/r4_half_stamps.py): [blockIdx][16] u64 in p.aux: start, then (K loop end, drain end) per tile; [15] = HW_ID | XCC_ID << 32
    unsigned long long* stp = (EPI == KZV_EPI_BF16 && tid == 0) ? (unsigned long long*)p.aux + blockIdx.x * 16 : nullptr;
    int stk = 0;
    if (stp) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stp[15] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
#define KZV_STAMP() do { if (stp && stk < 15) stp[stk++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KZV_STAMP() do {} while (0)
#endif
    KZV_STAMP();
    for (int seq = 0; ; ++seq) {
        const int id = seq * G + vblk;
        if (id >= tiles) break;
        int tm, tn;
        nt_tile_coords(id, tiles / tilesN, tilesN, strip, tm, tn);
        tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
        // nk % 6 == 0 (launcher): every tile starts on slot 0 with B0 in fbX, so the six K-tile bodies alternate statically
        for (int kt = 0; kt < nk; kt += 6) {
            ktile(I0{}, fbX, fbY, credit && kt == 0, false);
            ktile(I1{}, fbY, fbX, false, false);
            ktile(I2{}, fbX, fbY, false, false);
            ktile(I0{}, fbY, fbX, false, false);
            ktile(I1{}, fbX, fbY, false, false);
            ktile(I2{}, fbY, fbX, false, kt + 6 >= nk);
        }
        KZV_STAMP();
        credit = tm * 256 + 256 <= p.M && tn * 128 + 128 <= p.n_valid;     // interior tile (n_valid <= N)
        if (credit) drain(tm, tn, std::true_type{}); else drain(tm, tn, std::false_type{});
        zero_acc();
        first_frags();                              // A0 / B0 of the next tile's first K-tile (retired before the drain)
        KZV_STAMP();
    }
    vmcnt<0>();                                     // the refills issued past the end of the stream (into dead slots) land before the LDS is released
}

int nt256h_min_tiles() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("KZV_NT256H_MIN_TILES"); v = e ? atoi(e) : 768; }
    return v;
}
int device_cus_h() {
    static int v = -1;
    if (v < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        v = n;
    }
    return v;
}
int g_stagger = -1;

}  // namespace

extern "C" int kzv_set_nt_half_stagger(int us) { g_stagger = us; return KZV_OK; }

int kzv_nt256h_launch(const NtParams& p, int epilogue, hipStream_t s) {
    const int tilesN = (p.N + 127) / 128;
    const int tiles = ((p.M + 255) / 256) * tilesN;
    if (p.K < 384 || p.K % 384 || tiles < nt256h_min_tiles()) return 0;   // six K-tile bodies per round of the slot / fragment-set pattern
    if ((uint64_t)256 * (uint64_t)p.lda * 2 > 0xffffffffull || (uint64_t)p.n_valid * (uint64_t)p.ldb * 2 > 0xffffffffull) return 0;   // 32-bit DMA offsets
    if (p.n_valid <= (tilesN - 1) * 128 || p.lda * 2 < 128 || p.ldb * 2 < 128) return 0;          // every tile starts on a valid column (the DMA clamp needs one)
    if (g_stagger < 0) { const char* e = getenv("KZV_NTH_STAGGER"); g_stagger = e ? atoi(e) : 8; }
    const int grid = tiles < 2 * device_cus_h() ? tiles : 2 * device_cus_h();
    const int strip = (kzv_nt_strip() & 0xff) * 2;
#define KZV_NT256H_CASE(E)                                                                                          \
    case E: {                                                                                                       \
        static bool attr_done = false;                                                                              \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt256h_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done = true; } \
        hipLaunchKernelGGL((gemm_nt256h_kernel<E>), dim3(grid), dim3(256), LDS_BYTES, s, p, tiles, tilesN, strip, g_stagger);         \
    } break;
    switch (epilogue) {
        KZV_NT256H_CASE(KZV_EPI_BF16) KZV_NT256H_CASE(KZV_EPI_F32) KZV_NT256H_CASE(KZV_EPI_GELU)
        KZV_NT256H_CASE(KZV_EPI_RESID) KZV_NT256H_CASE(KZV_EPI_DGELU) KZV_NT256H_CASE(KZV_EPI_GELU_F32)
        default: return 0;
    }
#undef KZV_NT256H_CASE
    return 1;
}
