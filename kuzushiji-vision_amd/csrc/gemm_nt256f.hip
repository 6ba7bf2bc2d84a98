// gemm_nt, persistent 256x256 kernel with a FREE-RUNNING schedule for gfx950 (MI355X).
//
// Same tile, LDS ring (2 K-tiles x 4 half-tiles of 16 KiB), DMA pieces, swizzle, tile stream and per-wave drain as
// gemm_nt256p.hip (read its header and gemm_nt256.hip's first).  What changes is the K loop: gemm_nt256p is the eight-phase
// ping-pong (waves 4..7 one barrier behind, every phase = [reads + DMA issue] barrier [16 MFMA] barrier: while one wave of a
// SIMD multiplies, its partner loads), whose barrier interval measured 362 cycles where the 16 MFMAs are 256 (the
// load half is a 247-cycle latency chain of its own).  Here all eight waves run the SAME stream with TWO barriers per K-tile,
// and every wave hides its own loads under its own MFMAs:
//
//     p1 (A0,B0): 8 MFMA | vmcnt, BARRIER | read B1 (4 ds_read) ; 8 MFMA with the 4 LDS-DMA of h0(u+2) between them
//     p2 (A0,B1): 4 x [4 MFMA ; read A1[i] in place of A0[i] (2 ds_read)]
//     p3 (A1,B1): 8 MFMA | vmcnt, BARRIER | 8 MFMA with the 4 LDS-DMA of h1(u+2) between them
//     p4 (A1,B0): 4 x [4 MFMA ; read A0(u+1)[i] in place of A1[i] ; i < 2: read B0(u+1)[i] into the B set p2 / p3 used]
//
// (u = stream K-tile, h0 = {A-h0, B-h0}, h1 = {B-h1, A-h1} half-tiles.)  The two B fragment sets swap roles every K-tile (the
// body is unrolled by two, as the ring parity needs anyway), so the fragment registers are the 64 of the ping-pong kernel.
//
// LDS ordering.  RAW: a half-tile is read only after a barrier that follows every wave's counted vmcnt for its own pieces:
// h1(u) is retired before the barrier in p1(u) and read right after it (B1) and in p2 (A1); h0(u+1) is retired before the
// barrier in p3(u) and read in p4(u).  Each was issued six phases earlier (h1(u): second half of p3(u-2); h0(u+1): second half
// of p1(u-1)), with 8 younger DMAs behind it: vmcnt(8), never 0 in the steady loop.  WAR: a slot is re-filled only after a
// barrier that follows every wave's last read of it: the reads of p4(u-1) (h0(u)) are complete before the barrier in p1(u)
// (lgkmcnt(0) there: they were issued >= 8 MFMAs earlier), the refill h0(u+2) follows it; the reads of p1 / p2(u) (h1(u))
// are complete before the barrier in p3(u), the refill h1(u+2) follows it.
//
// Tile boundaries: the last K-tile of an output tile skips p4's reads (the fragments would be live across the drain); the
// first K-tile's A0 / B0 are read after the drain (their half-tiles were retired at the barrier in p3 of the previous K-tile).
// Eight DMAs (h0(u+2), h1(u+2)) are in flight across a drain; the first three waits after a credited drain are widened by the
// drain's own operation count (vmcnt bookkeeping as in gemm_nt256p.hip).  A stream that has run past its last tile keeps
// issuing (it re-reads tile 0 into slots nobody reads again), so the loop has no tail case and every count stays exact; the
// kernel ends on vmcnt(0).
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "gemm_nt.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HT_BYTES = 128 * 128;        // half-tile: 128 rows x 64 bf16
constexpr int RING_BYTES = 8 * HT_BYTES;   // 128 KiB: A [buf][h] in the first 64 KiB, B [buf][h] in the second
constexpr int LDS_BYTES = RING_BYTES + 8 * 4096;   // + one 4-KiB drain patch per wave = 160 KiB

__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else if constexpr (N == 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}
constexpr int cmin(int a, int b) { return a < b ? a : b; }
#define KZV_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef KZV_F_NOBAR
#define KZV_F_BARRIER() do {} while (0)
#else
#define KZV_F_BARRIER() __builtin_amdgcn_s_barrier()
#endif

// VMEM operations one wave issues while draining an interior tile (32 four-column groups per lane)
template <int EPI> constexpr int drain_ops() { return (EPI == KZV_EPI_BF16 || EPI == KZV_EPI_F32) ? 32 : 64; }

struct TileSrc {            // where the next half-tiles of one half index come from: wave-uniform (SGPRs) throughout
    const char* a; const char* b;      // tile row panel of A, tile column panel of B, both at the stream's current K-tile
    unsigned limA, limB;               // largest byte offset a lane may read in the panel (rows beyond M / n_valid are clamped onto it)
    int kt, seq; bool valid;
};

template <int EPI>
__global__ __launch_bounds__(512) void gemm_nt256f_kernel(const NtParams p, const int tiles, const int tilesN, const int strip_in) {
    constexpr bool F8 = false;
    const int strip = strip_in & 0xff;             // bit 8: the double-buffered bf16 drain
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wr = w >> 2, wc = w & 3;
    const int G = gridDim.x;
    // blocks land on XCD (blockIdx % 8): give each XCD a contiguous run of every step's tiles (shared A row panels)
    const int vblk = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int nk = p.K / 64;                       // K-tiles: even, >= 2 (checked by the launcher)
    const unsigned lda2 = (unsigned)p.lda * 2u, ldb2 = (unsigned)p.ldb * 2u;

    // LDS-DMA sources.  Wave w fills 1-KiB pieces w and w + 8 of every half-tile: piece j, lane l -> local row r = j*64 + q,
    // q = w*8 + (l >> 3), 16-byte chunk (l & 7) ^ (r & 7) of the 128-byte K-tile row.  r maps to tile row j*128 + h*64 + q of A
    // and to tile column (2j + (q >> 5))*64 + h*32 + (q & 31) of B, so a lane's byte offset is
    //     [per-lane, fixed for the kernel] qA / qB  +  [wave-uniform] (j*128 + h*64) * lda2 / (j*128 + h*32) * ldb2
    // and only the two per-lane terms live in VGPRs.  Rows beyond M / columns beyond n_valid (never stored / stored as 0) are
    // clamped by a min against the tile's largest valid offset (any valid address will do for them).
    unsigned qA, qB;
    {
        const int q = w * 8 + (lane >> 3);
        const unsigned cb = (unsigned)(((lane & 7) ^ (q & 7)) * 16);
        qA = (unsigned)q * lda2 + cb;
        qB = (unsigned)((q >> 5) * 64 + (q & 31)) * ldb2 + cb;
    }
    auto set_tile = [&](TileSrc& s, int seq) {
        s.seq = seq; s.kt = 0;
        const int id = seq * G + vblk;
        s.valid = id < tiles;
        const int idc = s.valid ? id : 0;
        int tm, tn;
        nt_tile_coords(idc, tiles / tilesN, tilesN, strip, tm, tn);
        tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
        s.a = (const char*)p.A + (int64_t)tm * 256 * p.lda * 2;
        s.b = (const char*)p.B + (int64_t)tn * 256 * p.ldb * 2;
        s.limA = (unsigned)min(p.M - 1 - tm * 256, 255) * lda2 + 112u;
        s.limB = (unsigned)min(p.n_valid - 1 - tn * 256, 255) * ldb2 + 112u;       // the launcher keeps every tile's first column < n_valid
    };
    auto advance = [&](TileSrc& s) {
        s.a += 128; s.b += 128;
        if (++s.kt == nk) set_tile(s, s.seq + 1);
    };
    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)smem) + (unsigned)w * 1024u);
    // LDS ring: A half-tiles in the first 64 KiB ([buf][h] x 16 KiB), B half-tiles in the second, so that each operand's
    // fragment reads are ONE base register + a 16-bit immediate.  One 1-KiB piece (j = 0 / 1) of this wave's share of a half-tile:
    // -DKZV_F_NODMA / _NOREAD / _NOBAR / _NOMFMA: ablation builds (tools/dev/r4_ablate.sh; garbage results, timing only)
    auto stageA1 = [&](const TileSrc& s, int buf, int h, int j) {
#ifdef KZV_F_NODMA
        return;
#endif
        const unsigned v = min(qA + (unsigned)(j * 128 + h * 64) * lda2, s.limA);
        glds16_s(v, s.a, ldsw + (unsigned)((buf * 2 + h) * HT_BYTES + j * 8192));
    };
    auto stageB1 = [&](const TileSrc& s, int buf, int h, int j) {
#ifdef KZV_F_NODMA
        return;
#endif
        const unsigned v = min(qB + (unsigned)(j * 128 + h * 32) * ldb2, s.limB);
        glds16_s(v, s.b, ldsw + (unsigned)(65536 + (buf * 2 + h) * HT_BYTES + j * 8192));
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
        const int a_off = (wr * 64 + l15) * 128, b_off = 65536 + (wc * 32 + l15) * 128;
        const KZV_LDS char* sm = (const KZV_LDS char*)smem;
        pA0 = sm + a_off + slot0; pA1 = sm + a_off + slot1; pB0 = sm + b_off + slot0; pB1 = sm + b_off + slot1;
        asm volatile("" : "+v"(pA0), "+v"(pA1), "+v"(pB0), "+v"(pB1));      // four base registers; everything else is an immediate
    }
    struct Frag { bf16x8 k[2]; };          // the two 16-byte K-chunks of one 16-row fragment row: two MFMAs
    Frag fa[4], fbX[2], fbY[2];
    bool rd_on = true;
    auto rdA = [&](int buf, int mh, int i) {
        if (!rd_on) return;
        const int o = (buf * 2 + mh) * HT_BYTES + i * 2048;
        fa[i].k[0] = *(const KZV_LDS bf16x8*)(pA0 + o); fa[i].k[1] = *(const KZV_LDS bf16x8*)(pA1 + o);
    };
    auto rdB = [&](int buf, int nh, int j, Frag (&fb)[2]) {
        if (!rd_on) return;
        const int o = (buf * 2 + nh) * HT_BYTES + j * 2048;
        fb[j].k[0] = *(const KZV_LDS bf16x8*)(pB0 + o); fb[j].k[1] = *(const KZV_LDS bf16x8*)(pB1 + o);
    };
    // two MFMAs: accumulator row block i of half mh against both column blocks of half nh, K-chunk kh
    auto mm2 = [&](int mh, int nh, int i, int kh, const Frag (&fb)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#ifdef KZV_F_NOMFMA
            asm volatile("" : : "v"(fb[j].k[kh]), "v"(fa[i].k[kh]));
#else
            acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j].k[kh], fa[i].k[kh], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
#endif
    };
    auto mm4 = [&](int mh, int nh, int i, const Frag (&fb)[2]) { mm2(mh, nh, i, 0, fb); mm2(mh, nh, i, 1, fb); };

    const bool late = w >= 4;          // wave-uniform: which half of a barrier interval this wave issues its DMAs in
    TileSrc s0, s1;                    // s0 feeds A-h0 / B-h0, s1 feeds B-h1 / A-h1; both stand at stream K-tile u + 2 when K-tile u starts
    constexpr int D = drain_ops<EPI>();
    constexpr int W8 = cmin(63, 8 + D);

    // One K-tile of the stream (header).  Bf: the B set holding B0(u) (p1, p4), Bs: the set B1(u) is read into (p2, p3) and,
    // in p4, B0(u+1).  wide1 / wide2: the wait of p1 / p3 follows a credited drain.  last: last K-tile of an output tile.
    auto ktile = [&](auto bufc, Frag (&Bf)[2], Frag (&Bs)[2], bool wide1, bool wide2, bool last) {
        constexpr int BUF = decltype(bufc)::value;
        // ---- p1 ----
        mm4(0, 0, 0, Bf); mm4(0, 0, 1, Bf);
        KZV_SB();
        if (wide1) vmcnt<W8>(); else vmcnt<8>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        KZV_F_BARRIER();
        KZV_SB();
        rdB(BUF, 1, 0, Bs); rdB(BUF, 1, 1, Bs);
        KZV_SB();
        // waves 0..3 issue the half-tile's four DMAs here, between the first MFMAs behind the barrier; waves 4..7 (the SIMD
        // partners) issue theirs 16 MFMAs later, inside p2 / p4: the two waves of a SIMD run this stream in lockstep, and a DMA
        // issue holds its wave for ~60..100 cycles -- placed alike, both partners stall together and the matrix pipe idles
        mm2(0, 0, 2, 0, Bf); KZV_SB(); if (!late) stageA1(s0, BUF, 0, 0); KZV_SB();
        mm2(0, 0, 2, 1, Bf); KZV_SB(); if (!late) stageA1(s0, BUF, 0, 1); KZV_SB();
        mm2(0, 0, 3, 0, Bf); KZV_SB(); if (!late) stageB1(s0, BUF, 0, 0); KZV_SB();
        mm2(0, 0, 3, 1, Bf); KZV_SB(); if (!late) stageB1(s0, BUF, 0, 1); KZV_SB();
        // ---- p2 ----
        mm4(0, 1, 0, Bs); KZV_SB(); rdA(BUF, 1, 0); KZV_SB();
        mm4(0, 1, 1, Bs); KZV_SB(); rdA(BUF, 1, 1); KZV_SB();
        mm2(0, 1, 2, 0, Bs); KZV_SB(); if (late) stageA1(s0, BUF, 0, 0); KZV_SB();
        mm2(0, 1, 2, 1, Bs); KZV_SB(); if (late) stageA1(s0, BUF, 0, 1); rdA(BUF, 1, 2); KZV_SB();
        mm2(0, 1, 3, 0, Bs); KZV_SB(); if (late) stageB1(s0, BUF, 0, 0); KZV_SB();
        mm2(0, 1, 3, 1, Bs); KZV_SB(); if (late) stageB1(s0, BUF, 0, 1); rdA(BUF, 1, 3); KZV_SB();
        advance(s0);
        // ---- p3 ----
        mm4(1, 1, 0, Bs); mm4(1, 1, 1, Bs);
        KZV_SB();
        if (wide2) vmcnt<W8>(); else vmcnt<8>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        KZV_F_BARRIER();
        KZV_SB();
        mm2(1, 1, 2, 0, Bs); KZV_SB(); if (!late) stageB1(s1, BUF, 1, 0); KZV_SB();
        mm2(1, 1, 2, 1, Bs); KZV_SB(); if (!late) stageB1(s1, BUF, 1, 1); KZV_SB();
        mm2(1, 1, 3, 0, Bs); KZV_SB(); if (!late) stageA1(s1, BUF, 1, 0); KZV_SB();
        mm2(1, 1, 3, 1, Bs); KZV_SB(); if (!late) stageA1(s1, BUF, 1, 1); KZV_SB();
        // ---- p4 ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            mm4(1, 0, i, Bf);
            KZV_SB();
            if (!last) { rdA(BUF ^ 1, 0, i); rdB(BUF ^ 1, 0, i, Bs); }
            KZV_SB();
        }
        mm2(1, 0, 2, 0, Bf); KZV_SB(); if (late) stageB1(s1, BUF, 1, 0); KZV_SB();
        mm2(1, 0, 2, 1, Bf); KZV_SB(); if (late) stageB1(s1, BUF, 1, 1); if (!last) rdA(BUF ^ 1, 0, 2); KZV_SB();
        mm2(1, 0, 3, 0, Bf); KZV_SB(); if (late) stageA1(s1, BUF, 1, 0); KZV_SB();
        mm2(1, 0, 3, 1, Bf); KZV_SB(); if (late) stageA1(s1, BUF, 1, 1); if (!last) rdA(BUF ^ 1, 0, 3); KZV_SB();
        advance(s1);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    // ---- drain: this wave's 128x64 accumulators -> global, through its private LDS patch ----
    // bf16 output without a second operand (the BF16 epilogue of interior tiles): bias added and converted BEFORE the transposition,
    // so a row block is 2 KiB in the patch and two of them alternate -- block i + 1 is written while block i is read back.  The
    // fp32 drain below is one LDS write -> read round trip per row block, 8 in a row (2.2 us per tile, all of it latency).
    auto drain_bf16 = [&](int tm, int tn) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        char* patch = smem + RING_BYTES + w * 4096;                 // [2][16 rows][128 B], 8-byte chunks XOR row
        const int l15 = ln & 15, g = ln >> 4;
        const int prow = ln >> 4, pchunk = ln & 15;                 // read-back: 4 rows x 128 B per wave-instruction
        const int nb0 = tn * 256 + wc * 64;
        float bj[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) t = *(const float4*)(p.bias + nb0 + j * 16 + 4 * g);
            bj[j][0] = t.x; bj[j][1] = t.y; bj[j][2] = t.z; bj[j][3] = t.w;
        }
        bf16_t* crow = (bf16_t*)p.C + (int64_t)(tm * 256 + wr * 128 + prow) * p.ldc + nb0 + pchunk * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            char* buf = patch + (i & 1) * 2048;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = acc[i][j];
                *(uint2*)(buf + l15 * 128 + (((j * 4 + g) ^ l15) << 3)) = make_uint2(pack_bf2(a[0] + bj[j][0], a[1] + bj[j][1]), pack_bf2(a[2] + bj[j][2], a[3] + bj[j][3]));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = q * 4 + prow;
                const uint2 v = *(const uint2*)(buf + row * 128 + ((pchunk ^ row) << 3));
                nt_st((uint2*)(crow + (int64_t)(i * 16 + q * 4) * p.ldc), v);
            }
        }
    };
    auto drain = [&](int tm, int tn, auto interiorc) {
        constexpr bool interior = decltype(interiorc)::value;
        if constexpr (interior && EPI == KZV_EPI_BF16 && !F8) { if (strip_in & 0x100) { drain_bf16(tm, tn); return; } }
        int ln = lane;
        asm volatile("" : "+v"(ln));       // as in set_tile: keep the drain's address terms out of the K loop's live set
        // patch = one accumulator row block: [16 rows][64 cols] fp32 (256-B rows), 16-B chunks XOR (row & 15)
        float* patch = (float*)(smem + RING_BYTES + w * 4096);
        const int prow = ln >> 4, pchunk = ln & 15;           // read-back: 4 rows x 256 B per wave-instruction
        const int l15 = ln & 15, g = ln >> 4;
        // wave columns: accumulator column block j (nh = j >> 1) sits at wc*64 + nh*32 + (j&1)*16 = wc*64 + j*16
        const int n0 = tn * 256 + wc * 64 + pchunk * 4;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        bool nv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) nv[r] = interior || n0 + r < p.n_valid;
        if (EPI != KZV_EPI_DGELU && p.bias) {                 // before any store (a later load could only be waited for with them)
            if constexpr (interior) { const float4 t = *(const float4*)(p.bias + n0); b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w; }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nv[r]) b4[r] = p.bias[n0 + r];
            }
        }
        // fp8: accumulator * a_scale[row] * b_scale[column]; the GELU output's e4m3 copy uses the per-tensor multiplier qs
        float sw4[4] = {1.f, 1.f, 1.f, 1.f};
        float qs = 0.f, amax = 0.f;
        if constexpr (F8) {
            if constexpr (interior) { const float4 t = *(const float4*)(p.b_scale + n0); sw4[0] = t.x; sw4[1] = t.y; sw4[2] = t.z; sw4[3] = t.w; }
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nv[r]) sw4[r] = p.b_scale[n0 + r];
            }
            if (EPI == KZV_EPI_GELU && p.c8) qs = *p.c8_qscale;
        }
        auto emit8 = [&](int m, const float (&y)[4], float q) {   // e4m3 copy of a finished row group (plain stores: L2 merges the 64-B pieces)
            if (EPI == KZV_EPI_GELU) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3]))));
            *(unsigned*)(p.c8 + (int64_t)m * p.ldc8 + n0) = pack_fp8x4(y[0] * q, y[1] * q, y[2] * q, y[3] * q);
        };
        auto block_loads = [&](int i, float4 (&r4)[4], uint2 (&u2)[4], float (&sa)[4], float (&rq)[4]) {
            const int m0 = tm * 256 + wr * 128 + i * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + q * 4 + prow;
                if (EPI == KZV_EPI_RESID) r4[q] = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
                if (EPI == KZV_EPI_DGELU) u2[q] = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
                if (F8) sa[q] = p.a_scale[m];
                if (F8 && EPI == KZV_EPI_DGELU) rq[q] = p.c8_rowq[m];      // (the launcher insists on c8 + c8_rowq for DGELU)
            }
        };
        // interior tiles: branch-free (counted vmcnt; see gemm_nt256.hip), the residual / pre-activation loads running
        // LOOK row blocks ahead of their use; edge tiles: guarded, row by row
        constexpr int LOOK = (F8 && (EPI == KZV_EPI_RESID || EPI == KZV_EPI_DGELU)) ? 3 : 4;     // fp8 + residual: one block less in flight (the row scales need registers too)
        float4 r4[8][4]; uint2 u2[8][4]; float sa[8][4], rq[8][4];
        if constexpr (interior) {
#pragma unroll
            for (int i = 0; i < LOOK; ++i) block_loads(i, r4[i], u2[i], sa[i], rq[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                         // accumulator row block i: tile rows wr*128 + i*16 .. +15
            const int m0 = tm * 256 + wr * 128 + i * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int chunk = j * 4 + g;
                *(f32x4*)(patch + l15 * 64 + ((chunk ^ l15) << 2)) = acc[i][j];
            }
            if constexpr (interior) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 4 + prow;
                    const f32x4 a4 = *(const f32x4*)(patch + row * 64 + ((pchunk ^ row) << 2));
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = F8 ? fmaf(a4[r], sa[i][q] * sw4[r], b4[r]) : a4[r] + b4[r];
                    nt_emit<EPI>(p, m0 + row, n0, v, r4[i][q], u2[i][q]);
                    if constexpr (F8 && EPI == KZV_EPI_GELU) { if (p.c8) emit8(m0 + row, v, qs); }
                    if constexpr (F8 && EPI == KZV_EPI_DGELU) emit8(m0 + row, v, rq[i][q]);
                }
                if (i + LOOK < 8) block_loads(i + LOOK, r4[i + LOOK], u2[i + LOOK], sa[i + LOOK], rq[i + LOOK]);
            } else {
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    const int row = q * 4 + prow;
                    const int m = m0 + row;
                    const f32x4 a4 = *(const f32x4*)(patch + row * 64 + ((pchunk ^ row) << 2));
                    if (m < p.M && n0 < p.N) {
                        float4 e4 = make_float4(0, 0, 0, 0); uint2 eu = make_uint2(0, 0);
                        if (EPI == KZV_EPI_RESID) e4 = *(const float4*)(p.resid + (int64_t)m * p.ldr + n0);
                        if (EPI == KZV_EPI_DGELU) eu = *(const uint2*)(p.aux + (int64_t)m * p.ldaux + n0);
                        const float sr = F8 ? p.a_scale[m] : 1.f;
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = nv[r] ? (F8 ? fmaf(a4[r], sr * sw4[r], b4[r]) : a4[r] + b4[r]) : 0.f;
                        nt_emit<EPI>(p, m, n0, v, e4, eu);
                        if constexpr (F8 && EPI == KZV_EPI_GELU) { if (p.c8) emit8(m, v, qs); }
                        if constexpr (F8 && EPI == KZV_EPI_DGELU) emit8(m, v, p.c8_rowq[m]);
                    }
                }
            }
        }
        if constexpr (F8 && EPI == KZV_EPI_GELU) {
            // one atomic per wave at most, and none once the running maximum is above this tile's (floats >= 0 order as integers)
            if (p.c8) {
                amax = wave_max(amax);
                if (ln == 0 && amax > *(volatile float*)p.c8_amax) atomicMax((unsigned*)p.c8_amax, __float_as_uint(amax));
            }
        }
    };

    // ---- the stream ----
    set_tile(s0, 0); set_tile(s1, 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {                   // stream K-tiles 0 and 1 (nk >= 2: both of tile 0)
        stageA1(s0, b, 0, 0); stageA1(s0, b, 0, 1); stageB1(s0, b, 0, 0); stageB1(s0, b, 0, 1);
        stageB1(s1, b, 1, 0); stageB1(s1, b, 1, 1); stageA1(s1, b, 1, 0); stageA1(s1, b, 1, 1);
        advance(s0); advance(s1);
    }
    vmcnt<12>();                                    // A-h0(0), B-h0(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    auto first_frags = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) rdA(0, 0, i);
        rdB(0, 0, 0, fbX); rdB(0, 0, 1, fbX);
    };
    first_frags();
#ifdef KZV_F_NOREAD
    rd_on = false;
#endif
    bool credit = false;                            // previous drain was of an interior tile
#ifdef KZV_STAMPS
    // per-block stamps: [blockIdx][16]: start, then (K loop end, drain end) per tile
    unsigned long long* stp = (EPI == KZV_EPI_BF16 && tid == 0) ? (unsigned long long*)p.aux + blockIdx.x * 16 : nullptr;
    int stk = 0;
#define KZV_STAMP() do { if (stp) stp[stk++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KZV_STAMP() do {} while (0)
#endif
    KZV_STAMP();
#ifdef KZV_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    for (int seq = 0; ; ++seq) {
        const int id = seq * G + vblk;
        if (id >= tiles) break;
        int tm, tn;
        nt_tile_coords(id, tiles / tilesN, tilesN, strip, tm, tn);
        tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
        // nk is even (launcher): every tile starts on ring buffer 0 with B0 in fbX, so the two K-tile bodies alternate statically
        for (int kt = 0; kt < nk; kt += 2) {
            const bool c0 = credit && kt == 0;
            ktile(I0{}, fbX, fbY, c0, c0, false);
            ktile(I1{}, fbY, fbX, c0, false, kt + 2 >= nk);
        }
        KZV_STAMP();
        credit = tm * 256 + 256 <= p.M && tn * 256 + 256 <= p.n_valid;     // interior tile (n_valid <= N)
        if (credit) drain(tm, tn, std::true_type{}); else drain(tm, tn, std::false_type{});
        zero_acc();
        first_frags();                              // A0 / B0 of the next tile's first K-tile (retired before the drain)
        KZV_STAMP();
    }
#ifdef KZV_STAMPS
    if (stp) { stp[14] = clk0; stp[15] = __builtin_amdgcn_s_memtime(); }
#endif
    vmcnt<0>();                                     // the refills issued past the end of the stream (into dead slots) land before the LDS is released
}

int nt256f_min_tiles() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("KZV_NT256P_MIN_TILES"); v = e ? atoi(e) : 384; }
    return v;
}
int device_cus_f() {
    static int v = -1;
    if (v < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        v = n;
    }
    return v;
}

}  // namespace

int kzv_nt256f_launch(const NtParams& p, int epilogue, hipStream_t s) {
    const int tilesN = (p.N + 255) / 256;
    const int tiles = ((p.M + 255) / 256) * tilesN;
    if (p.K < 128 || p.K % 128 || tiles < nt256f_min_tiles()) return 0;   // even number of K-tiles (odd: gemm_nt256.hip)
    if ((uint64_t)256 * (uint64_t)p.lda * 2 > 0xffffffffull || (uint64_t)p.n_valid * (uint64_t)p.ldb * 2 > 0xffffffffull) return 0;   // 32-bit DMA offsets
    if (p.n_valid <= (tilesN - 1) * 256 || p.lda * 2 < 128 || p.ldb * 2 < 128) return 0;          // every tile starts on a valid column (the DMA clamp needs one)
    const int grid = tiles < device_cus_f() ? tiles : device_cus_f();
#define KZV_NT256F_CASE(E)                                                                                          \
    case E: {                                                                                                       \
        static bool attr_done = false;                                                                              \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt256f_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done = true; } \
        hipLaunchKernelGGL((gemm_nt256f_kernel<E>), dim3(grid), dim3(512), LDS_BYTES, s, p, tiles, tilesN, kzv_nt_strip());         \
    } break;
    switch (epilogue) {
        KZV_NT256F_CASE(KZV_EPI_BF16) KZV_NT256F_CASE(KZV_EPI_F32) KZV_NT256F_CASE(KZV_EPI_GELU)
        KZV_NT256F_CASE(KZV_EPI_RESID) KZV_NT256F_CASE(KZV_EPI_DGELU) KZV_NT256F_CASE(KZV_EPI_GELU_F32)
        default: return 0;
    }
#undef KZV_NT256F_CASE
    return 1;
}
