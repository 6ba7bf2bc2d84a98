// gemm_nt, persistent 256x256 eight-phase kernel for gfx950 (MI355X).
//
// Same tile, LDS ring, phase schedule and ping-pong as gemm_nt256.hip (read its header first); what changes:
//
//  * ONE workgroup per CU walks a sequence of output tiles (tile = seq * gridDim + virtual block), and the K-tiles of
//    consecutive output tiles form ONE stream through the LDS ring: while the last K-tiles of tile i are multiplied,
//    the half-tiles of tile i+1 are already in flight, so no tile pays a cold prologue (memory latency with the MFMA
//    pipe idle) after the first.
//  * The epilogue never touches the ring: each wave drains its own accumulators through a wave-private 4-KiB LDS
//    patch (bytes 128K..160K of the CU's LDS) -- one accumulator row block (16 rows x the wave's 64 columns, fp32)
//    at a time is written in MFMA layout, read back row-major (4 rows x 256 B per wave-instruction) and stored /
//    combined with the residual or the saved pre-activation.  No workgroup barrier is involved, so the two wave groups drain half a phase apart
//    and the next tile's loads keep landing meanwhile.
//
// vmcnt bookkeeping across a drain.  vmcnt counts this wave's VMEM operations in issue order (loads, stores, atomics
// and LDS-DMA count together and retire in order: MI355X_MICROARCH.md, `s_waitcnt vmcnt(N)`).  A wait that must retire an LDS-DMA load issued BEFORE the drain may leave
// outstanding every operation issued after that load: the usual 8 (four half-tiles) plus the D loads/stores of the
// drain.  D is only credited for interior tiles, where every row and column is stored (an edge tile skips some
// stores; crediting too few is merely conservative, crediting too many would be a race).
//
// F8 = true: the same schedule on e4m3 operands.  A K-tile is still 128 BYTES per row (128 fp8 elements instead of 64 bf16),
// so the LDS ring, the LDS-DMA pieces, the swizzle and every ds_read_b128 are byte-for-byte those of the bf16 kernel; the two
// 16-byte fragments a lane reads per operand row (K-chunks g and 4+g) are the 32 bytes of ONE block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 (scales = 2^0) in place of two v_mfma_f32_16x16x32_bf16.  The instruction sums over its
// 128 k-positions whichever position a byte sits in, and A and B are permuted alike, so no re-ordering is needed.  Same MFMA
// cycles per K-tile, half as many K-tiles: twice the bf16 rate in the main loop.  The drain multiplies each accumulator by
// a_scale[row] * b_scale[column] (per-row quantisation of both operands) before the bias.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "gemm_nt.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HT_BYTES = 128 * 128;        // half-tile: 128 rows x 64 bf16
constexpr int BUF_BYTES = 4 * HT_BYTES;    // A-h0, A-h1, B-h0, B-h1
constexpr int RING_BYTES = 2 * BUF_BYTES;  // 128 KiB
constexpr int LDS_BYTES = RING_BYTES + 8 * 4096;   // + one 4-KiB drain patch per wave = 160 KiB
constexpr int KA0 = 0, KB0 = 2;             // half-tile slots of a ring buffer: A-h0, A-h1, B-h0, B-h1

__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// D += A(e4m3, 16x128) . B(e4m3, 128x16) with unit block scales.  Inline asm with the accumulator TIED to the result: through the
// builtin hipcc gives every scaled MFMA a fresh result tuple (no in-place form), which with 128 accumulator registers spills
// half of them.  hipcc pads no hazards around asm: the caller keeps VALU reads of the accumulators >= 18 wait states behind the
// last MFMA (s_nop before the drain); operands come from LDS reads, which the compiler still waits for (they are asm inputs).
__device__ __forceinline__ void mfma_f8(f32x4& acc, const i32x8& a, const i32x8& b, int one_scale) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(one_scale));
}
template <int N> __device__ __forceinline__ void vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
    else if constexpr (N == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    else if constexpr (N == 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}
constexpr int cmin(int a, int b) { return a < b ? a : b; }

// VMEM operations one wave issues while draining an interior tile (32 four-column groups per lane)
template <int EPI, bool F8> constexpr int drain_ops() {
    // GELU*: two stores; RESID/DGELU: load + store; fp8: + the row scale (and the e4m3 copy of a GELU output)
    return ((EPI == KZV_EPI_BF16 || EPI == KZV_EPI_F32) ? 32 : 64) + (F8 ? ((EPI == KZV_EPI_GELU || EPI == KZV_EPI_DGELU) ? 64 : 32) : 0);
}

struct TileSrc {            // where the next half-tiles of one half index (h) come from
    const char* a; const char* b;      // wave-uniform bases (A: tile row panel; B: absolute)
    unsigned va[2], vb[2];             // per-lane byte offsets of this wave's two 1-KiB pieces
    int kt, seq; bool valid;
};

// The kernel body as a device function (workgroup `bid` of `G`, the 160 KiB of dynamic LDS in `smem`): the plain kernel below
// wraps it; gemm_tn256.hip's dgrad + wgrad pair kernel runs it as its first phase.  Every wave leaves it with no LDS-DMA in flight
// and the barrier count balanced.
template <int EPI, bool F8>
__device__ __forceinline__ void nt256p_body(const NtParams& p, const int tiles, const int tilesN, const int strip_in, const int bid, const int G,
                                            char* const smem) {
    const int strip = strip_in & 0xff;             // bit 8: the double-buffered bf16 drain (A/B knob KZV_BF16_DRAIN, default on)
    constexpr int ES = F8 ? 1 : 2;                 // bytes per operand element
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wr = w >> 2, wc = w & 3;
    // blocks land on XCD (blockIdx % 8): give each XCD a contiguous run of every step's tiles (shared A row panels)
    const int vblk = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    const int nk = p.K * ES / 128;                 // K-tiles of 128 bytes per row: even, >= 2 (checked by the launcher)

    auto set_tile = [&](TileSrc& s, int seq, int h) {
        s.seq = seq; s.kt = 0;
        const int id = seq * G + vblk;
        s.valid = id < tiles;
        const int idc = s.valid ? id : 0;
        int tm, tn;
        nt_tile_coords(idc, tiles / tilesN, tilesN, strip, tm, tn);
        tm = __builtin_amdgcn_readfirstlane(tm); tn = __builtin_amdgcn_readfirstlane(tn);
        s.a = (const char*)p.A + (int64_t)tm * 256 * p.lda * ES;
        s.b = (const char*)p.B;
        int ln = lane;
        asm volatile("" : "+v"(ln));       // recompute the lane terms here: hoisted, they would live (and spill) across the K loop
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = j * 64 + w * 8 + (ln >> 3);
            const unsigned cb = (unsigned)(((ln & 7) ^ (r & 7)) * 16);
            int arow = (r >> 6) * 128 + h * 64 + (r & 63);
            arow = min(tm * 256 + arow, p.M - 1) - tm * 256;            // rows beyond M: clamp (never stored)
            s.va[j] = (unsigned)arow * (unsigned)(p.lda * ES) + cb;
            int bcol = (r >> 5) * 64 + h * 32 + (r & 31);
            bcol = min(tn * 256 + bcol, p.n_valid - 1);                 // columns beyond n_valid: clamp (stored as 0)
            s.vb[j] = (unsigned)bcol * (unsigned)(p.ldb * ES) + cb;
        }
    };
    auto advance = [&](TileSrc& s, int h) {
        if (++s.kt == nk) set_tile(s, s.seq + 1, h);
    };
    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)smem) + (unsigned)w * 1024u);
    auto stageA = [&](const TileSrc& s, int buf, int h) {
        const char* sb = s.a + (int64_t)s.kt * 128;
        const unsigned d = ldsw + (unsigned)(buf * BUF_BYTES + (KA0 + h) * HT_BYTES);
        glds16_s(s.va[0], sb, d); glds16_s(s.va[1], sb, d + 8192u);
    };
    auto stageB = [&](const TileSrc& s, int buf, int h) {
        const char* sb = s.b + (int64_t)s.kt * 128;
        const unsigned d = ldsw + (unsigned)(buf * BUF_BYTES + (KB0 + h) * HT_BYTES);
        glds16_s(s.vb[0], sb, d); glds16_s(s.vb[1], sb, d + 8192u);
    };

    f32x4 acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    const int sw = l15 & 7;
    const int slot0 = (g ^ sw) << 4, slot1 = ((4 + g) ^ sw) << 4;
    const int a_off = (wr * 64 + l15) * 128, b_off = (wc * 32 + l15) * 128;
    // fragments: bf16 = two 16-byte K-chunks per row used by two MFMAs; fp8 = the same two chunks as ONE 32-byte operand, kept
    // as an 8-register value from the read on (joined at each use, hipcc kept both forms alive and spilled the accumulators)
    struct FragBf { bf16x8 k[2]; };
    using Frag = std::conditional_t<F8, i32x8, FragBf>;
    Frag fa[4], fb0[2], fb1[2];
    int one_scale = 0x7f7f7f7f;                  // E8M0 2^0 in every byte (whichever one op_sel picks)
    asm volatile("" : "+v"(one_scale));          // a VGPR, set once (far from the first MFMA that reads it)
    auto rd = [&](const char* q) {
        if constexpr (F8) {
            const i32x4 lo = *(const i32x4*)(q + slot0), hi = *(const i32x4*)(q + slot1);
            return (i32x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            FragBf f;
            f.k[0] = *(const bf16x8*)(q + slot0); f.k[1] = *(const bf16x8*)(q + slot1);
            return f;
        }
    };
    auto readA = [&](int buf, int mh) {
        const char* b = smem + buf * BUF_BYTES + (KA0 + mh) * HT_BYTES + a_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = rd(b + i * 2048);
    };
    auto readB = [&](int buf, int nh, Frag (&fb)[2]) {
        const char* b = smem + buf * BUF_BYTES + (KB0 + nh) * HT_BYTES + b_off;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = rd(b + j * 2048);
    };
    auto mm = [&](int mh, int nh, const Frag (&fb)[2]) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (F8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    mfma_f8(acc[mh * 4 + i][nh * 2 + j], fb[j], fa[i], one_scale);
        } else {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mh * 4 + i][nh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j].k[kh], fa[i].k[kh], acc[mh * 4 + i][nh * 2 + j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    TileSrc s1, s2;                    // s1 feeds A-h1/B-h1 of stream K-tile u+1, s2 feeds A-h0/B-h0 of K-tile u+2
    constexpr int D = drain_ops<EPI, F8>();
    constexpr int W8 = cmin(63, 8 + D);

    // One K-tile of the stream = four phases (gemm_nt256.hip).  e1 / e2: stream K-tiles u+1 / u+2 exist.
    // `after_drain`: this is the first K-tile after a credited drain (waits widened by D).
    auto ktile = [&](auto bufc, bool after_drain, bool defer) {
        constexpr int BUF = decltype(bufc)::value;
        const bool e1 = s1.valid, e2 = s2.valid && !defer;
        // p1
        readA(BUF, 0); readB(BUF, 0, fb0);
        if (e1) { stageB(s1, BUF ^ 1, 1); if (after_drain) vmcnt<W8>(); else vmcnt<8>(); } else vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        mm(0, 0, fb0);
        __builtin_amdgcn_s_barrier();
        // p2
        readB(BUF, 1, fb1);
        if (e1) { stageA(s1, BUF ^ 1, 1); if (after_drain) vmcnt<W8>(); else vmcnt<8>(); } else vmcnt<0>();
        advance(s1, 1);
        __builtin_amdgcn_s_barrier();
        mm(0, 1, fb1);
        __builtin_amdgcn_s_barrier();
        // p3
        readA(BUF, 1);
        if (e2) stageA(s2, BUF, 0);
        __builtin_amdgcn_s_barrier();
        mm(1, 1, fb1);
        __builtin_amdgcn_s_barrier();
        // p4
        if (e2) { stageB(s2, BUF, 0); vmcnt<8>(); advance(s2, 0); }
        else if (e1) vmcnt<4>();                      // tail, or refills deferred past the drain: only p1/p2's are newer
        __builtin_amdgcn_s_barrier();
        mm(1, 0, fb0);
        __builtin_amdgcn_s_barrier();
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
                // once-read operands: streaming loads (they leave the operand panels in L2 alone): step 30.95 -> 30.80 ms over three same-box alternations
                if (EPI == KZV_EPI_RESID) { const f32x4 t = __builtin_nontemporal_load((const f32x4*)(p.resid + (int64_t)m * p.ldr + n0)); r4[q] = make_float4(t[0], t[1], t[2], t[3]); }
                if (EPI == KZV_EPI_DGELU) { typedef unsigned u32x2t __attribute__((ext_vector_type(2))); const u32x2t t = __builtin_nontemporal_load((const u32x2t*)(p.aux + (int64_t)m * p.ldaux + n0)); u2[q] = make_uint2(t[0], t[1]); }
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
    set_tile(s2, 0, 0); set_tile(s1, 0, 1);
    stageA(s2, 0, 0); stageB(s2, 0, 0); stageB(s1, 0, 1); stageA(s1, 0, 1);
    advance(s2, 0);                                 // nk >= 2: still tile 0, K-tile 1
    stageA(s2, 1, 0); stageB(s2, 1, 0);
    advance(s2, 0); advance(s1, 1);
    vmcnt<8>();                                     // A-h0(0), B-h0(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();      // waves 4..7 run one barrier interval behind
    bool credit = false;                            // previous drain was of an interior tile
#ifdef KZV_STAMPS
    // per-block stamps: [blockIdx][16]: start, then (K loop end, drain end) per tile
    unsigned long long* stp = (EPI == KZV_EPI_BF16 && tid == 0) ? (unsigned long long*)p.aux + bid * 16 : nullptr;
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
        // nk is even (launcher): every tile starts on ring buffer 0, so the two K-tile bodies alternate statically
        // (selecting the body by a run-time parity made hipcc spill half the accumulators)
        for (int kt = 0; kt < nk; kt += 2) {
            ktile(I0{}, credit && kt == 0, false);
            ktile(I1{}, false, kt + 2 >= nk);        // last K-tile of the tile: its A-h0/B-h0 refills wait for the drain
        }
        KZV_STAMP();
        if constexpr (F8) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // the asm MFMAs' results are read next (mfma_f8)
        credit = tm * 256 + 256 <= p.M && tn * 256 + 256 <= p.n_valid;     // interior tile (n_valid <= N)
        if (credit) drain(tm, tn, std::true_type{}); else drain(tm, tn, std::false_type{});
        // the refills deferred by the last K-tile (stream K-tile u+2 -> ring buffer 1): issued only now, so that the
        // drain's own loads (bias, residual, pre-activation), which retire in order behind every earlier LDS-DMA,
        // never wait on a load issued moments before
        if (s2.valid) { stageA(s2, 1, 0); stageB(s2, 1, 0); }
        advance(s2, 0);
        zero_acc();
        KZV_STAMP();
    }
#ifdef KZV_STAMPS
    if (stp) { stp[14] = clk0; stp[15] = __builtin_amdgcn_s_memtime(); }
#endif
    if (wr == 0) __builtin_amdgcn_s_barrier();      // balance the barrier count
}

template <int EPI, bool F8>
__global__ __launch_bounds__(512) void gemm_nt256p_kernel(const NtParams p, const int tiles, const int tilesN, const int strip_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    nt256p_body<EPI, F8>(p, tiles, tilesN, strip_in, (int)blockIdx.x, (int)gridDim.x, smem);
}

}  // namespace
#ifndef KZV_NT256P_BODY_ONLY
int kzv_nt_strip() {
    static int v = -1000;
    if (v == -1000) {
        const char* e = getenv("KZV_NT_STRIP"); v = e ? atoi(e) : 3;
        if (v < 0 || v > 255) v = 0;
        const char* d = getenv("KZV_BF16_DRAIN");
        if (!d || atoi(d) != 0) v |= 0x100;
    }
    return v;
}
namespace {
int nt256p_min_tiles() {
    static int v = -1;
    // 150 since the end of round 4 (384 before): the two 160-tile projections between encoder and decoder (41k x 256 x 768 forward, x 3072 input
    // gradient) and the decoder's 180-tile shapes run faster on fewer than 256 persistent workgroups than on the 128 x 128 kernel: step
    // 30.65 -> 30.58 ms, family +0.002 over five same-box alternations
    if (v < 0) { const char* e = getenv("KZV_NT256P_MIN_TILES"); v = e ? atoi(e) : 150; }
    return v;
}
int device_cus() {
    static int v = -1;
    if (v < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        v = n;
    }
    return v;
}

}  // namespace

int kzv_nt256p_launch(const NtParams& p, int epilogue, hipStream_t s) {
    const int tilesN = (p.N + 255) / 256;
    const int tiles = ((p.M + 255) / 256) * tilesN;
    if (p.K < 128 || p.K % 128 || tiles < nt256p_min_tiles()) return 0;   // even number of K-tiles (odd: gemm_nt256.hip)
    if ((uint64_t)256 * (uint64_t)p.lda * 2 > 0xffffffffull || (uint64_t)p.n_valid * (uint64_t)p.ldb * 2 > 0xffffffffull) return 0;   // 32-bit DMA offsets
    int grid = tiles < device_cus() ? tiles : device_cus();
    { const char* e = getenv("KZV_NT_GRID"); const int g = e ? atoi(e) : 0; if (g > 0 && g < grid) grid = g; }   // dev: fewer persistent workgroups (two-chain experiment)
#define KZV_NT256P_CASE(E)                                                                                          \
    case E: {                                                                                                       \
        static bool attr_done = false;                                                                              \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<E, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done = true; } \
        hipLaunchKernelGGL((gemm_nt256p_kernel<E, false>), dim3(grid), dim3(512), LDS_BYTES, s, p, tiles, tilesN, kzv_nt_strip());         \
    } break;
    switch (epilogue) {
        KZV_NT256P_CASE(KZV_EPI_BF16) KZV_NT256P_CASE(KZV_EPI_F32) KZV_NT256P_CASE(KZV_EPI_GELU)
        KZV_NT256P_CASE(KZV_EPI_RESID) KZV_NT256P_CASE(KZV_EPI_DGELU) KZV_NT256P_CASE(KZV_EPI_GELU_F32)
        default: return 0;
    }
#undef KZV_NT256P_CASE
    return 1;
}

int kzv_nt256p_fp8_launch(const NtParams& p, int epilogue, hipStream_t s) {
    const int tilesN = (p.N + 255) / 256;
    const int tiles = ((p.M + 255) / 256) * tilesN;
    if (p.K < 256 || p.K % 256) return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: K must be a multiple of 256 (an even number of 128-byte K-tiles)");
    if ((uint64_t)256 * (uint64_t)p.lda > 0xffffffffull || (uint64_t)p.n_valid * (uint64_t)p.ldb > 0xffffffffull)
        return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: operand panel beyond the 32-bit LDS-DMA offsets");
    const int grid = tiles < device_cus() ? tiles : device_cus();
#define KZV_NT256P8_CASE(E)                                                                                         \
    case E: {                                                                                                       \
        static bool attr_done = false;                                                                              \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<E, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done = true; } \
        hipLaunchKernelGGL((gemm_nt256p_kernel<E, true>), dim3(grid), dim3(512), LDS_BYTES, s, p, tiles, tilesN, kzv_nt_strip());   \
    } break;
    switch (epilogue) {
        KZV_NT256P8_CASE(KZV_EPI_BF16) KZV_NT256P8_CASE(KZV_EPI_GELU) KZV_NT256P8_CASE(KZV_EPI_RESID) KZV_NT256P8_CASE(KZV_EPI_DGELU)
        default: return kzv_fail(KZV_E_ARG, "gemm_nt_fp8: epilogue must be BF16, GELU, RESID or DGELU");
    }
#undef KZV_NT256P8_CASE
    return KZV_OK;
}
#endif  // KZV_NT256P_BODY_ONLY
