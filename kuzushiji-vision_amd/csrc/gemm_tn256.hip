// gemm_tn, 256x256 tile, eight-phase ping-pong schedule for gfx950 (MI355X).
//
//  OUT[n][k] += sum_t P[t][n] * Q[t][k]   (every large weight gradient dW = dY^T . X; optional dbias[n] += sum_t P[t][n])
//
// The schedule is gemm_nt256.hip's (read its header): one 512-thread workgroup per CU, 8 waves (2 per SIMD), wave
// (wr, wc) owns output rows n = wr*128..+127 and columns k = wc*64..+63 (8x4 MFMA accumulators), an LDS ring of two
// reduction stages x four 16-KiB half-tiles, one accumulator quadrant and one half-tile of LDS-DMA per phase, counted
// `vmcnt(8)` before a barrier, the first read one phase later, waves 4..7 one barrier interval behind waves 0..3.
// What differs is the operand geometry: the reduction index t is the ROW index of both operands (as in gemm.hip's
// gemm_tn), so
//   * a half-tile is 64 tokens x 128 contiguous tile columns (256-B rows = two full cache lines per DMA row):
//     P-h{h} / Q-h{h} hold tile columns h*128 .. +127, and
//     wave (wr, wc) owns output rows {mh*128 + wr*64 + 0..63} and columns {nh*128 + wc*32 + 0..31}, mh, nh in {0, 1};
//     slot s of row r holds source chunk s ^ ((r & 7) << 1);
//   * fragments come out of LDS through ds_read_b64_tr_b16 (two per fragment) with gemm_tn's permuted token mapping;
//   * the reduction is split over token ranges (one workgroup per CU in total).  A 256x256 fp32 partial tile per
//     workgroup is 62-66 MB per launch: as float atomics (~1 TB/s) that cost 70 us and ate the main-loop gain, so the
//     partials are written with plain streaming stores (1-KiB row segments through LDS) into a workspace and a second
//     small kernel folds the splits into OUT (reads 62 MB at HBM rate).
// The launcher takes only shapes with >= 9 output tiles (qkv, fc1, fc2, output projection of a ViT layer); everything else stays on the
// 128x128 kernel with its atomic epilogue.
//
// Measured (MI355X, 41216 tokens): a reduction stage takes ~2.0 us here against 1.48 us for a K-tile of gemm_nt256 with the
// same MFMA and LDS work and 0 bank conflicts (SQ_LDS_IDX_ACTIVE equal, 2.0 cycles per transposing read).  Not memory
// latency: a 10-slot ring over all 160 KiB of LDS (every half-tile issued two whole stages ahead, vmcnt(12)) was 5 %
// SLOWER, and contiguous vs interleaved half-tile columns made no difference.  What remains is the doubled LDS instruction
// count of the transposing reads in the load parts.  Net against the 128x128 kernel: qkv 181 -> 176 us, fc1 224 -> 203,
// fc2 229 -> 197 (-1 ms per training step).
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "gemm_tn.h"
#include "gemm_nt.h"
#include <vector>
#include <cstdlib>
#include <type_traits>

// the persistent gemm_nt kernel's body, for the dgrad + wgrad pair kernel below (kernels of a translation unit cannot call into
// another one's: the source is compiled here a second time, device functions only)
namespace kzv_pair_nt {
#define KZV_NT256P_BODY_ONLY
#include "gemm_nt256p.hip"
#undef KZV_NT256P_BODY_ONLY
}  // namespace kzv_pair_nt

namespace {

constexpr int HT_BYTES = 64 * 256;         // half-tile: 64 tokens x 128 bf16
constexpr int BUF_BYTES = 4 * HT_BYTES;    // P-h0, P-h1, Q-h0, Q-h1
constexpr int LDS_BYTES = 2 * BUF_BYTES;   // 128 KiB
constexpr int KP0 = 0, KP1 = 1, KQ0 = 2, KQ1 = 3;
// floats of workspace per 256 x 256 partial tile: bf16 partials (default) take half a float per element
#ifndef KZV_TN_F32_PARTIALS
constexpr size_t PART_FLOATS = 32768;
#else
constexpr size_t PART_FLOATS = 65536;
#endif

__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N == 0, "add the vmcnt literal");
}

__global__ __launch_bounds__(512) void gemm_tn256_kernel(const TnParams p, float* part_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wr = w >> 2, wc = w & 3;
    const int tilesN = (p.N + 255) / 256, tilesK = (p.K + 255) / 256;
    const int per = tilesN * tilesK;
    const int id = xcd_remap(blockIdx.x, per * p.splits);
    const int split = id / per, rem = id - split * per;
    const int tnb = rem / tilesK, tkb = rem - tnb * tilesK;
    const int t_begin = split * p.chunk;                      // chunk and Mtok are multiples of 64 (launcher)
    const int t_end = min(p.Mtok, t_begin + p.chunk);
    const int nt = (t_end - t_begin) >> 6;                    // >= 2 (launcher)

    // ---- LDS-DMA sources: wave w fills 1-KiB pieces w and w+8 of every half-tile (4 token rows x 256 B each) ----
    // Columns beyond N / K are clamped to the last valid 8-column chunk: their products are never stored.
    const char* baseP = (const char*)(p.P + (int64_t)t_begin * p.ldp);
    const char* baseQ = (const char*)(p.Q + (int64_t)t_begin * p.ldq);
    const int64_t stepP = 64 * p.ldp * 2, stepQ = 64 * p.ldq * 2;
    unsigned voff[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (w + 8 * j) * 4 + (lane >> 4);
        const int col = (((lane & 15) ^ ((r & 7) << 1))) * 8;           // first of this lane's 8 local columns
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int n = min(tnb * 256 + h * 128 + col, p.N - 8);
            voff[KP0 + h][j] = (unsigned)r * (unsigned)(p.ldp * 2) + (unsigned)n * 2u;
            const int k = min(tkb * 256 + h * 128 + col, p.K - 8);
            voff[KQ0 + h][j] = (unsigned)r * (unsigned)(p.ldq * 2) + (unsigned)k * 2u;
        }
    }
    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)smem) + (unsigned)w * 1024u);
    auto stage = [&](int buf, int kind, int kt) {
        const char* sb = kind < 2 ? baseP + (int64_t)kt * stepP : baseQ + (int64_t)kt * stepQ;
        const unsigned d = ldsw + (unsigned)(buf * BUF_BYTES + kind * HT_BYTES);
        glds16_s(voff[kind][0], sb, d);
        glds16_s(voff[kind][1], sb, d + 8192u);
    };

    f32x4 acc[4][8];                                          // [k-frag][n-frag]: D rows = n (4g+r), D cols = k (l15)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposing reads (gemm.hip gemm_tn): lane supplies token row 4g + (l15>>2) [+16], columns c0 + (l15&3)*4 .. +3
    const int trow = 4 * g + (l15 >> 2);
    const int tsw0 = (trow & 7) << 1;
    const int sub8 = (l15 & 1) * 8;
    const int cq = (l15 >> 1) & 1;
    bf16x8 fp[4][2], fq0[2][2], fq1[2][2];                    // [frag][token half ks]
    // (the swizzle term is laundered per call: hoisted, the six XOR-ed fragment addresses per ring buffer stay live across
    // the whole loop and push the kernel past 256 VGPRs; recomputing them is a handful of VALU ops in the load part)
    auto readP = [&](int buf, int mh) {
        const char* b = smem + buf * BUF_BYTES + (KP0 + mh) * HT_BYTES;
        int tsw = tsw0;
        asm volatile("" : "+v"(tsw));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r0 = (ks * 32 + trow) * 256, r1 = r0 + 16 * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int chunk = (wr * 8 + j * 2 + cq) ^ tsw;
                const bf16x4 lo = lds_tr16(b + r0 + chunk * 16 + sub8), hi = lds_tr16(b + r1 + chunk * 16 + sub8);
                fp[j][ks] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
    };
    auto readQ = [&](int buf, int nh, bf16x8 (&fq)[2][2]) {
        const char* b = smem + buf * BUF_BYTES + (KQ0 + nh) * HT_BYTES;
        int tsw = tsw0;
        asm volatile("" : "+v"(tsw));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r0 = (ks * 32 + trow) * 256, r1 = r0 + 16 * 256;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int chunk = (wc * 4 + i * 2 + cq) ^ tsw;
                const bf16x4 lo = lds_tr16(b + r0 + chunk * 16 + sub8), hi = lds_tr16(b + r1 + chunk * 16 + sub8);
                fq[i][ks] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
    };
    auto mma = [&](int mh, int nh, const bf16x8 (&fq)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[nh * 2 + i][mh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp[j][ks], fq[i][ks], acc[nh * 2 + i][mh * 4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // Bias gradient = column sums of P.  Each (n-tile, split) P panel is streamed by tilesK workgroups; workgroup tkb takes
    // the stages t with t % tilesK == tkb, waves with wc == 0 only (as gemm.hip).  The A fragment of lane (g, l15) is row
    // n = l15 of its 16-row block with 8 of the 32 tokens, so the sum is VALU work on fragments already in registers:
    // one float per block instead of the 4-register MFMA accumulator the 128x128 kernel spends (this kernel has none left).
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool bias_wave = p.dbias != nullptr && wc == 0;
    auto bias_add = [&](int mh) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 u = __builtin_bit_cast(u32x4, fp[j][ks]);
                float sacc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) sacc += __uint_as_float(u[q] << 16) + __uint_as_float(u[q] & 0xffff0000u);
                bsum[mh * 4 + j] += sacc;
            }
    };

    auto ktile = [&](auto bufc, int t) {
        constexpr int BUF = decltype(bufc)::value;
        const bool s1 = t + 1 < nt, s2 = t + 2 < nt;
        const bool wb = bias_wave && (t % tilesK) == tkb;
        // p1
        readP(BUF, 0); readQ(BUF, 0, fq0);
        if (s1) { stage(BUF ^ 1, KQ1, t + 1); vmcnt<8>(); } else vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        mma(0, 0, fq0);
        if (wb) bias_add(0);
        __builtin_amdgcn_s_barrier();
        // p2
        readQ(BUF, 1, fq1);
        if (s1) { stage(BUF ^ 1, KP1, t + 1); vmcnt<8>(); } else vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        mma(0, 1, fq1);
        __builtin_amdgcn_s_barrier();
        // p3
        readP(BUF, 1);
        if (s2) stage(BUF, KP0, t + 2);
        __builtin_amdgcn_s_barrier();
        mma(1, 1, fq1);
        if (wb) bias_add(1);
        __builtin_amdgcn_s_barrier();
        // p4
        if (s2) { stage(BUF, KQ0, t + 2); vmcnt<8>(); } else if (s1) vmcnt<4>();
        __builtin_amdgcn_s_barrier();
        mma(1, 0, fq0);
        __builtin_amdgcn_s_barrier();
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

    stage(0, KP0, 0); stage(0, KQ0, 0); stage(0, KQ1, 0); stage(0, KP1, 0); stage(1, KP0, 1); stage(1, KQ0, 1);
    vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();     // waves 4..7 run one barrier interval behind
    for (int t = 0; t < nt; t += 2) {
        ktile(I0{}, t);
        if (t + 1 < nt) ktile(I1{}, t + 1);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();     // balance the barrier count

    if (bias_wave) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = tnb * 256 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16 + l15;
            if (g == 0 && n < p.n_store) atomicAdd(p.dbias + n, v);
        }
    }

    // ---- epilogue: two chunks of 128 output rows (chunk c = tile rows c*128 .. +127 = the mh == c accumulators) through LDS
    // (fp32 [128][256], 16-B chunks XOR (row & 31)) -> this workgroup's partial tile in the workspace, one 1-KiB row
    // per wave-instruction, streamed past L2 ----
    float* tile = (float*)smem;
    float* part = part_ws + (size_t)id * PART_FLOATS;                       // [256][256] fp32 per workgroup (id = split * per + tile)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        __syncthreads();                           // the ring / the previous chunk is no longer being read
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ln = wr * 64 + j * 16 + 4 * g + r;                 // local row: tile row c*128 + ln
                    const int k = (i >> 1) * 128 + wc * 32 + (i & 1) * 16 + l15;
                    tile[ln * 256 + ((((k >> 2) ^ (ln & 31)) << 2) | (k & 3))] = acc[i][c * 4 + j][r];
                }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {          // wave w: local rows w*16 .. +15
            const int ln = w * 16 + it;
            const f32x4 v = *(const f32x4*)(tile + ln * 256 + ((lane ^ (ln & 31)) << 2));
            const int row = c * 128 + ln;
#ifndef KZV_TN_F32_PARTIALS          // partial tiles as bf16 (512-byte rows): half the round trip through the workspace (-0.26 ms per step);
            // each is a sum over >= 512 tokens rounded once (2^-9), like the bf16 weight gradient the reference's autocast GEMM returns
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2p;
            __builtin_nontemporal_store((u32x2p){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}, (u32x2p*)((bf16_t*)part + row * 256 + lane * 4));
#else
            __builtin_nontemporal_store(v, (f32x4*)(part + row * 256 + lane * 4));
#endif
        }
    }
}

// ---- the same tile on a FREE-RUNNING schedule (round 4) ---------------------------------------------------------------------
// gemm_nt256f.hip's schedule (read its header) carried over: all eight waves run ONE stream with two barriers per stage, every
// wave reading the fragments of phase p + 1 between its own MFMAs of phase p (P in place of the fragment it has just used up,
// Q into the set the other two phases do not use), the four LDS-DMA pieces of a half-tile pair between the MFMAs behind each
// barrier, six phases ahead of their retiring vmcnt(8).  In the ping-pong kernel above the LOAD half of a barrier interval is
// the long pole here (twice the LDS instructions of gemm_nt: 48 transposing reads per wave and stage), which is exactly what
// this schedule takes off the critical path.
//     p1 (P0,Q0): 8 MFMA | vmcnt, BARRIER | read Q1 (8 tr) ; 8 MFMA + the 4 DMA of {P-h0, Q-h0}(t+2)
//     p2 (P0,Q1): 4 x [4 MFMA ; read P1[j] in place (4 tr)]
//     p3 (P1,Q1): 8 MFMA | vmcnt, BARRIER | 8 MFMA + the 4 DMA of {Q-h1, P-h1}(t+2)
//     p4 (P1,Q0): 4 x [4 MFMA ; read P0(t+1)[j] in place (4 tr) ; j < 2: read half of Q0(t+1) (4 tr)]
// LDS: P half-tiles [buf][h] in the first 64 KiB, Q in the second: six per-lane fragment addresses + 16-bit immediates.
// DMA sources: per-lane (token row, clamped column) terms fixed for the kernel (4 registers), stage / piece terms in the SGPR base.
// A stage index past the split's last stage is clamped (it re-reads the last stage into slots nobody reads again).
// Body as a device function: output tile (tnb, tkb), tokens t_begin .. t_begin + 64 * nt (nt >= 2 stages), partial tile -> `part`.
__device__ __forceinline__ void tn256f_body(const TnParams& p, float* const part, const int tnb, const int tkb, const int t_begin, const int nt,
                                            char* const smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wr = w >> 2, wc = w & 3;
    const int tilesK = (p.K + 255) / 256;

    const unsigned ldp2 = (unsigned)p.ldp * 2u, ldq2 = (unsigned)p.ldq * 2u;
    const char* baseP = (const char*)(p.P + (int64_t)t_begin * p.ldp);
    const char* baseQ = (const char*)(p.Q + (int64_t)t_begin * p.ldq);
    const int64_t stepP = 64 * (int64_t)ldp2, stepQ = 64 * (int64_t)ldq2;
    unsigned cP[2], cQ[2];                                    // piece j adds 32 token rows: in the SGPR base
    {
        const int r = w * 4 + (lane >> 4);
        const int col2 = ((lane & 15) ^ ((r & 7) << 1)) * 16;                   // byte offset of this lane's 8 columns in the half-tile row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            cP[h] = (unsigned)r * ldp2 + (unsigned)min((tnb * 256 + h * 128) * 2 + col2, (p.N - 8) * 2);
            cQ[h] = (unsigned)r * ldq2 + (unsigned)min((tkb * 256 + h * 128) * 2 + col2, (p.K - 8) * 2);
        }
    }
    const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)smem) + (unsigned)w * 1024u);
    auto stP = [&](int buf, int h, int kt, int j) {
        const int kc = min(kt, nt - 1);
        glds16_s(cP[h], baseP + (int64_t)kc * stepP + (int64_t)j * 32 * ldp2, ldsw + (unsigned)((buf * 2 + h) * HT_BYTES + j * 8192));
    };
    auto stQ = [&](int buf, int h, int kt, int j) {
        const int kc = min(kt, nt - 1);
        glds16_s(cQ[h], baseQ + (int64_t)kc * stepQ + (int64_t)j * 32 * ldq2, ldsw + (unsigned)(65536 + (buf * 2 + h) * HT_BYTES + j * 8192));
    };

    f32x4 acc[4][8];                                          // [k-frag][n-frag]: D rows = n (4g+r), D cols = k (l15)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // transposing reads: lane supplies token row 4g + (l15>>2) [+16], columns c0 + (l15&3)*4 .. +3 (gemm.hip gemm_tn)
    const KZV_LDS char *aP[4], *aQ[2];
    {
        const int trow = 4 * g + (l15 >> 2);
        const int tsw = (trow & 7) << 1;
        const int sub8 = (l15 & 1) * 8, cq = (l15 >> 1) & 1;
        const KZV_LDS char* sm = (const KZV_LDS char*)smem;
#pragma unroll
        for (int j = 0; j < 4; ++j) aP[j] = sm + trow * 256 + ((((wr * 8 + j * 2) ^ tsw) | cq) * 16) + sub8;
#pragma unroll
        for (int i = 0; i < 2; ++i) aQ[i] = sm + 65536 + trow * 256 + ((((wc * 4 + i * 2) ^ tsw) | cq) * 16) + sub8;
        asm volatile("" : "+v"(aP[0]), "+v"(aP[1]), "+v"(aP[2]), "+v"(aP[3]), "+v"(aQ[0]), "+v"(aQ[1]));
    }
    bf16x8 fp[4][2], fqX[2][2], fqY[2][2];                    // [frag][token half ks]
    auto tr = [&](const KZV_LDS char* q) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((KZV_LDS bf16x4*)q); };
    auto rdP = [&](int buf, int mh, int j) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int o = (buf * 2 + mh) * HT_BYTES + ks * 8192;
            const bf16x4 lo = tr(aP[j] + o), hi = tr(aP[j] + o + 4096);
            fp[j][ks] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    auto rdQ = [&](int buf, int nh, int i, bf16x8 (&fq)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int o = (buf * 2 + nh) * HT_BYTES + ks * 8192;
            const bf16x4 lo = tr(aQ[i] + o), hi = tr(aQ[i] + o + 4096);
            fq[i][ks] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };
    // two MFMAs: P fragment j, token half ks, against both Q fragments of half nh
    auto mm2 = [&](int mh, int nh, int j, int ks, const bf16x8 (&fq)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            acc[nh * 2 + i][mh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp[j][ks], fq[i][ks], acc[nh * 2 + i][mh * 4 + j], 0, 0, 0);
    };
    auto mm4 = [&](int mh, int nh, int j, const bf16x8 (&fq)[2][2]) { mm2(mh, nh, j, 0, fq); mm2(mh, nh, j, 1, fq); };

    float bsum[8];                                            // bias gradient: see gemm_tn256_kernel
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
    const bool bias_wave = p.dbias != nullptr && wc == 0;
    auto bias_add = [&](int mh) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 u = __builtin_bit_cast(u32x4, fp[j][ks]);
                float sacc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) sacc += __uint_as_float(u[q] << 16) + __uint_as_float(u[q] & 0xffff0000u);
                bsum[mh * 4 + j] += sacc;
            }
    };
#define KZV_TSB() __builtin_amdgcn_sched_barrier(0)
    // One stage.  Qf: the set holding Q0(t) (p1, p4); Qs: the set Q1(t) is read into (p2, p3) and, in p4, Q0(t+1).
    auto ktile = [&](auto bufc, bf16x8 (&Qf)[2][2], bf16x8 (&Qs)[2][2], int t, bool last) {
        constexpr int BUF = decltype(bufc)::value;
        const bool wb = bias_wave && (t % tilesK) == tkb;
        // ---- p1 ----
        if (wb) bias_add(0);                                  // all of P0(t) is in fp here (p2 overwrites it fragment by fragment)
        mm4(0, 0, 0, Qf); mm4(0, 0, 1, Qf);
        KZV_TSB();
        vmcnt<8>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        KZV_TSB();
        rdQ(BUF, 1, 0, Qs); rdQ(BUF, 1, 1, Qs);
        KZV_TSB();
        mm2(0, 0, 2, 0, Qf); KZV_TSB(); stP(BUF, 0, t + 2, 0); KZV_TSB();
        mm2(0, 0, 2, 1, Qf); KZV_TSB(); stP(BUF, 0, t + 2, 1); KZV_TSB();
        mm2(0, 0, 3, 0, Qf); KZV_TSB(); stQ(BUF, 0, t + 2, 0); KZV_TSB();
        mm2(0, 0, 3, 1, Qf); KZV_TSB(); stQ(BUF, 0, t + 2, 1); KZV_TSB();
        // ---- p2 ----
#pragma unroll
        for (int j = 0; j < 4; ++j) { mm4(0, 1, j, Qs); KZV_TSB(); rdP(BUF, 1, j); KZV_TSB(); }
        // ---- p3 ----
        mm4(1, 1, 0, Qs); mm4(1, 1, 1, Qs);
        KZV_TSB();
        vmcnt<8>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        KZV_TSB();
        if (wb) bias_add(1);                                  // all of P1(t) is in fp (read in p2, waited for above)
        mm2(1, 1, 2, 0, Qs); KZV_TSB(); stQ(BUF, 1, t + 2, 0); KZV_TSB();
        mm2(1, 1, 2, 1, Qs); KZV_TSB(); stQ(BUF, 1, t + 2, 1); KZV_TSB();
        mm2(1, 1, 3, 0, Qs); KZV_TSB(); stP(BUF, 1, t + 2, 0); KZV_TSB();
        mm2(1, 1, 3, 1, Qs); KZV_TSB(); stP(BUF, 1, t + 2, 1); KZV_TSB();
        // ---- p4 ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mm4(1, 0, j, Qf);
            KZV_TSB();
            if (!last) { rdP(BUF ^ 1, 0, j); if (j < 2) rdQ(BUF ^ 1, 0, j, Qs); }
            KZV_TSB();
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

#pragma unroll
    for (int b = 0; b < 2; ++b) {                  // stages 0 and 1 (nt >= 2)
        stP(b, 0, b, 0); stP(b, 0, b, 1); stQ(b, 0, b, 0); stQ(b, 0, b, 1);
        stQ(b, 1, b, 0); stQ(b, 1, b, 1); stP(b, 1, b, 0); stP(b, 1, b, 1);
    }
    vmcnt<12>();                                   // P-h0(0), Q-h0(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) rdP(0, 0, j);
    rdQ(0, 0, 0, fqX); rdQ(0, 0, 1, fqX);
    for (int t = 0; t < nt; t += 2) {
        ktile(I0{}, fqX, fqY, t, t + 1 >= nt);
        if (t + 1 < nt) ktile(I1{}, fqY, fqX, t + 1, t + 2 >= nt);
    }
    vmcnt<0>();                                    // the refills issued past the last stage land before the epilogue reuses the ring
#undef KZV_TSB

    if (bias_wave) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = bsum[j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int n = tnb * 256 + (j >> 2) * 128 + wr * 64 + (j & 3) * 16 + l15;
            if (g == 0 && n < p.n_store) atomicAdd(p.dbias + n, v);
        }
    }

    // ---- epilogue: as gemm_tn256_kernel ----
    float* tile = (float*)smem;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ln = wr * 64 + j * 16 + 4 * g + r;
                    const int k = (i >> 1) * 128 + wc * 32 + (i & 1) * 16 + l15;
                    tile[ln * 256 + ((((k >> 2) ^ (ln & 31)) << 2) | (k & 3))] = acc[i][c * 4 + j][r];
                }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int ln = w * 16 + it;
            const f32x4 v = *(const f32x4*)(tile + ln * 256 + ((lane ^ (ln & 31)) << 2));
            const int row = c * 128 + ln;
#ifndef KZV_TN_F32_PARTIALS
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2p;
            __builtin_nontemporal_store((u32x2p){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])}, (u32x2p*)((bf16_t*)part + row * 256 + lane * 4));
#else
            __builtin_nontemporal_store(v, (f32x4*)(part + row * 256 + lane * 4));
#endif
        }
    }
}

__global__ __launch_bounds__(512) void gemm_tn256f_kernel(const TnParams p, float* part_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tilesN = (p.N + 255) / 256, tilesK = (p.K + 255) / 256;
    const int per = tilesN * tilesK;
    const int id = xcd_remap(blockIdx.x, per * p.splits);
    const int split = id / per, rem = id - split * per;
    const int t_begin = split * p.chunk;                      // chunk and Mtok are multiples of 64 (launcher)
    const int t_end = min(p.Mtok, t_begin + p.chunk);
    tn256f_body(p, part_ws + (size_t)id * PART_FLOATS, rem / tilesK, rem % tilesK, t_begin, (t_end - t_begin) >> 6, smem);      // >= 2 stages (launcher)
}

// ---- one launch for a layer's input-gradient GEMM and its weight-gradient GEMM (round 4) ------------------
// Both read the same dY and neither reads the other's output.  As two launches each ends on a partly filled round: the persistent
// gemm_nt gives its 256 workgroups 5 or 6 (1 or 2, 7 or 8) tiles each and the workgroups with the smaller count idle for a tile
// time (~6 % of the launch), the weight gradient runs on 243 - 252 of the 256 CUs, and a kernel boundary sits between them.  Here
// every workgroup runs its gemm_nt tiles (nt256p_body) and then ONE token range of ONE weight-gradient tile (tn256f_body) whose
// length the host sizes so that all workgroups finish together: a workgroup that had one tile less gets `r` stages more (r = the
// tile's duration in stages).  The plan (tile, split, first stage, stage count per workgroup) is a kernel argument.
// The workgroups that share a weight-gradient tile ROW and a split index take the same token range (the bias gradient is
// partitioned over them by stage index); their gemm_nt tile counts are equal except at one boundary (the ids are contiguous).
struct PairPlan { unsigned short tile[256], split[256], t0[256], cnt[256]; };      // by blockIdx; cnt == 0: no weight-gradient share

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pair_kernel(const NtParams pn, const int tiles, const int tilesN, const int strip_in, const TnParams pt,
                                                        float* part_ws, const int per, const PairPlan plan) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    kzv_pair_nt::nt256p_body<EPI, false>(pn, tiles, tilesN, strip_in, (int)blockIdx.x, (int)gridDim.x, smem);
    const int cnt = plan.cnt[blockIdx.x];
    if (cnt == 0) return;
    __syncthreads();                               // every wave is done with the gemm_nt ring and drain patches
    const int tile = plan.tile[blockIdx.x], split = plan.split[blockIdx.x];
    const int tilesK = (pt.K + 255) / 256;
    tn256f_body(pt, part_ws + ((size_t)split * per + tile) * PART_FLOATS, tile / tilesK, tile % tilesK, (int)plan.t0[blockIdx.x] * 64, cnt, smem);
}

// OUT[n][k] += sum over splits of the partial tiles (4 columns per thread, one 1-KiB row segment per wave-instruction)
__global__ __launch_bounds__(256) void gemm_tn256_fold_kernel(const float* part_ws, float* OUT, int64_t ldo, int n_store, int K,
                                                             int tilesK, int per, int splits) {
    const int tile = blockIdx.x >> 6;                                 // 64 workgroups per tile: 4 rows each
    const int row = (blockIdx.x & 63) * 4 + (threadIdx.x >> 6), k4 = (threadIdx.x & 63) * 4;
    const int tnb = tile / tilesK, tkb = tile - tnb * tilesK;
    const int gn = tnb * 256 + row, gk = tkb * 256 + k4;
    if (gn >= n_store || gk >= K) return;                             // K % 4 == 0 (launcher): a 4-column group is all in or all out
    const float* src = part_ws + ((size_t)tile * 256 + row) * 256 + k4;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef KZV_TN_F32_PARTIALS
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2p;
#pragma unroll 4
    for (int sp = 0; sp < splits; ++sp) {
        const u32x2p u = __builtin_nontemporal_load((const u32x2p*)((const bf16_t*)(part_ws + ((size_t)sp * per + tile) * PART_FLOATS) + row * 256 + k4));
        s[0] += bf2f((bf16_t)(u[0] & 0xffffu)); s[1] += bf2f((bf16_t)(u[0] >> 16)); s[2] += bf2f((bf16_t)(u[1] & 0xffffu)); s[3] += bf2f((bf16_t)(u[1] >> 16));
    }
#else
#pragma unroll 4
    for (int sp = 0; sp < splits; ++sp) s += __builtin_nontemporal_load((const f32x4*)(src + (size_t)sp * per * PART_FLOATS));
#endif
    f32x4* o = (f32x4*)(OUT + (int64_t)gn * ldo + gk);
    *o += s;
}

// Invariant of the deferred folds (KzvTnFoldScope, and KzvLnDeferScope in layernorm.hip): the pending lists are process-global and
// unsynchronised -- ONE host thread issues the launches of a scope, on ONE stream (model.cpp's backward); a scope held open defers
// every kzv_gemm_tn of the process.  Outputs inside a scope must be distinct (a repeated one is folded first, below).
// Partial-tile workspace: TN_REGIONS regions of the largest size asked for so far (grow-only; calls are stream-ordered).  Region 0
// serves a launch whose fold follows at once; regions 1.. the launches of a KzvTnFoldScope (the four weight gradients of an encoder
// layer), whose folds are ONE launch when the scope closes: a fold is ~5 us of launch latency + ~5 us of data, 49 times per step.
constexpr int TN_REGIONS = 6;
float* g_tn_buf = nullptr;
size_t g_tn_region = 0;                  // floats per region
float* tn_partials(size_t floats, int region) {
    if (floats > g_tn_region) {
        if (g_tn_buf) { (void)hipDeviceSynchronize(); (void)hipFree(g_tn_buf); g_tn_buf = nullptr; g_tn_region = 0; }
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)TN_REGIONS * floats * sizeof(float)) != hipSuccess) return nullptr;
        g_tn_buf = (float*)q; g_tn_region = floats;
    }
    return g_tn_buf + (size_t)region * g_tn_region;
}

struct TnFold { const float* ws; float* OUT; int64_t ldo; int n_store, K, tilesK, per, splits, block0; };
struct TnFoldTable { TnFold e[TN_REGIONS - 1]; int n; };
// the folds of a scope in one launch: a block finds its entry by its index (entries hold 64 blocks per output tile)
__global__ __launch_bounds__(256) void gemm_tn256_fold_multi_kernel(const TnFoldTable t) {
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.e[k + 1].block0) ++k;
    const TnFold& e = t.e[k];
    const int bid = blockIdx.x - e.block0;
    const int tile = bid >> 6;
    const int row = (bid & 63) * 4 + (threadIdx.x >> 6), k4 = (threadIdx.x & 63) * 4;
    const int tnb = tile / e.tilesK, tkb = tile - tnb * e.tilesK;
    const int gn = tnb * 256 + row, gk = tkb * 256 + k4;
    if (gn >= e.n_store || gk >= e.K) return;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef KZV_TN_F32_PARTIALS
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2p;
#pragma unroll 4
    for (int sp = 0; sp < e.splits; ++sp) {
        const u32x2p u = __builtin_nontemporal_load((const u32x2p*)((const bf16_t*)(e.ws + ((size_t)sp * e.per + tile) * PART_FLOATS) + row * 256 + k4));
        s[0] += bf2f((bf16_t)(u[0] & 0xffffu)); s[1] += bf2f((bf16_t)(u[0] >> 16)); s[2] += bf2f((bf16_t)(u[1] & 0xffffu)); s[3] += bf2f((bf16_t)(u[1] >> 16));
    }
#else
    const float* src = e.ws + ((size_t)tile * 256 + row) * 256 + k4;
#pragma unroll 4
    for (int sp = 0; sp < e.splits; ++sp) s += __builtin_nontemporal_load((const f32x4*)(src + (size_t)sp * e.per * PART_FLOATS));
#endif
    f32x4* o = (f32x4*)(e.OUT + (int64_t)gn * e.ldo + gk);
    *o += s;
}
int g_tn_defer = 0;
std::vector<TnFold> g_tn_pending;
std::vector<size_t> g_tn_pending_floats;
int tn_flush(hipStream_t s) {
    if (g_tn_pending.empty()) return KZV_OK;
    TnFoldTable t;
    int blocks = 0;
    for (size_t i = 0; i < g_tn_pending.size(); ++i) { t.e[i] = g_tn_pending[i]; t.e[i].block0 = blocks; blocks += g_tn_pending[i].per * 64; }
    t.n = (int)g_tn_pending.size();
    hipLaunchKernelGGL(gemm_tn256_fold_multi_kernel, dim3(blocks), dim3(256), 0, s, t);
    g_tn_pending.clear();
    return kzv_check_launch("gemm_tn256_fold");
}

// stage schedule: 0 = eight-phase ping-pong (gemm_tn256_kernel), 1 = free-running (gemm_tn256f_kernel); kzv_set_tn_schedule / KZV_TN_FREE
int g_tn_schedule = -1;
int tn_schedule() {
    if (g_tn_schedule < 0) { const char* e = getenv("KZV_TN_FREE"); g_tn_schedule = e ? (atoi(e) != 0) : 1; }      // default: free-running (+10..13 % per launch, same-box A/B, bit-identical partial tiles)
    return g_tn_schedule;
}
int tn256_min_tiles() {
    static int v = -1;
    // 9: the 768 x 768 outputs (attention output projection, patch embedding) take this kernel too -- 28 token splits each,
    // 83 -> ~70 us against the 128 x 128 kernel (-0.13 ms per step, same-box A/B; round 2's threshold was 24)
    if (v < 0) { const char* e = getenv("KZV_TN256_MIN_TILES"); v = e ? atoi(e) : 9; }
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

int kzv_tn256_launch(const TnParams& p0, hipStream_t s) {
    const int tiles = ((p0.N + 255) / 256) * ((p0.K + 255) / 256);
    const int tok_tiles = p0.Mtok / 64;
    if (tiles < tn256_min_tiles() || p0.Mtok % 64 || tok_tiles < 2 || p0.N < 8 || p0.K < 8) return 0;
    if ((uint64_t)64 * (uint64_t)p0.ldp * 2 + (uint64_t)p0.N * 2 > 0xffffffffull || (uint64_t)64 * (uint64_t)p0.ldq * 2 + (uint64_t)p0.K * 2 > 0xffffffffull) return 0;
    TnParams p = p0;
    // one workgroup per CU in total; every split keeps >= 8 reduction stages (and at least 2: the prologue stages two)
    int cus = device_cus() - kzv_cu_reserve();
    { const char* e = getenv("KZV_TN_CUS"); const int g = e ? atoi(e) : 0; if (g > 0 && g < cus) cus = g; }   // dev: two-stream experiment
    int splits = cus / tiles;      // leave the reserved CUs to a concurrent collective
    if (splits > tok_tiles / 8) splits = tok_tiles / 8;
    if (splits < 1) splits = 1;
    int chunk_tiles = (tok_tiles + splits - 1) / splits;
    if (chunk_tiles < 2) chunk_tiles = 2;
    splits = (tok_tiles + chunk_tiles - 1) / chunk_tiles;
    if (tok_tiles - (splits - 1) * chunk_tiles < 2) return 0;         // a one-stage last split: leave it to the 128x128 kernel
    p.splits = splits; p.chunk = chunk_tiles * 64;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_done = true; }
    if (p.K % 4 || p.ldo % 4 || ((uintptr_t)p.OUT & 15)) return 0;
    const size_t need = (size_t)tiles * splits * PART_FLOATS;
    const bool deferred = g_tn_defer > 0;
    bool same_out = false;                       // a second gradient into the SAME output inside one scope: the single fold launch adds every
    for (const TnFold& e : g_tn_pending) same_out |= e.OUT == p.OUT;      // entry with a plain read-modify-write, so the two would race
    if (deferred && ((int)g_tn_pending.size() == TN_REGIONS - 1 || need > g_tn_region || same_out)) {
        if (tn_flush(s) != KZV_OK) return 0;     // no free region, the workspace is about to be re-allocated, or a repeated output: fold what is pending first
    }
    float* ws = tn_partials(need, deferred ? 1 + (int)g_tn_pending.size() : 0);
    if (!ws) return 0;
    if (tn_schedule()) {
        static bool attr_f = false;
        if (!attr_f) { (void)hipFuncSetAttribute((const void*)gemm_tn256f_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); attr_f = true; }
        hipLaunchKernelGGL(gemm_tn256f_kernel, dim3(tiles * splits), dim3(512), LDS_BYTES, s, p, ws);
    } else
        hipLaunchKernelGGL(gemm_tn256_kernel, dim3(tiles * splits), dim3(512), LDS_BYTES, s, p, ws);
    if (deferred) {
        g_tn_pending.push_back(TnFold{ws, p.OUT, p.ldo, p.n_store, p.K, (p.K + 255) / 256, tiles, splits, 0});
        return 1;
    }
    hipLaunchKernelGGL(gemm_tn256_fold_kernel, dim3(tiles * 64), dim3(256), 0, s, ws, p.OUT, p.ldo, p.n_store, p.K,
                       (p.K + 255) / 256, tiles, splits);
    return 1;
}

// gemm.hip: the parameter block of a kzv_gemm_nt call (validated there)
int kzv_nt_params(const kzv_gemm_nt_args* a, int epilogue, NtParams* out);

namespace {
int g_pair = -1;
int pair_enabled() {
    // default OFF: measured on the step's four encoder pairs (tools/dev/r4_pair.py, same-box A/B): -23 us per layer at best (one kernel
    // boundary per pair), +0.4 % img/s on the whole step -- and NO gain from sizing the token ranges to the gemm_nt tile counts (the
    // workgroups that leave the gemm_nt phase early then run their weight-gradient stages beside the others' gemm_nt tiles, and the
    // two loops slow each other as they do on two streams).  DESIGN.md section 8.
    if (g_pair < 0) { const char* e = getenv("KZV_PAIR"); g_pair = e ? (atoi(e) != 0) : 0; }
    return g_pair;
}
// stages of the weight-gradient loop one gemm_nt tile is worth: (K-tiles x 1.45 us + drain) / 1.55 us per stage; KZV_PAIR_R overrides (x 0.01)
double pair_ratio(int nk, int epilogue) {
    static int ov = -2;
    if (ov == -2) { const char* e = getenv("KZV_PAIR_R"); ov = e ? atoi(e) : -1; }
    const double drain = (epilogue == KZV_EPI_DGELU || epilogue == KZV_EPI_RESID) ? 5.0 : 2.5;
    const double r = 0.6 * (nk * 1.45 + drain) / 1.55;          // 0.6: the best of a sweep over 0 / 0.6 / 1 / 1.4 / 2 (flat between 0 and 0.6)
    return ov >= 0 ? r * ov * 0.01 / 0.6 : r;
}
}  // namespace

// 1 = launched as one kernel; 0 = not taken (the caller issues the two GEMMs separately); < 0 = error
int kzv_gemm_pair_launch(const kzv_gemm_nt_args* na, int epilogue, const kzv_gemm_tn_args* ta, hipStream_t s) {
    if (!pair_enabled() || kzv_cu_reserve() != 0 || !tn_schedule() || device_cus() != 256) return 0;
    if (epilogue != KZV_EPI_BF16 && epilogue != KZV_EPI_F32 && epilogue != KZV_EPI_DGELU && epilogue != KZV_EPI_RESID) return 0;
    NtParams pn;
    if (kzv_nt_params(na, epilogue, &pn) != KZV_OK) return 0;
    // the conditions of kzv_nt256p_launch ...
    const int tilesN = (pn.N + 255) / 256, tiles = ((pn.M + 255) / 256) * tilesN;
    if (pn.K < 128 || pn.K % 128 || tiles < 256) return 0;
    if ((uint64_t)256 * (uint64_t)pn.lda * 2 > 0xffffffffull || (uint64_t)pn.n_valid * (uint64_t)pn.ldb * 2 > 0xffffffffull) return 0;
    // ... and of kzv_tn256_launch
    TnParams pt{};
    pt.P = (const bf16_t*)ta->P; pt.Q = (const bf16_t*)ta->Q; pt.OUT = ta->OUT; pt.zero16 = kzv_zero_page();
    pt.ldp = ta->ldp; pt.ldq = ta->ldq; pt.ldo = ta->ldo; pt.Mtok = ta->Mtok; pt.N = ta->N; pt.K = ta->K;
    pt.n_store = ta->n_store > 0 ? ta->n_store : ta->N; pt.dbias = ta->dbias;
    if (!pt.P || !pt.Q || !pt.OUT || !pt.zero16) return 0;
    const int tilesKw = (pt.K + 255) / 256, tilesNw = (pt.N + 255) / 256, per = tilesNw * tilesKw;
    const int stages = pt.Mtok / 64;
    if (per < tn256_min_tiles() || per > 128 || pt.Mtok % 64 || pt.N < 8 || pt.K < 8 || pt.N % 8 || pt.K % 8) return 0;
    if ((uint64_t)64 * (uint64_t)pt.ldp * 2 + (uint64_t)pt.N * 2 > 0xffffffffull || (uint64_t)64 * (uint64_t)pt.ldq * 2 + (uint64_t)pt.K * 2 > 0xffffffffull) return 0;
    if (pt.K % 4 || pt.ldo % 4 || ((uintptr_t)pt.OUT & 15) || ((uintptr_t)pt.P & 15) || ((uintptr_t)pt.Q & 15) || pt.ldp % 8 || pt.ldq % 8) return 0;
    const int G = 256, S = G / per;
    if (S < 1 || stages < 8 * S || stages > 65000) return 0;
    // ---- the plan ----
    PairPlan plan;
    int cb[256], idb[256];
    const int nwg = per * S;
    for (int b = 0; b < G; ++b) {
        const int vblk = (b & 7) * (G >> 3) + (b >> 3);
        cb[b] = vblk < tiles ? (tiles - 1 - vblk) / G + 1 : 0;
        idb[b] = -1;
        plan.tile[b] = plan.split[b] = plan.t0[b] = plan.cnt[b] = 0;
        if (b < nwg) {
            const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;                      // kzv_common.h xcd_remap
            idb[b] = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
        }
    }
    const double r = pair_ratio(pn.K / 64, epilogue);
    std::vector<int> cg((size_t)tilesNw * S, 0);                                    // gemm_nt tiles of the group (tile row, split): the largest of its members
    for (int b = 0; b < nwg; ++b) { const int sp = idb[b] / per, tl = idb[b] % per; int& c = cg[(size_t)(tl / tilesKw) * S + sp]; c = cb[b] > c ? cb[b] : c; }
    std::vector<int> t0((size_t)tilesNw * S), cnt((size_t)tilesNw * S);
    for (int tn = 0; tn < tilesNw; ++tn) {
        double csum = 0;
        for (int sp = 0; sp < S; ++sp) csum += cg[(size_t)tn * S + sp];
        const double level = (stages + r * csum) / S;                               // the common finishing time, in stages
        double acc = 0; int prev = 0;
        for (int sp = 0; sp < S; ++sp) {
            double n = level - r * cg[(size_t)tn * S + sp];
            if (n < 2) n = 2;
            acc += n;
            int end = sp == S - 1 ? stages : (int)(acc + 0.5);
            const int left = S - 1 - sp;                                             // every later split keeps >= 2 stages
            if (end > stages - 2 * left) end = stages - 2 * left;
            if (end < prev + 2) end = prev + 2;
            t0[(size_t)tn * S + sp] = prev; cnt[(size_t)tn * S + sp] = end - prev; prev = end;
        }
        if (prev != stages) return 0;
    }
    for (int b = 0; b < nwg; ++b) {
        const int sp = idb[b] / per, tl = idb[b] % per; const size_t gi = (size_t)(tl / tilesKw) * S + sp;
        plan.tile[b] = (unsigned short)tl; plan.split[b] = (unsigned short)sp; plan.t0[b] = (unsigned short)t0[gi]; plan.cnt[b] = (unsigned short)cnt[gi];
    }
    pt.splits = S; pt.chunk = 0;
    // ---- workspace / fold bookkeeping of kzv_tn256_launch ----
    const size_t need = (size_t)per * S * PART_FLOATS;
    const bool deferred = g_tn_defer > 0;
    bool same_out = false;
    for (const TnFold& e : g_tn_pending) same_out |= e.OUT == pt.OUT;
    if (deferred && ((int)g_tn_pending.size() == TN_REGIONS - 1 || need > g_tn_region || same_out)) {
        if (tn_flush(s) != KZV_OK) return 0;
    }
    float* ws = tn_partials(need, deferred ? 1 + (int)g_tn_pending.size() : 0);
    if (!ws) return 0;
    KzvProfScope prof(5, 2.0 * pn.M * pn.n_valid * pn.K + 2.0 * pt.Mtok * pt.N * pt.K, s);
    constexpr int PAIR_LDS = 160 * 1024;
#define KZV_PAIR_CASE(E)                                                                                            \
    case E: {                                                                                                       \
        static bool attr_done = false;                                                                              \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)gemm_pair_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS); attr_done = true; } \
        hipLaunchKernelGGL((gemm_pair_kernel<E>), dim3(G), dim3(512), PAIR_LDS, s, pn, tiles, tilesN, kzv_nt_strip(), pt, ws, per, plan);   \
    } break;
    switch (epilogue) {
        KZV_PAIR_CASE(KZV_EPI_BF16) KZV_PAIR_CASE(KZV_EPI_F32) KZV_PAIR_CASE(KZV_EPI_DGELU) KZV_PAIR_CASE(KZV_EPI_RESID)
        default: return 0;
    }
#undef KZV_PAIR_CASE
    if (deferred) g_tn_pending.push_back(TnFold{ws, pt.OUT, pt.ldo, pt.n_store, pt.K, tilesKw, per, S, 0});
    else hipLaunchKernelGGL(gemm_tn256_fold_kernel, dim3(per * 64), dim3(256), 0, s, ws, pt.OUT, pt.ldo, pt.n_store, pt.K, tilesKw, per, S);
    return kzv_check_launch("gemm_pair") == KZV_OK ? 1 : -1;
}

extern "C" int kzv_set_pair(int n) { g_pair = n < 0 ? -1 : (n != 0); return KZV_OK; }
extern "C" int kzv_gemm_dgrad_wgrad(const kzv_gemm_nt_args* na, int epilogue, const kzv_gemm_tn_args* ta, void* stream) {
    if (!na || !ta) return kzv_fail(KZV_E_ARG, "gemm_dgrad_wgrad: null");
    const int rc = kzv_gemm_pair_launch(na, epilogue, ta, (hipStream_t)stream);
    if (rc < 0) return kzv_fail(KZV_E_HIP, "gemm_dgrad_wgrad: pair launch");
    if (rc == 1) return KZV_OK;
    const int r2 = kzv_gemm_tn(ta, stream);
    return r2 != KZV_OK ? r2 : kzv_gemm_nt(na, epilogue, stream);
}
extern "C" int kzv_set_tn_schedule(int n) { g_tn_schedule = n < 0 ? -1 : (n != 0); return KZV_OK; }

KzvTnFoldScope::KzvTnFoldScope(hipStream_t stream) : s(stream) { ++g_tn_defer; }
KzvTnFoldScope::~KzvTnFoldScope() { if (--g_tn_defer == 0) (void)tn_flush(s); }
