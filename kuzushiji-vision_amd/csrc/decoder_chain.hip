// The linear chains of a RoBERTa decoder layer in the TRAINING forward (HF modeling_roberta.py:421-464 under
// src/models/trocr_model.py:258-297), two launches per layer instead of nine:
//   A (between the two attentions):  s1 = drop(ctx Wo^T + b) + x;  x1 = LN1(s1);  cq = x1 Wcq^T + b
//   B (after the cross-attention):   s2 = drop(cctx Wco^T + b) + x1;  x2 = LN2(s2);  act = gelu(x2 Wfc1^T + b);
//                                    s3 = drop(act Wfc2^T + b) + x2;  x3 = LN3(s3);  [qkv of the NEXT layer = x3 Wqkv^T + b]
// At the decoder's size (hidden 256, FFN 768, ~15k rows) every one of these is a 15 - 21 us launch for 2 GFLOP: 240 tiles of
// 128 x 128 leave the chip under one wave per SIMD and the chain is launch- and latency-bound.  The chains are ROW-LOCAL, so one
// workgroup takes 64 rows through a whole chain: the activations stay in LDS (bf16 A operands, fp32 pre-LayerNorm sums), the
// weights arrive in MFMA fragment order (decode_fused.hip's pack: a wave-instruction = one contiguous KiB) as one stream per wave
// with a window of 16 fragments in flight across the GEMM boundaries, and every tensor the backward reads is written on the way
// (same tensors, same rounding points as the launch-per-operation path: bf16 GEMM operands, fp32 sums / statistics / residuals;
// the dropout bits of the two residual epilogues are the engine's own -- key, element index m * 256 + n -- so the backward's
// regenerated masks match).  Each weight fragment feeds four MFMAs (the four 16-row tiles).
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "kzv_kernels.h"

namespace {

constexpr int HD = 256, FD = 768, RM = 64, RT = RM / 16;
constexpr int LDH = HD + 8;          // bf16 rows, 256 wide
constexpr int LDW = FD + 8;          // bf16 rows, 768 wide
constexpr int LDS_ = HD + 4;         // fp32 rows of a pre-LayerNorm sum
constexpr int WIN = 16;
constexpr int LDS_A1 = RM * LDH * 2;                                   // bytes of the 256-wide operand tile
constexpr int LDS_A = LDS_A1 + RM * LDS_ * 4, LDS_B = LDS_A1 + RM * LDW * 2;

__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// LDS hand-offs only: global loads stay in flight (see decode_fused.hip)
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CB, int KS>
__device__ __forceinline__ const char* wave_frags(const bf16_t* Wp, int w) { return (const char*)(Wp + (int64_t)w * KS * 512); }
template <int CB, int KS>
__device__ __forceinline__ bf16x8 ld_frag(const char* wb, unsigned wo, int i) {       // fragment i = (k-step i / CB, column block w + 8 (i % CB))
    const int ks = i / CB, c = i % CB;
    return *(const bf16x8*)(wb + ((int64_t)(8 * c) * KS + ks) * 1024 + wo);
}
template <int CB, int KS>
__device__ __forceinline__ void fill_window(bf16x8 (&R)[WIN], const char* wb, int lane) {
    const unsigned wo = (unsigned)lane * 16u;
#pragma unroll
    for (int i = 0; i < WIN; ++i) R[i] = ld_frag<CB, KS>(wb, wo, i);
}
// acc[c][rt][r] = sum_k W[n][k] * a[m][k],  m = 16 rt + (lane & 15),  n = (w + 8 c) * 16 + 4 (lane >> 4) + r
// NEXT (compile time): refill the window from `next` behind the last fragments -- no run-time branch around a load (see pin)
template <int CB, int KS, int NCB, int NKS, bool NEXT>
__device__ __forceinline__ void chain_gemm(bf16x8 (&R)[WIN], const char* wb, const char* next, const bf16_t* a_lds, int lda, int lane, f32x4 (&acc)[CB][RT]) {
    constexpr int F = CB * KS;
    static_assert(F % WIN == 0 && NCB * NKS >= WIN, "chain_gemm: window");
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned wo = (unsigned)lane * 16u;
    const bf16_t* ap = a_lds + l15 * lda + g * 8;
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[c][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[RT];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        const int ks = i / CB, c = i % CB;
        if (c == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) fa[rt] = *(const bf16x8*)(ap + rt * 16 * lda + ks * 32);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R[i % WIN], fa[rt], acc[c][rt], 0, 0, 0);
        if (i + WIN < F) R[i % WIN] = ld_frag<CB, KS>(wb, wo, i + WIN);
        else if constexpr (NEXT) R[i % WIN] = ld_frag<NCB, NKS>(next, wo, i + WIN - F);
    }
}

#ifdef KZV_STAMPS      // diagnostic build (tools/dev/stamps_chain.py): the phase timeline of workgroup 0 of the last dec_chain_b launch
__device__ long long g_chain_stamps[32];
#define CH_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_chain_stamps[k] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#ifndef KZV_STAMP_SEG_KS1
#define KZV_STAMP_SEG_KS1 8
#define KZV_STAMP_SEG_NP2 1
#endif
#define SEG_STAMP(k) do { if (KS1 == KZV_STAMP_SEG_KS1 && NP2 == KZV_STAMP_SEG_NP2) CH_STAMP(k); } while (0)
#else
#define CH_STAMP(k)
#define SEG_STAMP(k)
#endif

struct Drop { unsigned thr16; float inv_keep; unsigned key; };
// The residual operand of a "dropout(linear) + residual" epilogue: a stored fp32 tensor x, or -- when x is null -- the LayerNorm output
// it would hold, recomputed from that LayerNorm's INPUT sum s, its row statistics st = (mean, rstd) and its weights (the arithmetic of
// ln_rows / ln_fwd_kernel, so the same bits): the fp32 LayerNorm outputs of a decoder layer are read by nothing but the next residual,
// so the chains do not write them (3 x 15.7 MB per layer at 15k rows).
struct Resid { const float* x; const float* s; const float* st; const float* g; const float* b; };

// 64 rows x 256 bf16 columns, global -> LDS (rows past M read as zeros)
__device__ __forceinline__ void load_rows(const bf16_t* __restrict__ src, bf16_t* dst, int m0, int M, int tid) {
#pragma unroll
    for (int q = 0; q < RM * 32 / 512; ++q) {
        const int idx = tid + q * 512, row = idx >> 5, ch = idx & 31;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m0 + row < M) v = *(const uint4*)(src + (int64_t)(m0 + row) * HD + ch * 8);
        *(uint4*)(dst + row * LDH + ch * 8) = v;
    }
}

// "dropout(linear) + residual" epilogue of a 256-wide GEMM: s = drop(acc + bias) + resid -> the LDS sum tile.  (Nothing goes to
// global memory from the MFMA layout -- a lane holds 4 columns of 16 different rows there, i.e. 32- / 64-byte pieces of 16 rows per
// store instruction, which is what made the first version of these kernels 2.7x slower than its byte count: every tensor leaves
// through an LDS tile as whole rows.)
// The epilogue's global operands (residual rows in the MFMA layout: 64-byte pieces of 16 rows per instruction, plus the row statistics when
// the residual is recomputed) are requested BEFORE the GEMM whose result they meet -- behind it they were 6 - 8 k cycles of exposed latency
// per epilogue (tools/dev/stamps_chain.py).  Rows past M read the last row (never used).
// Registers loaded long before their use are retired BY HAND: `s_waitcnt vmcnt(0)` and then every such register through an empty asm,
// so that no use can be scheduled above the wait (volatile asms keep their order; plain arithmetic does not stay behind one).  hipcc's
// own count is not enough here: with loads inside uniform branches (the residual's statistics) between the load and its use, the wait it
// emitted let the use run first -- chain A read rstd = 0 (the register's initial value) in 2 - 5 % of its workgroups once the launch had
// more workgroups than CUs (slower loads), i.e. residual = beta for 16 rows (tools/dev/r5_det3.py; rounds 3 - 4 before this).
__device__ __forceinline__ void pin(float& f) { asm volatile("" : "+v"(f)); }
__device__ __forceinline__ void pin(unsigned& f) { asm volatile("" : "+v"(f)); }
__device__ __forceinline__ void pin(float4& v) { pin(v.x); pin(v.y); pin(v.z); pin(v.w); }
__device__ __forceinline__ void pin(float2& v) { pin(v.x); pin(v.y); }
__device__ __forceinline__ void pin(uint2& v) { pin(v.x); pin(v.y); }
__device__ __forceinline__ void retire_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
struct ResidRegs { float4 v[2][RT]; float2 st[RT]; };
__device__ __forceinline__ void resid_prefetch(const Resid& resid, ResidRegs& r, int m0, int M, int w, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    const float* src = resid.x ? resid.x : resid.s;
    const float* stp = resid.x ? resid.x : resid.st;         // no branch around a load: a stored residual reads two of its own floats here (unused)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t m = min(m0 + rt * 16 + l15, M - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) r.v[c][rt] = *(const float4*)(src + m * HD + (w + 8 * c) * 16 + 4 * g);
        r.st[rt] = *(const float2*)(stp + 2 * m);
    }
}
__device__ __forceinline__ void resid_epilogue(const f32x4 (&acc)[2][RT], const float* __restrict__ bias, const Resid& resid, const ResidRegs& pre_in, const Drop& d,
                                               float* s_lds, int m0, int M, int w, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
    ResidRegs pre = pre_in;                                   // requested one GEMM ago: retired by hand (see pin)
    retire_loads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { pin(pre.v[0][rt]); pin(pre.v[1][rt]); pin(pre.st[rt]); }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n0 = (w + 8 * c) * 16 + 4 * g;
        const float4 bb = *(const float4*)(bias + n0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = rt * 16 + l15, m = m0 + row;
            if (m < M) {
                float v[4] = {acc[c][rt][0] + bb.x, acc[c][rt][1] + bb.y, acc[c][rt][2] + bb.z, acc[c][rt][3] + bb.w};
                if (d.thr16) {
                    const unsigned e = (unsigned)m * (unsigned)HD + (unsigned)n0;
                    const unsigned b0 = drop_bits(d.key, e >> 1), b1 = drop_bits(d.key, (e >> 1) + 1);
                    v[0] *= drop_keep(b0, 0, d.thr16, d.inv_keep); v[1] *= drop_keep(b0, 1, d.thr16, d.inv_keep);
                    v[2] *= drop_keep(b1, 0, d.thr16, d.inv_keep); v[3] *= drop_keep(b1, 1, d.thr16, d.inv_keep);
                }
                float4 x4;
                if (resid.x) x4 = pre.v[c][rt];
                else {
                    const float4 sv = pre.v[c][rt], gm = *(const float4*)(resid.g + n0), bt = *(const float4*)(resid.b + n0);
                    const float mean = pre.st[rt].x, rstd = pre.st[rt].y;
                    const float a0 = sv.x - mean, a1 = sv.y - mean, a2 = sv.z - mean, a3 = sv.w - mean;
                    x4 = make_float4(a0 * rstd * gm.x + bt.x, a1 * rstd * gm.y + bt.y, a2 * rstd * gm.z + bt.z, a3 * rstd * gm.w + bt.w);
                }
                *(float4*)(s_lds + row * LDS_ + n0) = make_float4(v[0] + x4.x, v[1] + x4.y, v[2] + x4.z, v[3] + x4.w);
            }
        }
    }
}
// LayerNorm of the 64 sum rows (wave w: rows 8 w .. 8 w + 7; a row = one KiB per store), the arithmetic of ln_fwd_kernel: the sum s,
// x (fp32) and xh (bf16) to global, xh also to the LDS operand tile, (mean, rstd) to stats
__device__ __forceinline__ void ln_rows(const float* s_lds, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ s_out,
                                        float* __restrict__ x_out, bf16_t* __restrict__ xh_out, float* __restrict__ stats, bf16_t* a_lds, int m0, int M, int w, int lane) {
    constexpr int NR = RM / 8;
    const float4 gm = *(const float4*)(gamma + lane * 4), bt = *(const float4*)(beta + lane * 4);
    float4 v8[NR];              // every row out of LDS before the first LDS store below: hipcc cannot tell the sum tile from the operand tile and
#pragma unroll                  // would otherwise keep each row's LDS read behind the previous row's write
    for (int q = 0; q < NR; ++q) v8[q] = *(const float4*)(s_lds + (w * NR + q) * LDS_ + lane * 4);
    // The two wave reductions of a row are 12 dependent ds_bpermute round trips (~120 cycles each): row by row that was 1.8 k cycles per
    // row, 14 k per LayerNorm phase (tools/dev/stamps_chain.py).  The eight rows of a wave are independent, so each butterfly step is
    // taken for all of them together -- the same additions in the same order per row (wave_sum's), eight shuffles in flight at a time.
    float t8[NR], mean8[NR], rstd8[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) t8[q] = v8[q].x + v8[q].y + v8[q].z + v8[q].w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float u8[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) u8[q] = __shfl_xor(t8[q], o, 64);
#pragma unroll
        for (int q = 0; q < NR; ++q) t8[q] += u8[q];
    }
    float4 c8[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        mean8[q] = t8[q] / (float)HD;
        c8[q] = make_float4(v8[q].x - mean8[q], v8[q].y - mean8[q], v8[q].z - mean8[q], v8[q].w - mean8[q]);
        t8[q] = c8[q].x * c8[q].x + c8[q].y * c8[q].y + c8[q].z * c8[q].z + c8[q].w * c8[q].w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float u8[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) u8[q] = __shfl_xor(t8[q], o, 64);
#pragma unroll
        for (int q = 0; q < NR; ++q) t8[q] += u8[q];
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) rstd8[q] = rsqrtf(t8[q] / (float)HD + eps);
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        const int row = w * NR + q, m = m0 + row;
        if (m >= M) {                                 // wave-uniform
            *(uint2*)(a_lds + row * LDH + lane * 4) = make_uint2(0, 0);
            continue;
        }
        const float4 v = v8[q];
        *(float4*)(s_out + (int64_t)m * HD + lane * 4) = v;
        const float mean = mean8[q], rstd = rstd8[q];
        const float a = c8[q].x, b = c8[q].y, c = c8[q].z, d = c8[q].w;
        if (lane == 0) { stats[2 * (int64_t)m] = mean; stats[2 * (int64_t)m + 1] = rstd; }
        const float o0 = a * rstd * gm.x + bt.x, o1 = b * rstd * gm.y + bt.y, o2 = c * rstd * gm.z + bt.z, o3 = d * rstd * gm.w + bt.w;
        const uint2 h = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
        if (x_out) *(float4*)(x_out + (int64_t)m * HD + lane * 4) = make_float4(o0, o1, o2, o3);
        *(uint2*)(xh_out + (int64_t)m * HD + lane * 4) = h;
        *(uint2*)(a_lds + row * LDH + lane * 4) = h;
    }
}
// bf16 output of a GEMM with CB column blocks per wave into an LDS tile: t[row][n] = acc + bias
template <int CB>
__device__ __forceinline__ void bf16_to_lds(const f32x4 (&acc)[CB][RT], const float* __restrict__ bias, bf16_t* t, int ldt, int w, int lane) {
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int n0 = (w + 8 * c) * 16 + 4 * g;
        const float4 bb = *(const float4*)(bias + n0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            *(uint2*)(t + (rt * 16 + l15) * ldt + n0) = make_uint2(pack_bf2(acc[c][rt][0] + bb.x, acc[c][rt][1] + bb.y), pack_bf2(acc[c][rt][2] + bb.z, acc[c][rt][3] + bb.w));
    }
}
// an LDS tile of 64 rows x (16 NCH)-byte... NCH 16-byte pieces per row -> global rows of ldo elements (whole rows, 16 bytes per lane)
template <int NCH>
__device__ __forceinline__ void rows_out(const bf16_t* t, int ldt, bf16_t* __restrict__ out, int64_t ldo, int m0, int M, int tid) {
#pragma unroll
    for (int q = 0; q < RM * NCH / 512; ++q) {
        const int idx = tid + q * 512, row = idx / NCH, ch = idx - row * NCH;
        if (m0 + row < M) *(uint4*)(out + (int64_t)(m0 + row) * ldo + ch * 8) = *(const uint4*)(t + row * ldt + ch * 8);
    }
}

struct SegA {
    const bf16_t* ctx; Resid xres; const bf16_t* wo; const float* bo; Drop drop; const float *g1, *b1; const bf16_t* wcq; const float* bcq;
    float* s1; float* st1; float* x1; bf16_t* x1h; bf16_t* cq; int M; float eps;
};
struct SegB {
    const bf16_t* cctx; Resid x1; const bf16_t* wco; const float* bco; Drop drop3; const float *g2, *b2;
    const bf16_t* wfc1; const float* bfc1; const bf16_t* wfc2; const float* bfc2; Drop drop4; const float *g3, *b3;
    const bf16_t* wqkv; const float* bqkv;                       // the next layer's QKV projection, or null
    float *s2, *st2, *x2; bf16_t* x2h; bf16_t *pre, *act; float *s3, *st3, *x3; bf16_t* x3h; bf16_t* qkv;
    int M; float eps;
};

__global__ __launch_bounds__(512) void dec_chain_a_kernel(const SegA p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* a1 = (bf16_t*)smem;                                   // [RM][LDH] bf16 operand tile
    float* ssum = (float*)(smem + LDS_A1);                        // [RM][LDS_] pre-LayerNorm sums
    const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * RM;
    bf16x8 R[WIN];
    fill_window<2, 8>(R, wave_frags<2, 8>(p.wo, w), opaque(lane0));
    load_rows(p.ctx, a1, m0, p.M, opaque(tid));
    ResidRegs rr;
    resid_prefetch(p.xres, rr, m0, p.M, w, opaque(lane0));
    wg_barrier();
    {
        const int lane = opaque(lane0);
        f32x4 acc[2][RT];
        chain_gemm<2, 8, 2, 8, true>(R, wave_frags<2, 8>(p.wo, w), wave_frags<2, 8>(p.wcq, w), a1, LDH, lane, acc);
        resid_epilogue(acc, p.bo, p.xres, rr, p.drop, ssum, m0, p.M, w, lane);
    }
    wg_barrier();                // every wave has read the ctx tile and written its part of the sum tile
    ln_rows(ssum, p.g1, p.b1, p.eps, p.s1, p.x1, p.x1h, p.st1, a1, m0, p.M, w, opaque(lane0));
    wg_barrier();                // the sum tile is dead: it stages the cross query
    {
        const int lane = opaque(lane0);
        f32x4 acc[2][RT];
        chain_gemm<2, 8, 2, 8, false>(R, wave_frags<2, 8>(p.wcq, w), nullptr, a1, LDH, lane, acc);
        bf16_to_lds<2>(acc, p.bcq, (bf16_t*)ssum, LDH, w, lane);
    }
    wg_barrier();
    rows_out<32>((const bf16_t*)ssum, LDH, p.cq, HD, m0, p.M, opaque(tid));
}

__global__ __launch_bounds__(512) void dec_chain_b_kernel(const SegB p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* a1 = (bf16_t*)smem;                                   // [RM][LDH] bf16 operand tile
    bf16_t* a2 = (bf16_t*)(smem + LDS_A1);                        // [RM][LDW] the FFN activation; the sum tiles alias it
    float* ssum = (float*)a2;
    static_assert(RM * LDS_ * 4 <= RM * LDW * 2, "the sum tile must fit the activation tile");
    const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * RM;
    bf16x8 R[WIN];
    CH_STAMP(0);
    fill_window<2, 8>(R, wave_frags<2, 8>(p.wco, w), opaque(lane0));
    load_rows(p.cctx, a1, m0, p.M, opaque(tid));
    ResidRegs rr;
    resid_prefetch(p.x1, rr, m0, p.M, w, opaque(lane0));
    wg_barrier();
    CH_STAMP(1);
    {   // s2 = drop(cctx Wco^T + b) + x1
        const int lane = opaque(lane0);
        f32x4 acc[2][RT];
        chain_gemm<2, 8, 6, 8, true>(R, wave_frags<2, 8>(p.wco, w), wave_frags<6, 8>(p.wfc1, w), a1, LDH, lane, acc);
        resid_epilogue(acc, p.bco, p.x1, rr, p.drop3, ssum, m0, p.M, w, lane);
    }
    wg_barrier();
    CH_STAMP(2);
    ln_rows(ssum, p.g2, p.b2, p.eps, p.s2, p.x2, p.x2h, p.st2, a1, m0, p.M, w, opaque(lane0));
    // s2 / st2 come back from global memory below (the fc2 epilogue's residual, read in the MFMA layout by OTHER waves than the ones that
    // stored them): the stores must have completed before the barrier that orders the two -- a workgroup barrier alone orders LDS only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                // the sum tile is dead from here: the activation tile may be written
    CH_STAMP(3);
    {   // act = gelu(x2 Wfc1^T + b) -> the activation tile; the derivative (saved for the backward, KZV_EPI_GELU) leaves in three
        // passes of 256 columns through the operand tile, which the MFMAs no longer read
        const int lane = opaque(lane0), l15 = lane & 15, g = lane >> 4;
        f32x4 acc[6][RT];
        chain_gemm<6, 8, 2, 24, true>(R, wave_frags<6, 8>(p.wfc1, w), wave_frags<2, 24>(p.wfc2, w), a1, LDH, lane, acc);
        wg_barrier();            // every wave has read x2h
        CH_STAMP(4);
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = 2 * pass + cc;
                const int n0 = (w + 8 * c) * 16 + 4 * g;
                const float4 bb = *(const float4*)(p.bfc1 + n0);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int row = rt * 16 + l15;
                    float y[4], d[4];
                    gelu_erf_both(acc[c][rt][0] + bb.x, &y[0], &d[0]); gelu_erf_both(acc[c][rt][1] + bb.y, &y[1], &d[1]);
                    gelu_erf_both(acc[c][rt][2] + bb.z, &y[2], &d[2]); gelu_erf_both(acc[c][rt][3] + bb.w, &y[3], &d[3]);
                    *(uint2*)(a2 + row * LDW + n0) = make_uint2(pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]));
                    *(uint2*)(a1 + row * LDH + n0 - 256 * pass) = make_uint2(pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3]));
                }
            }
            wg_barrier();
            rows_out<32>(a1, LDH, p.pre + 256 * pass, FD, m0, p.M, opaque(tid));
            wg_barrier();
        }
    }
    CH_STAMP(5);
    rows_out<96>(a2, LDW, p.act, FD, m0, p.M, opaque(tid));       // (the last barrier above also covers the activation tile)
    CH_STAMP(6);
    f32x4 acc3[2][RT];
    const Resid res2{p.x2, p.s2, p.st2, p.g2, p.b2};              // x2: this workgroup's own rows, written above (ln_rows; many barriers ago)
    resid_prefetch(res2, rr, m0, p.M, w, opaque(lane0));
    {   // s3 = drop(act Wfc2^T + b) + x2
        const int lane = opaque(lane0);
        if (p.wqkv) chain_gemm<2, 24, 6, 8, true>(R, wave_frags<2, 24>(p.wfc2, w), wave_frags<6, 8>(p.wqkv, w), a2, LDW, lane, acc3);
        else chain_gemm<2, 24, 6, 8, false>(R, wave_frags<2, 24>(p.wfc2, w), nullptr, a2, LDW, lane, acc3);
    }
    wg_barrier();                // every wave has read the activation tile: the sum tile (same memory) may be written
    CH_STAMP(7);
    resid_epilogue(acc3, p.bfc2, res2, rr, p.drop4, ssum, m0, p.M, w, opaque(lane0));
    wg_barrier();
    CH_STAMP(8);
    ln_rows(ssum, p.g3, p.b3, p.eps, p.s3, p.x3, p.x3h, p.st3, a1, m0, p.M, w, opaque(lane0));
    CH_STAMP(9);
    if (!p.wqkv) return;
    wg_barrier();
    {   // the next layer's q | k | v = x3 Wqkv^T + b, staged through the (dead) activation tile
        const int lane = opaque(lane0);
        f32x4 acc[6][RT];
        chain_gemm<6, 8, 2, 8, false>(R, wave_frags<6, 8>(p.wqkv, w), nullptr, a1, LDH, lane, acc);
        bf16_to_lds<6>(acc, p.bqkv, a2, LDW, w, lane);
    }
    wg_barrier();
    CH_STAMP(10);
    rows_out<96>(a2, LDW, p.qkv, 3 * HD, m0, p.M, opaque(tid));
    CH_STAMP(11);
}

// ---- LM head + cross-entropy in one launch, logits never materialised (round 4, VERDICT r03 "missing" 1) -------------
// Replaces, for a training / validation step that does not return logits: lm_head.decoder (HF modeling_roberta.py:888-893, tied to the
// word embeddings) + CrossEntropyLoss(ignore_index = pad) (src/models/trocr_model.py:256,292) -- a [B*T, 4352] fp32 logits tensor written
// by the GEMM (267 MB at T = 60), read back by ce_kernel, and the bf16 gradient written behind it.  Rows are independent: one workgroup
// takes 64 rows of the head's LayerNorm output (bf16, in LDS) against the whole vocabulary twice, 256 columns at a time, the tied
// weight arriving in fragment order from L2 (2.2 MB, shared by every workgroup) through chain_gemm's window:
//   pass 1: logits chunk -> per-lane online (max, sum exp) over the lane's own columns, the target's logit picked out on the way;
//           the 32 partial pairs of a row (8 waves x 4 lane groups) meet in LDS -> lse[row]; loss += (lse - logit[target]) / count
//   pass 2: the same chunks again -> (exp(logit - lse) - onehot) / count as bf16 through an LDS tile, whole rows out (dlogits, the
//           operand of the head's two gradient GEMMs).  Skipped without a gradient buffer (validation).
// Same arithmetic as gemm_nt + ce_kernel up to fp32 summation order (bf16 operands, fp32 accumulation, fp32 softmax statistics).
struct HeadCE {
    const bf16_t* x; const bf16_t* wp; const float* bias; const int64_t* labels; const float* count; float* loss; bf16_t* dlogits;
    int M, L, T, V, Vp, nch, pad;
    const bf16_t* wpt; bf16_t* dh;      // optional: dh[M, 256] = dlogits . W (the head's input gradient), wpt = W^T [256, 256 nch] in fragment order
};
template <bool DH>
__global__ __launch_bounds__(512) void head_ce_kernel(const HeadCE p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* a1 = (bf16_t*)smem;                                   // [RM][LDH] the rows' operand tile
    bf16_t* stage = (bf16_t*)(smem + LDS_A1);                     // [RM][LDH] one chunk of dlogits on its way out
    float* red = (float*)(smem + 2 * LDS_A1);                     // [RM][32][2] partial (max, sum) pairs
    float* lse_s = red + RM * 64;                                 // [RM]
    float* tl_s = lse_s + RM;                                     // [RM] the target's logit
    int* tgt_s = (int*)(tl_s + RM);                               // [RM] target class, -1 = ignored row
    const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * RM;
    bf16x8 R[WIN];
    fill_window<2, 8>(R, wave_frags<2, 8>(p.wp, w), opaque(lane0));
    load_rows(p.x, a1, m0, p.M, opaque(tid));
    if (tid < RM) {
        const int m = m0 + tid;
        int t = -1;
        if (m < p.M) {
            const int b = m / p.T, pos = m - b * p.T;
            const int64_t g = p.labels[(int64_t)b * p.L + pos + 1];
            t = (g == p.pad || g < 0 || g >= p.V) ? -1 : (int)g;
        }
        tgt_s[tid] = t; tl_s[tid] = 0.f;
    }
    wg_barrier();
    const int lane = opaque(lane0), l15 = lane & 15, g = lane >> 4;
    int tg[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) tg[rt] = tgt_s[rt * 16 + l15];
    const float inv_cnt = 1.f / *p.count;
    // the chunk's logits of this lane: v[c][rt][r] at column n0(c) + r, row rt * 16 + l15; columns >= V -> -inf
    auto chunk_logits = [&](int j, const f32x4 (&acc)[2][RT], float (&v)[2][RT][4]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n0 = j * 256 + (w + 8 * c) * 16 + 4 * g;
            float bb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bb[r] = n0 + r < p.V ? p.bias[n0 + r] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[c][rt][r] = n0 + r < p.V ? acc[c][rt][r] + bb[r] : -INFINITY;
        }
    };
    // ---- pass 1 ----
    float mrun[RT], lrun[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { mrun[rt] = -INFINITY; lrun[rt] = 0.f; }
#pragma unroll 1
    for (int j = 0; j < p.nch; ++j) {
        f32x4 acc[2][RT];
        const bf16_t* nextw = p.wp + (int64_t)(j + 1 < p.nch ? j + 1 : 0) * 65536;          // the last chunk prefetches pass 2's first
        chain_gemm<2, 8, 2, 8, true>(R, wave_frags<2, 8>(p.wp + (int64_t)j * 65536, w), wave_frags<2, 8>(nextw, w), a1, LDH, lane, acc);
        float v[2][RT][4];
        chunk_logits(j, acc, v);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float mx = mrun[rt];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, v[c][rt][r]);
            if (mx > -INFINITY) {
                float sacc = lrun[rt] * __expf(mrun[rt] - mx);            // (exp(-inf - finite) = 0 on the first finite chunk)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc += __expf(v[c][rt][r] - mx);
                lrun[rt] = sacc; mrun[rt] = mx;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int n0 = j * 256 + (w + 8 * c) * 16 + 4 * g;
                const int d = tg[rt] - n0;
                if (d >= 0 && d < 4) tl_s[rt * 16 + l15] = d == 0 ? v[c][rt][0] : d == 1 ? v[c][rt][1] : d == 2 ? v[c][rt][2] : v[c][rt][3];
            }
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float* q = red + ((rt * 16 + l15) * 32 + w * 4 + g) * 2;
        q[0] = mrun[rt]; q[1] = lrun[rt];
    }
    wg_barrier();
    if (tid < RM) {                                   // one thread per row folds the 32 pairs in a fixed order
        float mx = -INFINITY;
        for (int k = 0; k < 32; ++k) mx = fmaxf(mx, red[(tid * 32 + k) * 2]);
        float se = 0.f;
        for (int k = 0; k < 32; ++k) se += red[(tid * 32 + k) * 2 + 1] * __expf(red[(tid * 32 + k) * 2] - mx);
        const float lse = mx + __logf(se);
        lse_s[tid] = lse;
        float contrib = tgt_s[tid] >= 0 ? (lse - tl_s[tid]) * inv_cnt : 0.f;
        contrib = wave_sum(contrib);
        if (tid == 0) atomicAdd(p.loss, contrib);
    }
    if (!p.dlogits) return;                           // wave-uniform (a kernel argument)
    wg_barrier();
    float ls[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ls[rt] = lse_s[rt * 16 + l15];
    // ---- pass 2 ----
    // With wpt / dh the head's input gradient rides along: dh = dlogits . W accumulates chunk by chunk from the staged dlogits tile (the
    // launch it replaces read the 134 MB of dlogits back: 92 us in the step), the window alternating between the two weight streams.
    constexpr bool with_dh = DH;                                               // compile time: no run-time branch around a load (see pin)
    const int KSV = p.nch * 8;                                                 // 32-deep steps of the vocabulary in the transposed pack
    const unsigned wo = (unsigned)lane * 16u;
    auto tfrag = [&](int j, int i) -> bf16x8 {                                 // fragment i = (vocabulary step 8 j + i / 2, hidden column block w + 8 (i % 2))
        return *(const bf16x8*)((const char*)p.wpt + ((int64_t)(w + 8 * (i & 1)) * KSV + 8 * j + (i >> 1)) * 1024 + wo);
    };
    f32x4 acc2[2][RT];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc2[c][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int j = 0; j < p.nch; ++j) {
        f32x4 acc[2][RT];
        if constexpr (with_dh) {
            chain_gemm<2, 8, 2, 8, false>(R, wave_frags<2, 8>(p.wp + (int64_t)j * 65536, w), nullptr, a1, LDH, lane, acc);
#pragma unroll
            for (int i = 0; i < WIN; ++i) R[i] = tfrag(j, i);                  // in flight under the softmax arithmetic below
        } else {
            if (j + 1 < p.nch) chain_gemm<2, 8, 2, 8, true>(R, wave_frags<2, 8>(p.wp + (int64_t)j * 65536, w), wave_frags<2, 8>(p.wp + (int64_t)(j + 1) * 65536, w), a1, LDH, lane, acc);
            else chain_gemm<2, 8, 2, 8, false>(R, wave_frags<2, 8>(p.wp + (int64_t)j * 65536, w), nullptr, a1, LDH, lane, acc);
        }
        float v[2][RT][4];
        chunk_logits(j, acc, v);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int nl = (w + 8 * c) * 16 + 4 * g, n0 = j * 256 + nl;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = tg[rt] >= 0 ? (__expf(v[c][rt][r] - ls[rt]) - (n0 + r == tg[rt] ? 1.f : 0.f)) * inv_cnt : 0.f;      // exp(-inf) = 0: padded columns
                *(uint2*)(stage + (rt * 16 + l15) * LDH + nl) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
        wg_barrier();
        {                                             // whole rows out: 16 bytes per lane, only the columns dlogits has (Vp may end inside the chunk)
            const int pieces = min(32, (p.Vp - j * 256) / 8);
            const int t2 = opaque(tid);
#pragma unroll
            for (int q = 0; q < RM * 32 / 512; ++q) {
                const int idx = t2 + q * 512, row = idx >> 5, ch = idx & 31;
                if (m0 + row < p.M && ch < pieces) *(uint4*)(p.dlogits + (int64_t)(m0 + row) * p.Vp + j * 256 + ch * 8) = *(const uint4*)(stage + row * LDH + ch * 8);
            }
        }
        if constexpr (with_dh) {                          // acc2 += (this chunk of dlogits) . W^T fragments; then the window goes back to the forward stream
            const bf16_t* ap = stage + l15 * LDH + g * 8;
            bf16x8 fa[RT];
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                if ((i & 1) == 0) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) fa[rt] = *(const bf16x8*)(ap + rt * 16 * LDH + (i >> 1) * 32);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc2[i & 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R[i], fa[rt], acc2[i & 1][rt], 0, 0, 0);
            }
            const int jn = j + 1 < p.nch ? j + 1 : 0;     // (the last refill is unused: no branch around the loads)
#pragma unroll
            for (int i = 0; i < WIN; ++i) R[i] = ld_frag<2, 8>(wave_frags<2, 8>(p.wp + (int64_t)jn * 65536, w), wo, i);
        }
        wg_barrier();
    }
    if constexpr (with_dh) {                              // dh rows out through the stage tile (bf16, as the GEMM epilogue rounded them)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int nl = (w + 8 * c) * 16 + 4 * g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                *(uint2*)(stage + (rt * 16 + l15) * LDH + nl) = make_uint2(pack_bf2(acc2[c][rt][0], acc2[c][rt][1]), pack_bf2(acc2[c][rt][2], acc2[c][rt][3]));
        }
        wg_barrier();
        rows_out<32>(stage, LDH, p.dh, HD, m0, p.M, opaque(tid));
    }
}

// ---- the decoder's input-gradient GEMMs on the same row-panel scheme (round 4; VERDICT r03 item 5, first half) --------------------
// dX[M, N] = dY[M, K] . W[K-out, N-in] for the six nn.Linear of a RobertaLayer (HF modeling_roberta.py:421-464) and the head's dense
// layer, B operand = the transposed weight copy in fragment order (a second set of packs, refreshed with the first).  At ~15k rows
// each of these was a 16 - 22 us launch of the 128 x 128 kernel at 7 % MFMA utilisation; one workgroup per 64 rows streams the
// weights from L2 through chain_gemm's window and leaves through an LDS tile as whole rows.  Epilogues: bf16 store; fp32 store of
// (acc + residual gradient) -- the "RESID" input-gradient form, residual optional; bf16 store of acc * gelu'(pre-activation) (DGELU).
// The weight gradients stay the grouped gemm_tn launch; the LayerNorm backward between the GEMMs stays its own kernel.
enum { DL_BF16 = 0, DL_RESID = 1, DL_DGELU = 2 };
struct DecLin { const bf16_t* a; const bf16_t* wp; void* out; const float* resid; const bf16_t* aux; int M; };
template <int CB, int KS, int EPI>
__global__ __launch_bounds__(512) void dec_lin_kernel(const DecLin p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K = KS * 32, N = CB * 128, LDA = K + 8, LDO16 = N + 8, LDO32 = N + 4;
    bf16_t* at = (bf16_t*)smem;                                   // [RM][LDA] operand rows; the output tile aliases it after the GEMM
    const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * RM;
    bf16x8 R[WIN];
    fill_window<CB, KS>(R, wave_frags<CB, KS>(p.wp, w), opaque(lane0));
    {
        const int t2 = opaque(tid);
#pragma unroll
        for (int q = 0; q < RM * (K / 8) / 512; ++q) {
            const int idx = t2 + q * 512, row = idx / (K / 8), ch = idx - row * (K / 8);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m0 + row < p.M) v = *(const uint4*)(p.a + (int64_t)(m0 + row) * K + ch * 8);
            *(uint4*)(at + row * LDA + ch * 8) = v;
        }
    }
    wg_barrier();
    const int lane = opaque(lane0), l15 = lane & 15, g = lane >> 4;
    f32x4 acc[CB][RT];
    chain_gemm<CB, KS, CB, KS, false>(R, wave_frags<CB, KS>(p.wp, w), nullptr, at, LDA, lane, acc);
    wg_barrier();                                                 // every wave has read the operand tile: it becomes the output tile
    if constexpr (EPI == DL_RESID) {
        float* ot = (float*)smem;                                 // [RM][LDO32] fp32
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int n0 = (w + 8 * c) * 16 + 4 * g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) *(f32x4*)(ot + (rt * 16 + l15) * LDO32 + n0) = acc[c][rt];
        }
        wg_barrier();
        const int t2 = opaque(tid);
#pragma unroll
        for (int q = 0; q < RM * (N / 4) / 512; ++q) {
            const int idx = t2 + q * 512, row = idx / (N / 4), ch = idx - row * (N / 4);
            if (m0 + row < p.M) {
                float4 v = *(const float4*)(ot + row * LDO32 + ch * 4);
                if (p.resid) { const float4 r = *(const float4*)(p.resid + (int64_t)(m0 + row) * N + ch * 4); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
                *(float4*)((float*)p.out + (int64_t)(m0 + row) * N + ch * 4) = v;
            }
        }
    } else {
        bf16_t* ot = (bf16_t*)smem;                               // [RM][LDO16] bf16
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int n0 = (w + 8 * c) * 16 + 4 * g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 v = acc[c][rt];
                if constexpr (EPI == DL_DGELU) {                  // the forward's saved gelu'(pre-activation), as nt_emit<KZV_EPI_DGELU>
                    const int m = m0 + rt * 16 + l15;
                    uint2 u = make_uint2(0, 0);
                    if (m < p.M) u = *(const uint2*)(p.aux + (int64_t)m * N + n0);
                    v[0] *= bf2f((bf16_t)(u.x & 0xffff)); v[1] *= bf2f((bf16_t)(u.x >> 16)); v[2] *= bf2f((bf16_t)(u.y & 0xffff)); v[3] *= bf2f((bf16_t)(u.y >> 16));
                }
                *(uint2*)(ot + (rt * 16 + l15) * LDO16 + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
        wg_barrier();
        const int t2 = opaque(tid);
#pragma unroll
        for (int q = 0; q < RM * (N / 8) / 512; ++q) {
            const int idx = t2 + q * 512, row = idx / (N / 8), ch = idx - row * (N / 8);
            if (m0 + row < p.M) *(uint4*)((bf16_t*)p.out + (int64_t)(m0 + row) * N + ch * 8) = *(const uint4*)(ot + row * LDO16 + ch * 8);
        }
    }
}

// ---- a decoder layer's BACKWARD linear chains (round 4, VERDICT r03 item 5, second half) ---------------------------------------------
// Between the two attention backwards a RobertaLayer's backward is row-local too (HF modeling_roberta.py:421-464 read upwards):
//   [input gradient of the Linear below a LayerNorm (+ the residual gradient)] -> LayerNorm backward -> [dropout mask, bf16] ->
//   [input gradient of the Linear above it]
// three times per layer (LN3: next layer's qkv | fc2 with gelu';  LN2: fc1 | cross output;  LN1: cross query | self output).  As
// launches these were gemm_nt / dec_lin (12.5 - 20.7 us) + ln_bwd_fast<1> (15.5 us) + gemm_nt / dec_lin (7.4 - 20.7 us); one workgroup
// takes 64 rows through all three with the forward chains' machinery (weights in fragment order through chain_gemm's window, operand
// tiles in LDS, every tensor out as whole rows).  Same arithmetic and rounding points as the launches it replaces: fp32 GEMM result
// + fp32 residual gradient, ln_bwd_fast_kernel's row arithmetic (lane = 4 columns, wave_sum), bf16(mask * dx) as the next operand,
// bf16 output (times the saved gelu' for fc2).  The weight gradients (which read dy16 / out2) stay the grouped gemm_tn launch; gamma /
// beta partial sums go to layernorm.hip's regions (one atomic per column and workgroup).
struct SegBwd {
    const bf16_t* a; const bf16_t* wp1; const float* resid;
    const float* x; const float* st; const float* gamma;
    float* dsum; bf16_t* dy16; Drop odrop; float* partial;
    const bf16_t* wp2; const bf16_t* aux; bf16_t* out2; int M;
};
template <int KS1, int NP2>
__global__ __launch_bounds__(512) void dec_bwd_seg_kernel(const SegBwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int K1 = KS1 * 32, LDA = K1 + 8, N2 = NP2 * 256, LDO = N2 + 8;
    bf16_t* a1 = (bf16_t*)smem;                                   // [RM][LDH] bf16(mask * dx): GEMM 2's operand
    char* xr = smem + LDS_A1;                                     // GEMM 1's operand rows -> the fp32 d tile -> partial sums -> GEMM 2's output rows
    bf16_t* at = (bf16_t*)xr; float* t32 = (float*)xr; bf16_t* ot = (bf16_t*)xr;
    const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * RM;
    bf16x8 R[WIN];
    SEG_STAMP(16);
    fill_window<2, KS1>(R, wave_frags<2, KS1>(p.wp1, w), opaque(lane0));
    {
        const int t2 = opaque(tid);
#pragma unroll
        for (int q = 0; q < RM * (K1 / 8) / 512; ++q) {
            const int idx = t2 + q * 512, row = idx / (K1 / 8), ch = idx - row * (K1 / 8);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m0 + row < p.M) v = *(const uint4*)(p.a + (int64_t)(m0 + row) * K1 + ch * 8);
            *(uint4*)(at + row * LDA + ch * 8) = v;
        }
    }
    // the LayerNorm backward's global operands of this wave's eight rows, requested now: they arrive under GEMM 1 (rows past M: the last row)
    float4 xv8[RM / 8], rv8[RM / 8]; float2 st8[RM / 8];
    {
        const int lane = opaque(lane0);
#pragma unroll
        for (int q = 0; q < RM / 8; ++q) {
            const int64_t mc = min(m0 + w * (RM / 8) + q, p.M - 1);
            xv8[q] = *(const float4*)(p.x + mc * HD + lane * 4);
            rv8[q] = *(const float4*)((p.resid ? p.resid : p.x) + mc * HD + lane * 4);      // no branch around a load (unused without a residual)
            st8[q] = *(const float2*)(p.st + 2 * mc);
        }
    }
    wg_barrier();
    SEG_STAMP(17);
    {
        const int lane = opaque(lane0), l15 = lane & 15, g = lane >> 4;
        f32x4 acc[2][RT];
        chain_gemm<2, KS1, 2, 8, true>(R, wave_frags<2, KS1>(p.wp1, w), wave_frags<2, 8>(p.wp2, w), at, LDA, lane, acc);
        wg_barrier();                                             // every wave has read the operand rows: the fp32 tile takes their place
        SEG_STAMP(18);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n0 = (w + 8 * c) * 16 + 4 * g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) *(f32x4*)(t32 + (rt * 16 + l15) * LDS_ + n0) = acc[c][rt];
        }
    }
    wg_barrier();
    SEG_STAMP(19);
    float4 dg = make_float4(0, 0, 0, 0), db = make_float4(0, 0, 0, 0);
    {   // LayerNorm backward, wave w: rows 8 w .. 8 w + 7 (ln_bwd_fast_kernel<1, true, false>'s arithmetic: kzv_common.h ln_bwd_*)
        const int lane = opaque(lane0);
        const float4 gm = *(const float4*)(p.gamma + lane * 4);
        const float invH = 1.f / (float)HD;
        retire_loads();                                           // the operands requested before GEMM 1: retired by hand (see pin)
#pragma unroll
        for (int q = 0; q < RM / 8; ++q) { pin(xv8[q]); pin(rv8[q]); pin(st8[q]); }
        float4 d8[RM / 8];                                        // every row out of LDS before the first LDS store below (hipcc keeps their order otherwise: one row at a time)
#pragma unroll
        for (int q = 0; q < RM / 8; ++q) d8[q] = *(const float4*)(t32 + (w * (RM / 8) + q) * LDS_ + lane * 4);
        // per row ln_bwd_fast_kernel's statements (kzv_common.h ln_bwd_*; the rows' wave sums one after the other, as there)
#pragma unroll
        for (int q = 0; q < RM / 8; ++q) {
            const int row = w * (RM / 8) + q, m = m0 + row;
            if (m >= p.M) {                                       // wave-uniform
                *(uint2*)(a1 + row * LDH + lane * 4) = make_uint2(0, 0);
                continue;
            }
            float4 d = d8[q];
            if (p.resid) { const float4 r = rv8[q]; d.x += r.x; d.y += r.y; d.z += r.z; d.w += r.w; }
            const float rstd = st8[q].y;
            const LnBwdTerms t = ln_bwd_terms(d, xv8[q], gm, st8[q].x, rstd);
            const float s1 = __fadd_rn(0.f, t.s1), s2 = __fadd_rn(0.f, t.s2);
            ln_bwd_accum(dg, db, d, t);
            const float m1 = __fmul_rn(wave_sum(s1), invH), m2 = __fmul_rn(wave_sum(s2), invH);
            float4 o = ln_bwd_dx(t, m1, m2, rstd);
            *(float4*)(p.dsum + (int64_t)m * HD + lane * 4) = o;
            if (p.odrop.thr16) {
                const unsigned e = (unsigned)m * (unsigned)HD + 4u * lane;
                const unsigned b0 = drop_bits(p.odrop.key, e >> 1), b1 = drop_bits(p.odrop.key, (e >> 1) + 1);
                o.x *= drop_keep(b0, 0, p.odrop.thr16, p.odrop.inv_keep); o.y *= drop_keep(b0, 1, p.odrop.thr16, p.odrop.inv_keep);
                o.z *= drop_keep(b1, 0, p.odrop.thr16, p.odrop.inv_keep); o.w *= drop_keep(b1, 1, p.odrop.thr16, p.odrop.inv_keep);
            }
            const uint2 h = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
            *(uint2*)(p.dy16 + (int64_t)m * HD + lane * 4) = h;
            *(uint2*)(a1 + row * LDH + lane * 4) = h;
        }
    }
    SEG_STAMP(20);
    wg_barrier();                                                 // the fp32 tile is dead: it holds the eight waves' gamma / beta partial sums
    {
        const int lane = opaque(lane0);
        float4* sg = (float4*)xr + (w * 2) * 64;
        sg[lane] = dg; sg[64 + lane] = db;
    }
    wg_barrier();
    {
        const int t2 = opaque(tid), col = t2 & 255, which = t2 >> 8;
        const float* sf = (const float*)xr;
        float a = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) a += sf[(ww * 2 + which) * HD + col];
        atomicAdd(p.partial + (size_t)(blockIdx.x % KZV_LN_SLOTS) * 2 * HD + which * HD + col, a);
    }
    wg_barrier();                                                 // ... and now GEMM 2's output rows
    SEG_STAMP(21);
    {
        const int lane = opaque(lane0), l15 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int pass = 0; pass < NP2; ++pass) {
            uint2 u[2][RT];
            if constexpr (NP2 > 1) {                              // fc2: the forward's saved gelu'(pre-activation), requested before the MFMAs
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int64_t m = min(m0 + rt * 16 + l15, p.M - 1);     // rows past M: the last row's (their products are never stored)
                        u[c][rt] = *(const uint2*)(p.aux + m * N2 + 256 * pass + (w + 8 * c) * 16 + 4 * g);
                    }
            }
            f32x4 acc[2][RT];
            if (pass + 1 < NP2) chain_gemm<2, 8, 2, 8, true>(R, wave_frags<2, 8>(p.wp2 + (int64_t)pass * 65536, w), wave_frags<2, 8>(p.wp2 + (int64_t)(pass + 1) * 65536, w), a1, LDH, lane, acc);
            else chain_gemm<2, 8, 2, 8, false>(R, wave_frags<2, 8>(p.wp2 + (int64_t)pass * 65536, w), nullptr, a1, LDH, lane, acc);
            if constexpr (NP2 > 1) {                              // gelu' (requested before the MFMAs): retired by hand
                retire_loads();
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) pin(u[c][rt]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int n0 = 256 * pass + (w + 8 * c) * 16 + 4 * g;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 v = acc[c][rt];
                    if constexpr (NP2 > 1) {
                        const uint2 uu = u[c][rt];
                        v[0] *= bf2f((bf16_t)(uu.x & 0xffff)); v[1] *= bf2f((bf16_t)(uu.x >> 16)); v[2] *= bf2f((bf16_t)(uu.y & 0xffff)); v[3] *= bf2f((bf16_t)(uu.y >> 16));
                    }
                    *(uint2*)(ot + (rt * 16 + l15) * LDO + n0) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                }
            }
        }
    }
    SEG_STAMP(22);
    wg_barrier();
    rows_out<32 * NP2>(ot, LDO, p.out2, N2, m0, p.M, opaque(tid));
    SEG_STAMP(23);
}

// every decoder weight of the model -> fragment order, one launch
struct PackDesc { const uint4* src; uint4* dst; int N, K, t0, n_valid, ld, k_valid; };
constexpr int PACK_PER_LAUNCH = 96;        // 40-byte descriptors in the kernel argument block (4 KiB limit)
struct PackTable { PackDesc d[PACK_PER_LAUNCH]; int n; };
__global__ void pack_frag_multi_kernel(const PackTable tab, int total) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    int k = 0;
    while (k + 1 < tab.n && t >= tab.d[k + 1].t0) ++k;
    const PackDesc& d = tab.d[k];
    const int u = t - d.t0, lane = u & 63, f = u >> 6, KS = d.K / 32;
    const int nb = f / KS, ks = f - nb * KS;
    const int row = nb * 16 + (lane & 15), col = ks * 32 + (lane >> 4) * 8;      // rows beyond n_valid / columns beyond k_valid (the vocabulary's padding) pack as zeros
    d.dst[u] = (row < d.n_valid && col < d.k_valid) ? d.src[((int64_t)row * d.ld + col) / 8] : make_uint4(0, 0, 0, 0);
}

}  // namespace

#ifdef KZV_STAMPS
extern "C" int kzv_debug_chain_stamps(long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chain_stamps), sizeof(long long) * (n < 32 ? n : 32)) == hipSuccess ? 0 : 1;
}
#endif

int kzv_dec_chain_supported(int Hd, int Fd) { return Hd == HD && Fd == FD; }

int kzv_dec_chain_a(const KzvDecChainA& a, hipStream_t s) {
    if (a.M < 1) return kzv_fail(KZV_E_ARG, "dec_chain_a: rows");
    SegA p;
    p.ctx = a.ctx; p.xres = Resid{a.xres, a.xres_s, a.xres_st, a.xres_g, a.xres_b}; p.wo = a.wo; p.bo = a.bo; p.g1 = a.g1; p.b1 = a.b1; p.wcq = a.wcq; p.bcq = a.bcq;
    p.s1 = a.s1; p.st1 = a.st1; p.x1 = a.x1; p.x1h = a.x1h; p.cq = a.cq; p.M = a.M; p.eps = a.eps;
    p.drop.key = a.drop_key; kzv_drop_params(a.drop_p, &p.drop.thr16, &p.drop.inv_keep);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)dec_chain_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_A); attr_done = true; }
    hipLaunchKernelGGL(dec_chain_a_kernel, dim3((a.M + RM - 1) / RM), dim3(512), LDS_A, s, p);
    return kzv_check_launch("dec_chain_a");
}

int kzv_dec_chain_b(const KzvDecChainB& a, hipStream_t s) {
    if (a.M < 1) return kzv_fail(KZV_E_ARG, "dec_chain_b: rows");
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)dec_chain_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B); attr_done = true; }
    SegB p;
    p.cctx = a.cctx; p.x1 = Resid{a.x1, a.s1, a.st1, a.g1, a.b1}; p.wco = a.wco; p.bco = a.bco; p.g2 = a.g2; p.b2 = a.b2; p.wfc1 = a.wfc1; p.bfc1 = a.bfc1; p.wfc2 = a.wfc2; p.bfc2 = a.bfc2;
    p.g3 = a.g3; p.b3 = a.b3; p.wqkv = a.wqkv; p.bqkv = a.bqkv;
    p.s2 = a.s2; p.st2 = a.st2; p.x2 = a.x2; p.x2h = a.x2h; p.pre = a.pre; p.act = a.act; p.s3 = a.s3; p.st3 = a.st3; p.x3 = a.x3; p.x3h = a.x3h; p.qkv = a.qkv;
    p.M = a.M; p.eps = a.eps;
    p.drop3.key = a.drop3_key; kzv_drop_params(a.drop_p, &p.drop3.thr16, &p.drop3.inv_keep);
    p.drop4.key = a.drop4_key; kzv_drop_params(a.drop_p, &p.drop4.thr16, &p.drop4.inv_keep);
    hipLaunchKernelGGL(dec_chain_b_kernel, dim3((a.M + RM - 1) / RM), dim3(512), LDS_B, s, p);
    return kzv_check_launch("dec_chain_b");
}

// N / K: output / reduction width (256 or 768, not both 768); epi: 0 bf16, 1 fp32 (+ resid), 2 bf16 * aux
int kzv_dec_lin(const bf16_t* a, const bf16_t* wp, void* out, const float* resid, const bf16_t* aux, int M, int N, int K, int epi, hipStream_t s) {
    if (!a || !wp || !out || M < 1 || (epi == DL_DGELU && !aux)) return kzv_fail(KZV_E_ARG, "dec_lin: bad argument");
    DecLin p{a, wp, out, resid, aux, M};
    const dim3 grid((M + RM - 1) / RM);
#define KZV_DL(CB_, KS_, E_)                                                                                         \
    do {                                                                                                             \
        constexpr int lds = RM * ((KS_ * 32 + 8) * 2 > (E_ == DL_RESID ? (CB_ * 128 + 4) * 4 : (CB_ * 128 + 8) * 2) ? (KS_ * 32 + 8) * 2 : (E_ == DL_RESID ? (CB_ * 128 + 4) * 4 : (CB_ * 128 + 8) * 2)); \
        static bool attr_done = false;                                                                               \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)dec_lin_kernel<CB_, KS_, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; } \
        hipLaunchKernelGGL((dec_lin_kernel<CB_, KS_, E_>), grid, dim3(512), lds, s, p);                              \
        return kzv_check_launch("dec_lin");                                                                          \
    } while (0)
    if (N == 256 && K == 256 && epi == DL_BF16) KZV_DL(2, 8, DL_BF16);
    if (N == 256 && K == 256 && epi == DL_RESID) KZV_DL(2, 8, DL_RESID);
    if (N == 256 && K == 768 && epi == DL_RESID) KZV_DL(2, 24, DL_RESID);
    if (N == 768 && K == 256 && epi == DL_DGELU) KZV_DL(6, 8, DL_DGELU);
#undef KZV_DL
    return kzv_fail(KZV_E_ARG, "dec_lin: no instance for N %d, K %d, epilogue %d", N, K, epi);
}

int kzv_dec_bwd_seg(const KzvDecBwdSeg& a, hipStream_t s) {
    if (!a.a || !a.wp1 || !a.x || !a.st || !a.gamma || !a.dgamma || !a.dbeta || !a.dsum || !a.dy16 || !a.wp2 || !a.out2 || a.M < 1 || (a.K1 != 256 && a.K1 != 768))
        return kzv_fail(KZV_E_ARG, "dec_bwd_seg: bad argument");
    bool fold_now = false;
    float* partial = kzv_ln_partial_region(a.dgamma, a.dbeta, HD, s, &fold_now);
    if (!partial) return KZV_E_ARG;
    SegBwd p{a.a, a.wp1, a.resid, a.x, a.st, a.gamma, a.dsum, a.dy16, Drop{0, 1.f, a.drop_key}, partial, a.wp2, a.aux, a.out2, a.M};
    kzv_drop_params(a.drop_p, &p.odrop.thr16, &p.odrop.inv_keep);
    const dim3 grid((a.M + RM - 1) / RM);
#define KZV_SEG(KS1_, NP2_)                                                                                          \
    do {                                                                                                             \
        constexpr int xa = RM * (KS1_ * 32 + 8) * 2, xo = RM * (NP2_ * 256 + 8) * 2, xt = RM * LDS_ * 4;             \
        constexpr int lds = LDS_A1 + (xa > xo ? (xa > xt ? xa : xt) : (xo > xt ? xo : xt));                          \
        static bool attr_done = false;                                                                               \
        if (!attr_done) { (void)hipFuncSetAttribute((const void*)dec_bwd_seg_kernel<KS1_, NP2_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; } \
        hipLaunchKernelGGL((dec_bwd_seg_kernel<KS1_, NP2_>), grid, dim3(512), lds, s, p);                            \
    } while (0)
    if (a.K1 == 256 && !a.aux) KZV_SEG(8, 1);
    else if (a.K1 == 768 && !a.aux) KZV_SEG(24, 1);
    else if (a.K1 == 256) KZV_SEG(8, 3);
    else KZV_SEG(24, 3);
#undef KZV_SEG
    { const int rc = kzv_check_launch("dec_bwd_seg"); if (rc != KZV_OK) return rc; }
    return fold_now ? kzv_ln_partial_fold(partial, a.dgamma, a.dbeta, HD, s) : KZV_OK;
}

int kzv_head_ce(const KzvHeadCE& a, hipStream_t s) {
    if (!a.x || !a.wp || !a.bias || !a.labels || !a.count || !a.loss || a.M < 1 || a.V < 1 || a.Vp % 8 || a.Vp < a.V) return kzv_fail(KZV_E_ARG, "head_ce: bad argument");
    HeadCE p;
    p.x = a.x; p.wp = a.wp; p.bias = a.bias; p.labels = a.labels; p.count = a.count; p.loss = a.loss; p.dlogits = a.dlogits;
    p.M = a.M; p.L = a.L; p.T = a.T; p.V = a.V; p.Vp = a.Vp; p.nch = (a.V + 255) / 256; p.pad = a.pad;
    p.wpt = (a.dlogits && a.dh) ? a.wpt : nullptr; p.dh = a.dh;
    constexpr int LDS_H = 2 * LDS_A1 + RM * 64 * 4 + RM * 4 * 3;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)head_ce_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_H);
        (void)hipFuncSetAttribute((const void*)head_ce_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_H);
        attr_done = true;
    }
    if (p.wpt) hipLaunchKernelGGL(head_ce_kernel<true>, dim3((a.M + RM - 1) / RM), dim3(512), LDS_H, s, p);
    else hipLaunchKernelGGL(head_ce_kernel<false>, dim3((a.M + RM - 1) / RM), dim3(512), LDS_H, s, p);
    return kzv_check_launch("head_ce");
}

int kzv_pack_frag_multi(const KzvPackJob* jobs, int n, hipStream_t s) {
    if (n < 1 || n > KZV_PACK_MAX_JOBS) return kzv_fail(KZV_E_ARG, "pack_frag_multi: 1..%d matrices", KZV_PACK_MAX_JOBS);
    for (int j0 = 0; j0 < n; j0 += PACK_PER_LAUNCH) {
        PackTable tab;
        int total = 0;
        const int cnt = n - j0 < PACK_PER_LAUNCH ? n - j0 : PACK_PER_LAUNCH;
        for (int i = 0; i < cnt; ++i) {
            const KzvPackJob& jb = jobs[j0 + i];
            if (jb.N % 16 || jb.K % 32) return kzv_fail(KZV_E_ARG, "pack_frag_multi: N %% 16, K %% 32");
            if (jb.ld % 8 || jb.k_valid % 8) return kzv_fail(KZV_E_ARG, "pack_frag_multi: ld %% 8, k_valid %% 8");
            tab.d[i] = PackDesc{(const uint4*)jb.src, (uint4*)jb.dst, jb.N, jb.K, total, jb.n_valid > 0 ? jb.n_valid : jb.N, jb.ld > 0 ? jb.ld : jb.K, jb.k_valid > 0 ? jb.k_valid : jb.K};
            total += jb.N * jb.K / 8;
        }
        tab.n = cnt;
        hipLaunchKernelGGL(pack_frag_multi_kernel, dim3((total + 255) / 256), dim3(256), 0, s, tab, total);
    }
    return kzv_check_launch("pack_frag_multi");
}
