// Multi-head attention forward / backward for short sequences (Sq, Sk <= 288), head_dim 64, bf16 MFMA.
//
// Replaces eager_attention_forward in HF ViT (modeling_vit.py:164-189: fp32 softmax, prob dropout) and the
// RoBERTa self/cross attention (modeling_roberta.py:158-183, 186-326) with masks from :645-680 built as
// src/models/trocr_model.py:278 does (causal AND key != pad; cross attention unmasked).
//
// One workgroup per (batch, head): the head's whole K and V (<= 288 x 64 bf16 = 36 KiB each)
// are staged once into LDS by LDS-DMA (XOR-swizzled 128-B rows), so HBM traffic is the algorithmic
// minimum (Q, K, V read once, O written once).
//
// forward : each wave owns 16-query tiles.  S^T = K.Q^T is computed with the KEY on the accumulator
//           rows, so a query's scores live in one lane column: softmax = register max/sum + 2 shuffles,
//           and the P^T accumulators are directly the B operand of O^T = V^T.P^T (no LDS round trip);
//           V^T fragments come from the row-major V image through ds_read_b64_tr_b16.
// backward: flash-style recompute from the saved log-sum-exp.  Each wave owns key tiles and keeps
//           dK^T, dV^T for them in registers across the query sweep (no cross-workgroup reduction);
//           S and dP are computed with the key on the lane so their accumulators are the B operands of
//           dV^T += dO^T.P and dK^T += Q^T.dS; only dS crosses LDS (bf16, 32-query slab) for dQ.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"

namespace {

// Geometry is a template parameter: KT key tiles of 16 (LDS images hold SP = 16*KT rows), NW waves per workgroup.
// (12, 4) serves Sq, Sk <= 192 (64x640 crops: 161 tokens); (18, 4 forward / 8 backward) serves <= 288 tokens
// (the reference's default 1024x64 columns: 257 tokens).
constexpr float LOG2E = 1.4426950408889634f;

struct AttnP {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; float* LSE;
    const bf16_t* dO; bf16_t* dQ; bf16_t* dK; bf16_t* dV;
    const void* zero16;
    int64_t ldq, ldk, ldv, ldo;
    const int64_t* ids; int64_t ld_ids; int pad_id;
    int B, heads, Sq, Sk;
    float scale; unsigned thr16; float inv_keep; unsigned key;
};

// stage `nvalid` rows (64 bf16 each, row stride ld) into a swizzled [SP][64] LDS image; rows >= nvalid are zero
template <int SP, int NW>
__device__ __forceinline__ void stage_image(char* img, const bf16_t* src, int64_t ld, int nvalid, const void* zero16,
                                            int w, int lane) {
    const int r8 = lane >> 3;
    for (int pc = w; pc < SP / 8; pc += NW) {
        const int row = pc * 8 + r8;
        const int chunk = (lane & 7) ^ (row & 7);
        const void* s = row < nvalid ? (const void*)(src + (int64_t)row * ld + chunk * 8) : zero16;
        glds16(s, img + pc * 1024);
    }
}
__device__ __forceinline__ bf16x8 frag_row(const char* img, int row, int chunk) {
    return *(const bf16x8*)(img + row * 128 + ((chunk ^ (row & 7)) << 4));
}
// transposing read: block rows rb..rb+3 (supplied by lane groups of 4), 16 columns starting at chunk c2 (2 chunks)
__device__ __forceinline__ bf16x4 frag_tr(const char* img, int rb, int c2, int l15) {
    const int row = rb + (l15 >> 2);
    const int chunk = c2 + ((l15 >> 1) & 1);
    return lds_tr16(img + row * 128 + ((chunk ^ (row & 7)) << 4) + (l15 & 1) * 8);
}
__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ bf16x8 pack8(const float* a, const float* b) {
    const unsigned u0 = pack_bf2(a[0], a[1]), u1 = pack_bf2(a[2], a[3]), u2 = pack_bf2(b[0], b[1]), u3 = pack_bf2(b[2], b[3]);
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    return __builtin_bit_cast(bf16x8, (u32x4){u0, u1, u2, u3});
}
__device__ __forceinline__ float keep_of(const AttnP& p, unsigned e) {
    const unsigned bits = drop_bits(p.key, e >> 1);
    return drop_keep(bits, e & 1, p.thr16, p.inv_keep);
}

// ================================================================================================ forward
// launch bounds = the occupancy the LDS images allow anyway (3 workgroups per CU at 192 key rows, 2 at 288): with the
// default target hipcc squeezed the kernel into 112 VGPRs and spent ~45 % of its VALU slots on v_accvgpr moves
template <int MODE, int KT>
__global__ __launch_bounds__(256, KT <= 12 ? 3 : 2) void attn_fwd_kernel(const AttnP p) {
    constexpr int SP = KT * 16, IMG = SP * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem; char* Vs = smem + IMG;
    int* kvalid = (int*)(smem + 2 * IMG);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads;
#ifdef KZV_STAMPS      // diagnostic build: workgroup timeline of a few blocks into the (unused in forward) dQ buffer
    unsigned long long* stp = (p.dQ && (blockIdx.x % 257) == 0 && tid == 0) ? (unsigned long long*)p.dQ + (blockIdx.x / 257) * 8 : nullptr;
    int stk = 0;
#define KZV_ASTAMP() do { if (stp) stp[stk++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KZV_ASTAMP() do {} while (0)
#endif
    KZV_ASTAMP();
    const bf16_t* Kb = p.K + (int64_t)b * p.Sk * p.ldk + h * 64;
    const bf16_t* Vb = p.V + (int64_t)b * p.Sk * p.ldv + h * 64;
    stage_image<SP, 4>(Ks, Kb, p.ldk, p.Sk, p.zero16, w, lane);
    stage_image<SP, 4>(Vs, Vb, p.ldv, p.Sk, p.zero16, w, lane);
    for (int i = tid; i < SP; i += 256) kvalid[i] = i < p.Sk && (MODE == 0 || p.ids[(int64_t)b * p.ld_ids + i] != p.pad_id);
    // every query fragment this wave will need, requested while the K/V images are still in flight (one exposed
    // memory latency per workgroup instead of one per query tile)
    constexpr int QI = (KT + 3) / 4;
    const int nkt = (p.Sk + 15) >> 4, nqt = (p.Sq + 15) >> 4;
    bf16x8 qf[QI][2];
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int qc = min((w + 4 * it) * 16 + l15, p.Sq - 1);
        const bf16_t* qrow = p.Q + ((int64_t)b * p.Sq + qc) * p.ldq + h * 64 + 8 * g;
        qf[it][0] = *(const bf16x8*)qrow; qf[it][1] = *(const bf16x8*)(qrow + 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    KZV_ASTAMP();

    const float sc = p.scale * LOG2E;
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int qt = w + 4 * it;
        if (qt >= nqt) break;
        const int q = qt * 16 + l15;
        const bf16x8 q0 = qf[it][0], q1 = qf[it][1];
        const int nk = MODE == 1 ? min(nkt, qt + 1) : nkt;   // causal: key tiles above the diagonal are empty
        // Unmasked attention (MODE 0) walks ALL KT key tiles of the LDS image (rows past Sk are zero and masked: <= 1 tile
        // of waste at 161 tokens) so that the tile loops carry no branches; the causal mode skips tiles above the diagonal.
        // The softmax / dropout arithmetic below is what bounds this kernel (VALU, not MFMA or HBM), so it is kept lean:
        // masks only where a key can be invalid, the 1/sqrt(d)*log2(e) scale folded into the exponent's FMA, the raw
        // v_exp_f32, and 1/sum * 1/keep folded into the dropout select.
        f32x4 s[KT];
        float mx = -INFINITY;                                 // max of the RAW scores (scale > 0 keeps the order)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (MODE == 0 || kt < nk) {
                const int krow = kt * 16 + l15;
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ks, krow, g), q0, s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Ks, krow, 4 + g), q1, s[kt], 0, 0, 0);
                if (MODE == 1 || kt >= nkt - 1) {             // unmasked attention: only key tiles from the last valid one on can run past Sk
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + 4 * g + r;
                        const bool ok = MODE == 0 ? key < p.Sk : (kvalid[key] && key <= q);
                        s[kt][r] = ok ? s[kt][r] : -INFINITY;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const bool dead = mx == -INFINITY;                    // fully masked row -> zeros, LSE = +inf
        const float mref = dead ? 0.f : mx * sc;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            if (MODE == 0 || kt < nk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sc, -mref)); sum += s[kt][r]; }
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = dead ? 0.f : 1.f / sum;
        if (p.LSE && g == 0 && q < p.Sq)
            p.LSE[((int64_t)b * p.heads + h) * p.Sq + q] = dead ? INFINITY : (mx * sc + log2f(sum)) * (1.f / LOG2E);
        // dropout element index = (row of P) * Sk_even + key: the row base is even, so this lane's 4 consecutive keys
        // are exactly two hash pairs (2 hashes per 4 probabilities instead of 4)
        const unsigned ebase = (unsigned)((b * p.heads + h) * p.Sq + q) * (unsigned)((p.Sk + 1) & ~1);
        const float keepv = inv * p.inv_keep;                 // value of a kept probability's multiplier
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            if (MODE == 0 || kt < nk) {
                if (p.thr16) {
                    const unsigned pr = (ebase + kt * 16 + 4 * g) >> 1;
                    const unsigned b0 = drop_bits(p.key, pr), b1 = drop_bits(p.key, pr + 1);
                    s[kt][0] *= (b0 & 0xffffu) >= p.thr16 ? keepv : 0.f; s[kt][1] *= (b0 >> 16) >= p.thr16 ? keepv : 0.f;
                    s[kt][2] *= (b1 & 0xffffu) >= p.thr16 ? keepv : 0.f; s[kt][3] *= (b1 >> 16) >= p.thr16 ? keepv : 0.f;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
                }
            }
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < KT / 2; ++kp) {
            if (MODE == 0 || 2 * kp < nk) {
                const bf16x8 pf = pack8((const float*)&s[2 * kp], (const float*)&s[2 * kp + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vf = cat8(frag_tr(Vs, kp * 32 + 4 * g, dt * 2, l15), frag_tr(Vs, kp * 32 + 16 + 4 * g, dt * 2, l15));
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        }
        if (q < p.Sq) {
            bf16_t* orow = p.O + ((int64_t)b * p.Sq + q) * p.ldo + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(uint2*)(orow + dt * 16) = make_uint2(pack_bf2(o[dt][0], o[dt][1]), pack_bf2(o[dt][2], o[dt][3]));
        }
        KZV_ASTAMP();
    }
}

// =============================================================================================== backward
// LDS budget <= 80 KiB so TWO workgroups share a CU (one stages / waits at a barrier while the other computes):
// K and V images stay resident; Q and dO come in 32-query slabs through a 2-deep ring (the slab of block qb+1
// is in flight during block qb); dS for the current slab only.
constexpr int SLAB = 32 * 128;                  // 32 rows x 64 bf16
constexpr int bwd_lds_bytes(int KT) { return 2 * (KT * 16 * 128) + 4 * SLAB + 32 * (KT * 32 + 16) + 3 * KT * 16 * 4; }
static_assert(bwd_lds_bytes(12) <= 80 * 1024, "attention backward (<= 192 tokens) must fit two workgroups per CU");

// stage one 32-row slab of Q and of dO: 4 + 4 one-KiB pieces spread over the NW waves
template <int NW>
__device__ __forceinline__ void stage_slabs(char* qs, char* os, const bf16_t* Qb, const bf16_t* dOb, int64_t ldq, int64_t ldo,
                                            int row0, int nvalid, const void* zero16, int w, int lane) {
    for (int pc = w; pc < 8; pc += NW) {
        const int pq = pc & 3;
        const int row = pq * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        const bool ok = row0 + row < nvalid;
        if (pc < 4) glds16_asm(ok ? (const void*)(Qb + (int64_t)(row0 + row) * ldq + chunk * 8) : zero16, qs + pq * 1024);
        else glds16_asm(ok ? (const void*)(dOb + (int64_t)(row0 + row) * ldo + chunk * 8) : zero16, os + pq * 1024);
    }
}
template <int SP, int NW>
__device__ __forceinline__ void stage_image_asm(char* img, const bf16_t* src, int64_t ld, int nvalid, const void* zero16,
                                                int w, int lane) {
    const int r8 = lane >> 3;
    for (int pc = w; pc < SP / 8; pc += NW) {
        const int row = pc * 8 + r8;
        const int chunk = (lane & 7) ^ (row & 7);
        glds16_asm(row < nvalid ? (const void*)(src + (int64_t)row * ld + chunk * 8) : zero16, img + pc * 1024);
    }
}

#ifdef KZV_STAMPS
__device__ unsigned long long kzv_bwd_stamps[64];
#define KZV_BSTAMP() do { if (blockIdx.x == 771 && threadIdx.x == 0 && bsk < 64) kzv_bwd_stamps[bsk++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KZV_BSTAMP() do {} while (0)
#endif
template <int MODE, int KT, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_kernel(const AttnP p) {
    [[maybe_unused]] int bsk = 0;
    KZV_BSTAMP();
    constexpr int SP = KT * 16, IMG = SP * 128, DS_STRIDE = SP * 2 + 16, DS_BYTES = 32 * DS_STRIDE;
    constexpr int NT = NW * 64, TPW = (KT + NW - 1) / NW, TB = 8 / NW;   // threads, key tiles per wave, dQ tiles per wave
    static_assert(TPW <= 3, "dK/dV accumulators of more than 3 key tiles per wave do not fit the register file");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem; char* Vs = smem + IMG;
    char* Qring = smem + 2 * IMG; char* Oring = Qring + 2 * SLAB;
    char* dS = Oring + 2 * SLAB;
    float* lse = (float*)(dS + DS_BYTES);
    float* dlt = lse + SP;
    int* kvalid = (int*)(dlt + SP);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads;
    const bf16_t* Qb = p.Q + (int64_t)b * p.Sq * p.ldq + h * 64;
    const bf16_t* dOb = p.dO + (int64_t)b * p.Sq * p.ldo + h * 64;
    stage_image_asm<SP, NW>(Ks, p.K + (int64_t)b * p.Sk * p.ldk + h * 64, p.ldk, p.Sk, p.zero16, w, lane);
    stage_image_asm<SP, NW>(Vs, p.V + (int64_t)b * p.Sk * p.ldv + h * 64, p.ldv, p.Sk, p.zero16, w, lane);
    stage_slabs<NW>(Qring, Oring, Qb, dOb, p.ldq, p.ldo, 0, p.Sq, p.zero16, w, lane);
    for (int row = tid; row < SP; row += NT) {
        kvalid[row] = row < p.Sk && (MODE == 0 || p.ids[(int64_t)b * p.ld_ids + row] != p.pad_id);
        float l = INFINITY, d = 0.f;
        if (row < p.Sq) {
            l = p.LSE[((int64_t)b * p.heads + h) * p.Sq + row];
            const bf16_t* orow = p.O + ((int64_t)b * p.Sq + row) * p.ldo + h * 64;
            const bf16_t* drow = dOb + (int64_t)row * p.ldo;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bf16x8 a = *(const bf16x8*)(orow + c * 8), e = *(const bf16x8*)(drow + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) d += bf2f((bf16_t)a[j]) * bf2f((bf16_t)e[j]);
            }
        }
        lse[row] = l * LOG2E; dlt[row] = d;
    }
    for (int i = tid; i < DS_BYTES / 16; i += NT) ((uint4*)dS)[i] = make_uint4(0, 0, 0, 0);   // key columns no wave writes stay 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    KZV_BSTAMP();

    const int nkt = (p.Sk + 15) >> 4, nqb = (p.Sq + 31) >> 5;
    const float sc = p.scale * LOG2E;
    f32x4 dk[TPW][4], dv[TPW][4];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int d = 0; d < 4; ++d) { dk[a][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[a][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    for (int qb = 0; qb < nqb; ++qb) {
        const char* Qs = Qring + (qb & 1) * SLAB;
        const char* Os = Oring + (qb & 1) * SLAB;
        if (qb + 1 < nqb)      // next slab flies during this block's two phases
            stage_slabs<NW>(Qring + ((qb + 1) & 1) * SLAB, Oring + ((qb + 1) & 1) * SLAB, Qb, dOb, p.ldq, p.ldo, (qb + 1) * 32, p.Sq, p.zero16, w, lane);
        // ---------------- phase A: per owned key tile, S / dP / P / dS for 32 queries; dV^T, dK^T ----------
        bf16x8 dOt[4], Qt[4];     // A operands shared by all key tiles of this wave: dO^T and Q^T over the slab
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dOt[dt] = cat8(frag_tr(Os, 4 * g, dt * 2, l15), frag_tr(Os, 16 + 4 * g, dt * 2, l15));
            Qt[dt] = cat8(frag_tr(Qs, 4 * g, dt * 2, l15), frag_tr(Qs, 16 + 4 * g, dt * 2, l15));
        }
        float lq[8], dq8[8];      // log-sum-exp and delta of this lane's 8 query rows: read once per slab, not per key tile
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int q = qb * 32 + (r >> 2) * 16 + 4 * g + (r & 3);
            lq[r] = lse[q]; dq8[r] = dlt[q];
        }
#pragma unroll
        for (int a = 0; a < TPW; ++a) {
            const int kt = w + NW * a;
            if (kt >= nkt) continue;
            if (MODE == 1 && kt * 16 > qb * 32 + 31) {   // key tile entirely above the diagonal: dS = 0
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    *(bf16_t*)(dS + ((r >> 2) * 16 + 4 * g + (r & 3)) * DS_STRIDE + (kt * 16 + l15) * 2) = 0;
                continue;
            }
            const int key = kt * 16 + l15;
            const bf16x8 k0 = frag_row(Ks, key, g), k1 = frag_row(Ks, key, 4 + g);
            const bf16x8 v0 = frag_row(Vs, key, g), v1 = frag_row(Vs, key, 4 + g);
            const bool kok = kvalid[key];
            float pd[8], ds[8];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int qrow = t2 * 16 + l15;
                f32x4 S = (f32x4){0.f, 0.f, 0.f, 0.f}, dP = (f32x4){0.f, 0.f, 0.f, 0.f};
                S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Qs, qrow, g), k0, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Qs, qrow, 4 + g), k1, S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Os, qrow, g), v0, dP, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(Os, qrow, 4 + g), v1, dP, 0, 0, 0);
                // dropout bits: a hash covers the key PAIR (key & ~1, key | 1) of one query row, and this lane's neighbour
                // (l15 ^ 1) needs the same four (row, pair) hashes: each computes two rows and they swap through DPP
                unsigned hb[4] = {0u, 0u, 0u, 0u};
                if (p.thr16) {
                    const int par = l15 & 1;
                    unsigned own[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q = qb * 32 + t2 * 16 + 4 * g + 2 * par + u;
                        const unsigned e = (unsigned)((b * p.heads + h) * p.Sq + q) * (unsigned)((p.Sk + 1) & ~1) + (unsigned)key;
                        own[u] = drop_bits(p.key, e >> 1);
                    }
                    const unsigned n0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own[0], 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    const unsigned n1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own[1], 0xB1, 0xF, 0xF, true);
                    hb[0] = par ? n0 : own[0]; hb[1] = par ? n1 : own[1];
                    hb[2] = par ? own[0] : n0; hb[3] = par ? own[1] : n1;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qb * 32 + t2 * 16 + 4 * g + r;
                    const bool ok = kok && (MODE == 0 || key <= q);
                    float pr = ok ? __builtin_amdgcn_exp2f(S[r] * sc - lq[t2 * 4 + r]) : 0.f;      // lse = +inf for q >= Sq / dead rows
                    float kp = 1.f;
                    if (p.thr16) kp = drop_keep(hb[r], key & 1, p.thr16, p.inv_keep);
                    pd[t2 * 4 + r] = pr * kp;
                    ds[t2 * 4 + r] = pr * (dP[r] * kp - dq8[t2 * 4 + r]);
                    *(bf16_t*)(dS + (t2 * 16 + 4 * g + r) * DS_STRIDE + key * 2) = f2bf(ds[t2 * 4 + r]);
                }
            }
            const bf16x8 pf = pack8(pd, pd + 4), df = pack8(ds, ds + 4);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dOt[dt], pf, dv[a][dt], 0, 0, 0);
                dk[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qt[dt], df, dk[a][dt], 0, 0, 0);
            }
        }
        KZV_BSTAMP();
        __syncthreads();
        KZV_BSTAMP();
        // ---------------- phase B: dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] for this 32-query slab ----
        {
            const int t2 = w / (NW / 2), dt0 = (w % (NW / 2)) * TB;
            const int qloc = t2 * 16 + l15;
            const int q = qb * 32 + qloc;
            f32x4 acc[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int nks = (nkt + 1) >> 1;
            for (int ks = 0; ks < nks; ++ks) {
                const bf16x8 dsf = *(const bf16x8*)(dS + qloc * DS_STRIDE + (ks * 32 + 8 * g) * 2);
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    const int dt = dt0 + u;
                    const bf16x8 kf = cat8(frag_tr(Ks, ks * 32 + 8 * g, dt * 2, l15), frag_tr(Ks, ks * 32 + 8 * g + 4, dt * 2, l15));
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, acc[u], 0, 0, 0);
                }
            }
            // The next slab (this wave's pieces, issued a whole block ago) has landed.  The wait sits BEFORE the dQ
            // stores: vmcnt retires in issue order, so after them it would also wait for stores issued a moment ago.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (q < p.Sq) {
                bf16_t* row = p.dQ + ((int64_t)b * p.Sq + q) * p.ldq + h * 64 + 4 * g;
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    const int dt = dt0 + u;
                    *(uint2*)(row + dt * 16) = make_uint2(pack_bf2(acc[u][0] * p.scale, acc[u][1] * p.scale),
                                                          pack_bf2(acc[u][2] * p.scale, acc[u][3] * p.scale));
                }
            }
        }
        KZV_BSTAMP();
        __syncthreads();          // publishes everyone's slab pieces and frees dS
        KZV_BSTAMP();
    }
#pragma unroll
    for (int a = 0; a < TPW; ++a) {
        const int key = (w + NW * a) * 16 + l15;
        if (w + NW * a >= nkt || key >= p.Sk) continue;
        bf16_t* krow = p.dK + ((int64_t)b * p.Sk + key) * p.ldk + h * 64 + 4 * g;
        bf16_t* vrow = p.dV + ((int64_t)b * p.Sk + key) * p.ldv + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *(uint2*)(krow + dt * 16) = make_uint2(pack_bf2(dk[a][dt][0] * p.scale, dk[a][dt][1] * p.scale),
                                                   pack_bf2(dk[a][dt][2] * p.scale, dk[a][dt][3] * p.scale));
            *(uint2*)(vrow + dt * 16) = make_uint2(pack_bf2(dv[a][dt][0], dv[a][dt][1]), pack_bf2(dv[a][dt][2], dv[a][dt][3]));
        }
    }
}

int fill(AttnP& p, const kzv_attn_args* a, bool bwd) {
    if (!a || !a->Q || !a->K || !a->V || !a->O) return kzv_fail(KZV_E_ARG, "attn: null operand");
    if (a->Sq <= 0 || a->Sk <= 0 || a->Sq > 288 || a->Sk > 288) return kzv_fail(KZV_E_ARG, "attn: Sq/Sk must be in 1..288");
    if (a->mode == 1 && a->Sq > 192) return kzv_fail(KZV_E_ARG, "attn: causal mode is built for <= 192 tokens");
    if (a->mode == 1 && (!a->ids || a->Sq != a->Sk)) return kzv_fail(KZV_E_ARG, "attn: causal mode needs ids and Sq == Sk");
    if (a->mode != 0 && a->mode != 1) return kzv_fail(KZV_E_ARG, "attn: unknown mode");
    if ((a->ldq | a->ldk | a->ldv | a->ldo) % 8) return kzv_fail(KZV_E_ARG, "attn: row strides must be multiples of 8");
    if (bwd && (!a->dO || !a->dQ || !a->dK || !a->dV || !a->LSE)) return kzv_fail(KZV_E_ARG, "attn_bwd: null gradient operand");
    p.Q = (const bf16_t*)a->Q; p.K = (const bf16_t*)a->K; p.V = (const bf16_t*)a->V; p.O = (bf16_t*)a->O; p.LSE = a->LSE;
    p.dO = (const bf16_t*)a->dO; p.dQ = (bf16_t*)a->dQ; p.dK = (bf16_t*)a->dK; p.dV = (bf16_t*)a->dV;
    p.zero16 = kzv_zero_page();
    if (!p.zero16) return kzv_fail(KZV_E_HIP, "attn: zero page unavailable");
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.ids = a->ids; p.ld_ids = a->ld_ids; p.pad_id = a->pad_id;
    p.B = a->B; p.heads = a->heads; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = 0.125f;   // head_dim^-0.5, head_dim = 64
    kzv_drop_params(a->drop_p, &p.thr16, &p.inv_keep);
    p.key = a->drop_key;
    return KZV_OK;
}

}  // namespace

#ifdef KZV_STAMPS
extern "C" int kzv_debug_bwd_stamps(unsigned long long* host64) {
    return hipMemcpyFromSymbol(host64, HIP_SYMBOL(kzv_bwd_stamps), 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

int kzv_attn_generic(const kzv_attn_args* a, int D, bool bwd, hipStream_t s);      // attention_generic.hip

extern "C" int kzv_attn_fwd(const kzv_attn_args* a, void* stream) {
    if (a && a->head_dim != 0 && a->head_dim != 64) {
        if (!a->Q || !a->K || !a->V || !a->O) return kzv_fail(KZV_E_ARG, "attn: null operand");
        if ((a->ldq | a->ldk | a->ldv | a->ldo) % 8) return kzv_fail(KZV_E_ARG, "attn: row strides must be multiples of 8");
        KzvProfScope prof(2, 4.0 * a->B * a->heads * (double)a->Sq * a->Sk * a->head_dim, (hipStream_t)stream);
        return kzv_attn_generic(a, a->head_dim, false, (hipStream_t)stream);
    }
    AttnP p;
    if (int rc = fill(p, a, false)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const bool big = a->Sq > 192 || a->Sk > 192;
    const int lds = big ? 2 * (288 * 128) + 288 * 4 : 2 * (192 * 128) + 192 * 4;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<0, 18>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (288 * 128) + 288 * 4); attr = true; }
    KzvProfScope prof(2, 4.0 * a->B * a->heads * (double)a->Sq * a->Sk * 64, s);
    const dim3 grid(a->B * a->heads);
    if (a->mode == 1) hipLaunchKernelGGL((attn_fwd_kernel<1, 12>), grid, dim3(256), lds, s, p);
    else if (big) hipLaunchKernelGGL((attn_fwd_kernel<0, 18>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<0, 12>), grid, dim3(256), lds, s, p);
    return kzv_check_launch("attn_fwd");
}

extern "C" int kzv_attn_bwd(const kzv_attn_args* a, void* stream) {
    if (a && a->head_dim != 0 && a->head_dim != 64) {
        if (!a->Q || !a->K || !a->V || !a->O || !a->dO || !a->dQ || !a->dK || !a->dV || !a->LSE) return kzv_fail(KZV_E_ARG, "attn_bwd: null operand");
        if ((a->ldq | a->ldk | a->ldv | a->ldo) % 8) return kzv_fail(KZV_E_ARG, "attn: row strides must be multiples of 8");
        KzvProfScope prof(3, 10.0 * a->B * a->heads * (double)a->Sq * a->Sk * a->head_dim, (hipStream_t)stream);
        return kzv_attn_generic(a, a->head_dim, true, (hipStream_t)stream);
    }
    AttnP p;
    if (int rc = fill(p, a, true)) return rc;
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<0, 12, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes(12));
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<1, 12, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes(12));
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<0, 18, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_lds_bytes(18));
        attr = true;
    }
    const bool big = a->Sq > 192 || a->Sk > 192;
    KzvProfScope prof(3, 10.0 * a->B * a->heads * (double)a->Sq * a->Sk * 64, s);
    const dim3 grid(a->B * a->heads);
    if (a->mode == 1) hipLaunchKernelGGL((attn_bwd_kernel<1, 12, 4>), grid, dim3(256), bwd_lds_bytes(12), s, p);
    else if (big) hipLaunchKernelGGL((attn_bwd_kernel<0, 18, 8>), grid, dim3(512), bwd_lds_bytes(18), s, p);
    else hipLaunchKernelGGL((attn_bwd_kernel<0, 12, 4>), grid, dim3(256), bwd_lds_bytes(12), s, p);
    return kzv_check_launch("attn_bwd");
}
