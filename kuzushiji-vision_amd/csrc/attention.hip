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
// Q / K / V / O / dO are read once per (batch, head): every LDS-DMA of this unit carries the streaming hint (kzv_common.h KZV_GLDS_NT):
// attention forward 1.044 -> 1.012 ms per step, backward 2.63 -> 2.54, four same-box alternations (round 4)
#define KZV_GLDS_NT
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"

namespace {

// Geometry is a template parameter: NKT key tiles of 16 (EXACT: the launch has exactly NKT of them, so the tile loops carry no
// branches), 4 waves per workgroup (8 in the <= 288-token backward).  Exact instances exist for the two hot shapes -- 11 tiles
// (64x640 crops: 161 tokens) and 10 (cross attention over 160 patch tokens) -- the generic ones serve <= 192 and <= 288 tokens
// (the reference's default 1024x64 columns: 257 tokens).
//
// Both kernels are bound by VALU ISSUE, not by MFMA or HBM (round 3 measurement: a wave issues one vector instruction per ~4.7
// cycles whatever it is -- v_mul_lo_u32 included -- and the round-2 kernels spent ~1,300 instructions per 16-query tile
// forward and ~1,700 per 32-query slab backward).  So the code below is written for instruction count: LDS addresses are
// per-lane constants + immediates, the softmax normalisation and 1 / P(keep) are applied to the 16 outputs instead of the
// 44 probabilities, the dropout bits come from the packed 4 x 4-block generator of kzv_common.h, and dS crosses LDS as
// packed 8-byte pieces of a [key][query] image that the dQ product reads back transposed (ds_read_b64_tr_b16).
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
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ bf16x8 words8(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(bf16x8, (u32x4){a, b, c, d}); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// ================================================================================================ forward
// launch bounds = the occupancy the LDS images allow anyway (3 workgroups per CU at <= 192 key rows, 2 at 288)
template <int MODE, int NKT, bool EXACT>
__global__ __launch_bounds__(256, NKT <= 12 ? 3 : 2) void attn_fwd_kernel(const AttnP p) {
    constexpr int NKP = (NKT + 1) / 2;                // PV steps of 32 keys
    constexpr int SPK = NKT * 16, SPV = NKP * 32;     // K image rows; V image rows (zero rows up to the pair boundary)
    constexpr int QI = (NKT + 3) / 4;                 // query tiles per wave (the host checks ceil(Sq / 16) <= 4 * QI)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem; char* Vs = smem + SPK * 128;
    int* kvalid = (int*)(smem + (SPK + SPV) * 128);
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
    stage_image<SPK, 4>(Ks, p.K + (int64_t)b * p.Sk * p.ldk + h * 64, p.ldk, p.Sk, p.zero16, w, lane);
    stage_image<SPV, 4>(Vs, p.V + (int64_t)b * p.Sk * p.ldv + h * 64, p.ldv, p.Sk, p.zero16, w, lane);
    if (MODE == 1)
        for (int i = tid; i < SPK; i += 256) kvalid[i] = i < p.Sk && p.ids[(int64_t)b * p.ld_ids + i] != p.pad_id;
    // every query fragment this wave will need, requested while the K/V images are still in flight (one exposed
    // memory latency per workgroup instead of one per query tile)
    const int nkt = EXACT ? NKT : (p.Sk + 15) >> 4, nqt = (p.Sq + 15) >> 4;
    bf16x8 qf[QI][2];
#pragma unroll
    for (int it = 0; it < QI; ++it) {
        const int qc = min((w + 4 * it) * 16 + l15, p.Sq - 1);
        const bf16_t* qrow = p.Q + ((int64_t)b * p.Sq + qc) * p.ldq + h * 64 + 8 * g;
        qf[it][0] = *(const bf16x8*)qrow; qf[it][1] = *(const bf16x8*)(qrow + 32);
    }
    // per-lane LDS offsets: everything below is `constant + immediate`
    const unsigned kA = l15 * 128 + ((g ^ (l15 & 7)) << 4), kB = l15 * 128 + (((4 + g) ^ (l15 & 7)) << 4);
    unsigned vT[4];
    {
        const int row = 4 * g + (l15 >> 2), hb = (l15 >> 1) & 1;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vT[dt] = row * 128 + ((((dt * 2 + hb) ^ (row & 7))) << 4) + (l15 & 1) * 8;
    }
    const AttDropLane dl = att_drop_lane(l15 & 3, true);
    const unsigned thrm1x2 = (unsigned)((p.thr16 - 32768 - 1) & 0xffff) * 0x10001u;
    const unsigned nQ4 = (unsigned)(p.Sq + 3) >> 2, nK4 = (unsigned)(p.Sk + 3) >> 2;
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
        // S^T tile: key on the accumulator rows (4g + r), query on the lane column
        f32x4 s[NKT];
        float mx = -INFINITY;                                 // max of the RAW scores (scale > 0 keeps the order)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (EXACT || kt < nk) {
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Ks + kA + kt * 2048), q0, s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Ks + kB + kt * 2048), q1, s[kt], 0, 0, 0);
                if (MODE == 1 || (EXACT ? kt == NKT - 1 : kt >= nkt - 1)) {      // only the last valid tile can run past Sk
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + 4 * g + r;
                        const bool ok = MODE == 0 ? key < p.Sk : (kvalid[key] && key <= q);
                        s[kt][r] = ok ? s[kt][r] : -INFINITY;
                    }
                }
                mx = fmax3(mx, fmax3(s[kt][0], s[kt][1], s[kt][2]), s[kt][3]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const bool dead = mx == -INFINITY;                    // fully masked row -> zeros, LSE = +inf
        const float mref = dead ? 0.f : mx * sc;
        float sum = 0.f;
        unsigned pw[NKP * 4];                                 // bf16 pairs of the (dropped, un-normalised) probabilities
        const unsigned xw0 = (((unsigned)(b * p.heads + h) * nQ4 + ((unsigned)q >> 2)) * nK4 + g) * KZV_ATT_GOLD + p.key;
#pragma unroll
        for (int kt = 0; kt < NKP * 2; ++kt) {
            if (kt < NKT && (EXACT || kt < nk)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], sc, -mref)); sum += s[kt][r]; }
                unsigned w01 = pack_bf2(s[kt][0], s[kt][1]), w23 = pack_bf2(s[kt][2], s[kt][3]);
                if (p.thr16) {
                    unsigned u01, u23;
                    att_drop_u(dl, att_mix(xw0 + (unsigned)kt * (4u * KZV_ATT_GOLD)), &u01, &u23);
                    w01 &= att_keep_mask(u01, thrm1x2); w23 &= att_keep_mask(u23, thrm1x2);
                }
                pw[kt * 2] = w01; pw[kt * 2 + 1] = w23;
            } else {
                pw[kt * 2] = 0u; pw[kt * 2 + 1] = 0u;
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (p.LSE && g == 0 && q < p.Sq)
            p.LSE[((int64_t)b * p.heads + h) * p.Sq + q] = dead ? INFINITY : (mx * sc + log2f(sum)) * (1.f / LOG2E);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NKP; ++kp) {
            if (EXACT || 2 * kp < nk) {
                const bf16x8 pf = words8(pw[kp * 4], pw[kp * 4 + 1], pw[kp * 4 + 2], pw[kp * 4 + 3]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vf = cat8(lds_tr16(Vs + vT[dt] + kp * 4096), lds_tr16(Vs + vT[dt] + kp * 4096 + 2048));
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        }
        // softmax normalisation and the dropout scale 1 / P(keep) on the 16 outputs (the probabilities above are exp2 values)
        const float onorm = dead ? 0.f : p.inv_keep / sum;
        if (q < p.Sq) {
            bf16_t* orow = p.O + ((int64_t)b * p.Sq + q) * p.ldo + h * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *(uint2*)(orow + dt * 16) = make_uint2(pack_bf2(o[dt][0] * onorm, o[dt][1] * onorm), pack_bf2(o[dt][2] * onorm, o[dt][3] * onorm));
        }
        KZV_ASTAMP();
    }
}

// =============================================================================================== backward
// LDS budget <= 80 KiB so TWO workgroups share a CU (one stages / waits at a barrier while the other computes):
// K and V images stay resident; Q and dO come in 32-query slabs through a 2-deep ring (the slab of block qb+1
// is in flight during block qb); dS^T for the current slab only, as a [key][32 queries] image of 64-byte rows.
constexpr int SLAB = 32 * 128;                  // 32 rows x 64 bf16
// DS ("delta from the slab"): rowsum(dO . O) is computed per slab from the dO slab already in LDS and an O slab that rides
// the same DMA (one more 4-KiB buffer), instead of from O and dO ROWS fetched a second time in the prologue: 59 instead of
// 96 KB fetched before the first MFMA (the prologue is bound by what one CU can fetch, ~11 B per cycle).
constexpr bool bwd_ds(int MODE, int NKT, int NW, bool EXACT) { return NW == 4 && (EXACT || (MODE == 1 && NKT <= 8)); }
constexpr int bwd_lds_bytes(int NKT, int MODE, bool DS) {
    return ((NKT + 1) / 2 * 32) * 128 + NKT * 16 * 128 + 4 * SLAB + ((NKT + 1) / 2 * 32) * 64 + (MODE == 1 ? 3 : 2) * ((NKT + 1) / 2 * 32) * 4 + (DS ? SLAB : 0);
}
static_assert(bwd_lds_bytes(11, 0, true) <= 80 * 1024 && bwd_lds_bytes(12, 0, false) <= 80 * 1024 && bwd_lds_bytes(8, 1, true) <= 80 * 1024,
              "attention backward (<= 192 tokens) must fit two workgroups per CU");

// stage one 32-row slab of Q and of dO: 4 + 4 one-KiB pieces spread over the NW waves.  Rows past the last query are CLAMPED to
// it, not zero-filled: their log-sum-exp is +inf, so P = dS = 0 there and the (finite) operand rows never reach a result; that
// keeps the address a wave-uniform base + a 32-bit lane offset.
template <int NW>
__device__ __forceinline__ void stage_slabs(char* qs, char* os, const bf16_t* Qb, const bf16_t* dOb, unsigned ldq, unsigned ldo,
                                            int row0, int nvalid, int w, int lane, char* oo = nullptr, const bf16_t* Ob = nullptr) {
    // pieces 0-3: Q, 4-7: dO, 8-11: O (DS only).  With 4 waves, wave w issues piece w of each: the 8 rows whose delta it reduces itself.
    const int npc = oo ? 12 : 8;
    for (int pc = w; pc < npc; pc += NW) {
        const int pq = pc & 3;
        const int row = pq * 8 + (lane >> 3);
        const unsigned chunk = (lane & 7) ^ (row & 7);
        const unsigned r = (unsigned)min(row0 + row, nvalid - 1);
#ifdef KZV_ATT_M0_RESTORE
        if (pc < 4) glds16_asm_soff(Qb, (r * ldq + chunk * 8) * 2, qs + pq * 1024);
        else if (pc < 8) glds16_asm_soff(dOb, (r * ldo + chunk * 8) * 2, os + pq * 1024);
        else glds16_asm_soff(Ob, (r * ldo + chunk * 8) * 2, oo + pq * 1024);
#else       // every LDS-DMA of the backward kernel is issued from asm: M0 need not be handed back to the compiler
        if (pc < 4) glds16_asm_soff_m0(Qb, (r * ldq + chunk * 8) * 2, qs + pq * 1024);
        else if (pc < 8) glds16_asm_soff_m0(dOb, (r * ldo + chunk * 8) * 2, os + pq * 1024);
        else glds16_asm_soff_m0(Ob, (r * ldo + chunk * 8) * 2, oo + pq * 1024);
#endif
    }
}
// delta' of the 8 rows this wave staged itself (rows 8w .. 8w+7 of the slab): lane = (row, 16-byte chunk); its own DMA pieces are
// visible to it after its vmcnt wait, no barrier needed
__device__ __forceinline__ void slab_delta(const char* os, const char* oo, float* dlt_slab, float keep_p, int w, int lane) {
    const int row = w * 8 + (lane >> 3);
    const unsigned off = row * 128 + ((((unsigned)lane & 7) ^ (row & 7)) << 4);
    const bf16x8 a = *(const bf16x8*)(os + off), e = *(const bf16x8*)(oo + off);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) d += bf2f((bf16_t)a[j]) * bf2f((bf16_t)e[j]);
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if ((lane & 7) == 0) dlt_slab[row] = d * keep_p;
}
template <int SP, int NW>
__device__ __forceinline__ void stage_image_asm(char* img, const bf16_t* src, int64_t ld, int nvalid, const void* zero16,
                                                int w, int lane) {
    const int r8 = lane >> 3;
    for (int pc = w; pc < SP / 8; pc += NW) {
        const int row = pc * 8 + r8;
        const int chunk = (lane & 7) ^ (row & 7);
#ifdef KZV_ATT_M0_RESTORE
        glds16_asm(row < nvalid ? (const void*)(src + (int64_t)row * ld + chunk * 8) : zero16, img + pc * 1024);
#else
        glds16_asm_m0(row < nvalid ? (const void*)(src + (int64_t)row * ld + chunk * 8) : zero16, img + pc * 1024);
#endif
    }
}

#ifdef KZV_STAMPS
__device__ unsigned long long kzv_bwd_stamps[128];
#define KZV_BSTAMP() do { if (blockIdx.x == 771 && threadIdx.x == 0 && bsk < 128) kzv_bwd_stamps[bsk++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KZV_BSTAMP() do {} while (0)
#endif
template <int MODE, int NKT, int NW, bool EXACT>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_kernel(const AttnP p) {
    [[maybe_unused]] int bsk = 0;
    KZV_BSTAMP();
    constexpr int NKP = (NKT + 1) / 2, SP = NKP * 32;        // rows of the K and dS^T images (zero past the last key tile)
    constexpr int NT = NW * 64, TPW = (NKT + NW - 1) / NW, TB = 8 / NW;   // threads, key tiles per wave, dQ tiles per wave
    static_assert(TPW <= 3, "dK/dV accumulators of more than 3 key tiles per wave do not fit the register file");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem; char* Vs = Ks + SP * 128;
    char* Qring = Vs + NKT * 16 * 128; char* Oring = Qring + 2 * SLAB;
    char* dST = Oring + 2 * SLAB;                              // [SP keys][32 queries] bf16, 8-byte units swizzled by (key >> 1) & 7
    constexpr bool DS = bwd_ds(MODE, NKT, NW, EXACT);
    float* lse = (float*)(dST + SP * 64);
    float* dlt = lse + SP;
    char* Oo = (char*)(dlt + SP);                              // DS: the O slab (single buffer: consumed the moment it lands)
    int* kvalid = (int*)(Oo + (DS ? SLAB : 0));                // MODE 1 only
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x - b * p.heads;
    const bf16_t* Qb = p.Q + (int64_t)b * p.Sq * p.ldq + h * 64;
    const bf16_t* dOb = p.dO + (int64_t)b * p.Sq * p.ldo + h * 64;
    const bf16_t* Ob = p.O + (int64_t)b * p.Sq * p.ldo + h * 64;
    const float keep_p = 1.f / p.inv_keep;
  if constexpr (DS) {
    // log-sum-exp rows first (tracked loads: behind the asm DMAs the compiler's wait for them would cover the DMAs too)
    float lv = INFINITY;
    if (tid < SP && tid < p.Sq) lv = p.LSE[((int64_t)b * p.heads + h) * p.Sq + tid] * LOG2E;
    stage_slabs<NW>(Qring, Oring, Qb, dOb, (unsigned)p.ldq, (unsigned)p.ldo, 0, p.Sq, w, lane, Oo, Ob);     // first: needed first
    stage_image_asm<SP, NW>(Ks, p.K + (int64_t)b * p.Sk * p.ldk + h * 64, p.ldk, p.Sk, p.zero16, w, lane);
    stage_image_asm<NKT * 16, NW>(Vs, p.V + (int64_t)b * p.Sk * p.ldv + h * 64, p.ldv, p.Sk, p.zero16, w, lane);
    KZV_BSTAMP();
    if (tid < SP) lse[tid] = lv;
    for (int row = tid; row < SP; row += NT) if (row >= 32) dlt[row] = 0.f;      // rows past Sq of later slabs are written by slab_delta (finite, clamped rows)
    KZV_BSTAMP();
  } else {
    // log-sum-exp (in log2 units) and delta' = rowsum(dO . O) * P(keep) per query row: four lanes per row, 32 bytes each
    // (1 / P(keep) is taken out of dS and multiplied back into dQ / dK / dV at the very end).  Their loads are issued
    // FIRST and all at once -- vmcnt retires in order, so behind the image DMAs they would wait for all 46 KiB of them --
    // and the arithmetic runs while the DMAs fly.
    constexpr int DR = (SP + NT / 4 - 1) / (NT / 4);          // rows per thread group
    bf16x8 ov[DR][2], dv8[DR][2];
    float lv[DR];
#pragma unroll
    for (int i = 0; i < DR; ++i) {
        const int row = min((tid >> 2) + i * (NT / 4), p.Sq - 1);
        const bf16_t* orow = p.O + ((int64_t)b * p.Sq + row) * p.ldo + h * 64 + (tid & 3) * 16;
        const bf16_t* drow = dOb + (int64_t)row * p.ldo + (tid & 3) * 16;
        ov[i][0] = *(const bf16x8*)orow; ov[i][1] = *(const bf16x8*)(orow + 8);
        dv8[i][0] = *(const bf16x8*)drow; dv8[i][1] = *(const bf16x8*)(drow + 8);
        lv[i] = p.LSE[((int64_t)b * p.heads + h) * p.Sq + row];
    }
    stage_image_asm<SP, NW>(Ks, p.K + (int64_t)b * p.Sk * p.ldk + h * 64, p.ldk, p.Sk, p.zero16, w, lane);
    stage_image_asm<NKT * 16, NW>(Vs, p.V + (int64_t)b * p.Sk * p.ldv + h * 64, p.ldv, p.Sk, p.zero16, w, lane);
    stage_slabs<NW>(Qring, Oring, Qb, dOb, (unsigned)p.ldq, (unsigned)p.ldo, 0, p.Sq, w, lane);
    KZV_BSTAMP();
#pragma unroll
    for (int i = 0; i < DR; ++i) {
        const int row = (tid >> 2) + i * (NT / 4);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) d += bf2f((bf16_t)ov[i][c][j]) * bf2f((bf16_t)dv8[i][c][j]);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        if ((tid & 3) == 0 && row < SP) {
            lse[row] = row < p.Sq ? lv[i] * LOG2E : INFINITY;
            dlt[row] = row < p.Sq ? d * keep_p : 0.f;
        }
    }
    KZV_BSTAMP();
  }
    if (MODE == 1)
        for (int row = tid; row < SP; row += NT) kvalid[row] = row < p.Sk && p.ids[(int64_t)b * p.ld_ids + row] != p.pad_id;
    for (int i = tid; i < SP * 64 / 16; i += NT) ((uint4*)dST)[i] = make_uint4(0, 0, 0, 0);   // key rows no wave writes stay 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (DS) slab_delta(Oring, Oo, dlt, keep_p, w, lane);
    __syncthreads();
    KZV_BSTAMP();

    const int nkt = EXACT ? NKT : (p.Sk + 15) >> 4, nqb = (p.Sq + 31) >> 5;
    const float sc = p.scale * LOG2E;
    f32x4 dk[TPW][4], dv[TPW][4];
#pragma unroll
    for (int a = 0; a < TPW; ++a)
#pragma unroll
        for (int d = 0; d < 4; ++d) { dk[a][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[a][d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    // per-lane LDS offsets (constant over the whole sweep)
    const unsigned rA = l15 * 128 + ((g ^ (l15 & 7)) << 4), rB = l15 * 128 + (((4 + g) ^ (l15 & 7)) << 4);   // row fragments
    unsigned tT[4];                                                                                          // transposed fragments
    {
        const int row = 4 * g + (l15 >> 2), hb = (l15 >> 1) & 1;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) tT[dt] = row * 128 + ((((dt * 2 + hb) ^ (row & 7))) << 4) + (l15 & 1) * 8;
    }
    const AttDropLane dl = att_drop_lane(l15 & 3, false);
    const int thr_s = p.thr16 ? (int)p.thr16 - 32768 : -40000;          // no dropout: below every int16, everything is kept
    const unsigned nQ4 = (unsigned)(p.Sq + 3) >> 2, nK4 = (unsigned)(p.Sk + 3) >> 2;
    // pre-mix word of block (q >> 2 = g, k >> 2 = l15 >> 2) of slab 0; + per slab / 16-query half / key tile multiples of GOLD
    unsigned xslab = (((unsigned)(b * p.heads + h) * nQ4 + g) * nK4 + (l15 >> 2)) * KZV_ATT_GOLD + p.key;
    const unsigned xstep_t2 = 4u * nK4 * KZV_ATT_GOLD;

    for (int qb = 0; qb < nqb; ++qb) {
        const char* Qs = Qring + (qb & 1) * SLAB;
        const char* Os = Oring + (qb & 1) * SLAB;
        // ---------------- phase A: per owned key tile, S / dP / P / dS for 32 queries; dV^T, dK^T ----------
        // The row fragments of Q and dO (A operands of S and dP) are shared by every key tile of this wave: read ONCE per slab.
        // dO^T and Q^T over the slab (A operands of the dV / dK products) are re-read per tile, late, when the S / dP registers
        // are free again: kept across the tiles too, they make the kernel spill.  (log-sum-exp and delta' of the lane's rows are
        // re-read per tile as two 16-byte pieces: kept across the tiles they cost the 16 registers that make the kernel spill.)
        bf16x8 Qr[2][2], Or[2][2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            Qr[t2][0] = *(const bf16x8*)(Qs + rA + t2 * 2048); Qr[t2][1] = *(const bf16x8*)(Qs + rB + t2 * 2048);
            Or[t2][0] = *(const bf16x8*)(Os + rA + t2 * 2048); Or[t2][1] = *(const bf16x8*)(Os + rB + t2 * 2048);
        }
        // the next slab flies during this block's two phases; its (slow to issue) DMAs go out behind the LDS reads above
        if (qb + 1 < nqb)
            stage_slabs<NW>(Qring + ((qb + 1) & 1) * SLAB, Oring + ((qb + 1) & 1) * SLAB, Qb, dOb, (unsigned)p.ldq, (unsigned)p.ldo, (qb + 1) * 32, p.Sq, w, lane,
                            DS ? Oo : nullptr, Ob);
        KZV_BSTAMP();
#pragma unroll
        for (int a = 0; a < TPW; ++a) {
            const int kt = w + NW * a;
            if (kt >= nkt) continue;
            const int key = kt * 16 + l15;
            char* dsrow = dST + key * 64;
            const int dsz = (key >> 1) & 7;
            if (MODE == 1 && kt * 16 > qb * 32 + 31) {   // key tile entirely above the diagonal: dS = 0
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) *(uint2*)(dsrow + (((t2 * 4 + g) ^ dsz) << 3)) = make_uint2(0u, 0u);
                continue;
            }
            const bf16x8 k0 = *(const bf16x8*)(Ks + rA + kt * 2048), k1 = *(const bf16x8*)(Ks + rB + kt * 2048);
            const bf16x8 v0 = *(const bf16x8*)(Vs + rA + kt * 2048), v1 = *(const bf16x8*)(Vs + rB + kt * 2048);
            // keys past Sk (last tile) / padding keys: -inf as the INITIAL score accumulator (the key sits on the lane, so all
            // four registers take the same value) makes their probabilities exp2(-inf) = 0 at no cost per element
            const bool kok = MODE == 1 ? kvalid[key] != 0 : key < p.Sk;
            const float sinit = kok ? 0.f : -INFINITY;
            unsigned pdw[4], dsw[4];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const f32x4 lq4 = *(const f32x4*)(lse + qb * 32 + t2 * 16 + 4 * g), dq4 = *(const f32x4*)(dlt + qb * 32 + t2 * 16 + 4 * g);
                f32x4 S = (f32x4){sinit, sinit, sinit, sinit}, dP = (f32x4){0.f, 0.f, 0.f, 0.f};
                S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qr[t2][0], k0, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qr[t2][1], k1, S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Or[t2][0], v0, dP, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Or[t2][1], v1, dP, 0, 0, 0);
                unsigned u01 = 0, u23 = 0;
                if (p.thr16)      // block (q >> 2 = qb * 8 + t2 * 4 + g, key >> 2 = kt * 4 + (l15 >> 2)); this lane's column is key & 3
                    att_drop_u(dl, att_mix(xslab + (unsigned)t2 * xstep_t2 + (unsigned)kt * (4u * KZV_ATT_GOLD)), &u01, &u23);
                float pm[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(fmaf(S[r], sc, -lq4[r]));      // lse = +inf for q >= Sq / dead rows
                    if (MODE == 1) pr = key <= qb * 32 + t2 * 16 + 4 * g + r ? pr : 0.f;     // causal
                    const unsigned ur = (r & 2) ? u23 : u01;
                    const int us = (r & 1) ? (int)ur >> 16 : (int)(short)(ur & 0xffffu);
                    pm[r] = us >= thr_s ? pr : 0.f;
                    ds[r] = fmaf(pm[r], dP[r], -pr * dq4[r]);
                }
                pdw[t2 * 2] = pack_bf2(pm[0], pm[1]); pdw[t2 * 2 + 1] = pack_bf2(pm[2], pm[3]);
                dsw[t2 * 2] = pack_bf2(ds[0], ds[1]); dsw[t2 * 2 + 1] = pack_bf2(ds[2], ds[3]);
                *(uint2*)(dsrow + (((t2 * 4 + g) ^ dsz) << 3)) = make_uint2(dsw[t2 * 2], dsw[t2 * 2 + 1]);
            }
            const bf16x8 pf = words8(pdw[0], pdw[1], pdw[2], pdw[3]), df = words8(dsw[0], dsw[1], dsw[2], dsw[3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 dOt = cat8(lds_tr16(Os + tT[dt]), lds_tr16(Os + tT[dt] + 2048));
                const bf16x8 Qt = cat8(lds_tr16(Qs + tT[dt]), lds_tr16(Qs + tT[dt] + 2048));
                dv[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dOt, pf, dv[a][dt], 0, 0, 0);
                dk[a][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qt, df, dk[a][dt], 0, 0, 0);
            }
            KZV_BSTAMP();
        }
        KZV_BSTAMP();
        __syncthreads();
        KZV_BSTAMP();
        // ---------------- phase B: dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] for this 32-query slab ----
        {
            const int t2 = w / (NW / 2), dt0 = (w % (NW / 2)) * TB;
            const int q = qb * 32 + t2 * 16 + l15;
            f32x4 acc[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // this lane's piece of a transposed dS^T read: key row 8g + (l15 >> 2) (+4), the 4 queries of unit t2 * 4 + (l15 & 3)
            const int krow = 8 * g + (l15 >> 2);
            const unsigned dA = krow * 64 + (((t2 * 4 + (l15 & 3)) ^ ((krow >> 1) & 7)) << 3);
            const unsigned dB = (krow + 4) * 64 + (((t2 * 4 + (l15 & 3)) ^ (((krow + 4) >> 1) & 7)) << 3);
            unsigned kT[TB][2];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const int row = 8 * g + (l15 >> 2), chunk = (dt0 + u) * 2 + ((l15 >> 1) & 1);
                kT[u][0] = row * 128 + ((chunk ^ (row & 7)) << 4) + (l15 & 1) * 8;
                kT[u][1] = (row + 4) * 128 + ((chunk ^ ((row + 4) & 7)) << 4) + (l15 & 1) * 8;
            }
            const int nks = EXACT ? NKP : (nkt + 1) >> 1;
#pragma unroll
            for (int ks = 0; ks < NKP; ++ks) {
                if (EXACT || ks < nks) {
                    const bf16x8 dsf = cat8(lds_tr16(dST + dA + ks * 2048), lds_tr16(dST + dB + ks * 2048));
#pragma unroll
                    for (int u = 0; u < TB; ++u) {
                        const bf16x8 kf = cat8(lds_tr16(Ks + kT[u][0] + ks * 4096), lds_tr16(Ks + kT[u][1] + ks * 4096));
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, acc[u], 0, 0, 0);
                    }
                }
            }
            KZV_BSTAMP();
            // The next slab (this wave's pieces, issued a whole block ago) has landed.  The wait sits BEFORE the dQ
            // stores: vmcnt retires in issue order, so after them it would also wait for stores issued a moment ago.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (DS)      // delta' of the next slab's rows this wave staged (published by the barrier below)
                if (qb + 1 < nqb) slab_delta(Oring + ((qb + 1) & 1) * SLAB, Oo, dlt + (qb + 1) * 32, keep_p, w, lane);
            if (q < p.Sq) {
                const float osc = p.scale * p.inv_keep;
                bf16_t* row = p.dQ + ((int64_t)b * p.Sq + q) * p.ldq + h * 64 + 4 * g;
#pragma unroll
                for (int u = 0; u < TB; ++u) {
                    const int dt = dt0 + u;
                    *(uint2*)(row + dt * 16) = make_uint2(pack_bf2(acc[u][0] * osc, acc[u][1] * osc), pack_bf2(acc[u][2] * osc, acc[u][3] * osc));
                }
            }
        }
        xslab += 2u * xstep_t2;
        KZV_BSTAMP();
        __syncthreads();          // publishes everyone's slab pieces and frees dS^T
        KZV_BSTAMP();
    }
    const float ksc = p.scale * p.inv_keep;
#pragma unroll
    for (int a = 0; a < TPW; ++a) {
        const int key = (w + NW * a) * 16 + l15;
        if (w + NW * a >= nkt || key >= p.Sk) continue;
        bf16_t* krow = p.dK + ((int64_t)b * p.Sk + key) * p.ldk + h * 64 + 4 * g;
        bf16_t* vrow = p.dV + ((int64_t)b * p.Sk + key) * p.ldv + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *(uint2*)(krow + dt * 16) = make_uint2(pack_bf2(dk[a][dt][0] * ksc, dk[a][dt][1] * ksc), pack_bf2(dk[a][dt][2] * ksc, dk[a][dt][3] * ksc));
            *(uint2*)(vrow + dt * 16) = make_uint2(pack_bf2(dv[a][dt][0] * p.inv_keep, dv[a][dt][1] * p.inv_keep),
                                                   pack_bf2(dv[a][dt][2] * p.inv_keep, dv[a][dt][3] * p.inv_keep));
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
extern "C" int kzv_debug_bwd_stamps(unsigned long long* host128) {
    return hipMemcpyFromSymbol(host128, HIP_SYMBOL(kzv_bwd_stamps), 128 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

int kzv_attn_generic(const kzv_attn_args* a, int D, bool bwd, hipStream_t s);      // attention_generic.hip

template <int MODE, int NKT, bool EXACT>
static void launch_fwd(const AttnP& p, int blocks, hipStream_t s) {
    constexpr int lds = (NKT * 16 + (NKT + 1) / 2 * 32) * 128 + NKT * 16 * 4;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<MODE, NKT, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((attn_fwd_kernel<MODE, NKT, EXACT>), dim3(blocks), dim3(256), lds, s, p);
}
template <int MODE, int NKT, int NW, bool EXACT>
static void launch_bwd(const AttnP& p, int blocks, hipStream_t s) {
    constexpr int lds = bwd_lds_bytes(NKT, MODE, bwd_ds(MODE, NKT, NW, EXACT));
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<MODE, NKT, NW, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
    hipLaunchKernelGGL((attn_bwd_kernel<MODE, NKT, NW, EXACT>), dim3(blocks), dim3(NW * 64), lds, s, p);
}

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
    const int nkt = (a->Sk + 15) >> 4, nqt = (a->Sq + 15) >> 4, blocks = a->B * a->heads;
    KzvProfScope prof(2, 4.0 * a->B * a->heads * (double)a->Sq * a->Sk * 64, s);
    // exact-tile instances for the two hot shapes (a wave holds ceil(NKT / 4) query tiles: 12 for both)
    if (a->mode == 1) launch_fwd<1, 12, false>(p, blocks, s);
    else if (nkt == 11 && nqt <= 12) launch_fwd<0, 11, true>(p, blocks, s);
    else if (nkt == 10 && nqt <= 12) launch_fwd<0, 10, true>(p, blocks, s);
    else if (nkt <= 12 && nqt <= 12) launch_fwd<0, 12, false>(p, blocks, s);
    else launch_fwd<0, 18, false>(p, blocks, s);
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
    const int nkt = (a->Sk + 15) >> 4, nqb = (a->Sq + 31) >> 5, blocks = a->B * a->heads;
    KzvProfScope prof(3, 10.0 * a->B * a->heads * (double)a->Sq * a->Sk * 64, s);
    // the per-row arrays (log-sum-exp, delta) hold 32 * ceil(NKT / 2) queries: 192 for every <= 12-tile instance
    if (a->mode == 1 && nkt <= 8) launch_bwd<1, 8, 4, false>(p, blocks, s);           // the decoder (<= 128 positions)
    else if (a->mode == 1) launch_bwd<1, 12, 4, false>(p, blocks, s);
    else if (nkt == 11 && nqb <= 6) launch_bwd<0, 11, 4, true>(p, blocks, s);
    else if (nkt == 10 && nqb <= 5) launch_bwd<0, 10, 4, true>(p, blocks, s);
    else if (nkt <= 12 && nqb <= 6) launch_bwd<0, 12, 4, false>(p, blocks, s);
    else launch_bwd<0, 18, 8, false>(p, blocks, s);
    return kzv_check_launch("attn_bwd");
}
