// Shared device helpers for the kzv HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // one ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
typedef __attribute__((ext_vector_type(2))) float f2_t;

#define KZV_LDS __attribute__((address_space(3)))
#define KZV_GLB __attribute__((address_space(1)))

__device__ __forceinline__ float bf2f(bf16_t x) { return __builtin_bit_cast(float, (unsigned)x << 16); }
// round-to-nearest-even pair conversion -> v_cvt_pk_bf16_f32 (lo in bits 0..15)
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    bf2_t v = __builtin_convertvector((f2_t){lo, hi}, bf2_t);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)(pack_bf2(x, 0.f) & 0xffffu); }

// ---- fp8 (OCP e4m3fn: the gfx950 format) ---------------------------------------
// Four floats -> four e4m3 bytes (byte 0 = a), round-to-nearest-even (v_cvt_pk_fp8_f32), clamped to the largest
// finite value first so that an out-of-range input saturates whatever the conversion's overflow mode is.
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;    // one scaled-MFMA 16x16x128 fp8 A/B fragment (8 VGPRs)
#define KZV_FP8_MAX 448.f
__device__ __forceinline__ float fp8_clamp(float x) { return __builtin_amdgcn_fmed3f(x, -KZV_FP8_MAX, KZV_FP8_MAX); }
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    unsigned r = 0;
    r = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(a), fp8_clamp(b), r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(c), fp8_clamp(d), r, true);
    return r;
}

// ---- wave64 reductions ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- LayerNorm backward: the per-lane part of one row (4 columns) -----------------------------------------------------------------
// Explicit roundings (no contraction left to the compiler's choice per context): ln_bwd_fast_kernel (layernorm.hip) and the decoder's
// backward segments (decoder_chain.hip) carry the same row arithmetic and must produce the same bits.
struct LnBwdTerms { float4 xh, gy; float s1, s2; };
__device__ __forceinline__ LnBwdTerms ln_bwd_terms(const float4 d, const float4 xv, const float4 gm, float mean, float rstd) {
    LnBwdTerms t;
    t.xh = make_float4(__fmul_rn(__fsub_rn(xv.x, mean), rstd), __fmul_rn(__fsub_rn(xv.y, mean), rstd), __fmul_rn(__fsub_rn(xv.z, mean), rstd), __fmul_rn(__fsub_rn(xv.w, mean), rstd));
    t.gy = make_float4(__fmul_rn(d.x, gm.x), __fmul_rn(d.y, gm.y), __fmul_rn(d.z, gm.z), __fmul_rn(d.w, gm.w));
    t.s1 = __fadd_rn(__fadd_rn(__fadd_rn(t.gy.x, t.gy.y), t.gy.z), t.gy.w);
    t.s2 = __fmaf_rn(t.gy.w, t.xh.w, __fmaf_rn(t.gy.z, t.xh.z, __fmaf_rn(t.gy.y, t.xh.y, __fmul_rn(t.gy.x, t.xh.x))));
    return t;
}
__device__ __forceinline__ void ln_bwd_accum(float4& dg, float4& db, const float4 d, const LnBwdTerms& t) {      // gamma / beta partial sums
    dg.x = __fmaf_rn(d.x, t.xh.x, dg.x); dg.y = __fmaf_rn(d.y, t.xh.y, dg.y); dg.z = __fmaf_rn(d.z, t.xh.z, dg.z); dg.w = __fmaf_rn(d.w, t.xh.w, dg.w);
    db.x = __fadd_rn(db.x, d.x); db.y = __fadd_rn(db.y, d.y); db.z = __fadd_rn(db.z, d.z); db.w = __fadd_rn(db.w, d.w);
}
__device__ __forceinline__ float4 ln_bwd_dx(const LnBwdTerms& t, float m1, float m2, float rstd) {              // rstd * (gy - m1 - xh * m2)
    return make_float4(__fmul_rn(rstd, __fmaf_rn(-t.xh.x, m2, __fsub_rn(t.gy.x, m1))), __fmul_rn(rstd, __fmaf_rn(-t.xh.y, m2, __fsub_rn(t.gy.y, m1))),
                       __fmul_rn(rstd, __fmaf_rn(-t.xh.z, m2, __fsub_rn(t.gy.z, m1))), __fmul_rn(rstd, __fmaf_rn(-t.xh.w, m2, __fsub_rn(t.gy.w, m1))));
}

// ---- counter-based dropout bits ------------------------------------------------
// One 32-bit hash per PAIR of elements; each element takes 16 bits and is KEPT iff bits >= thr16
// (thr16 = round(p * 65536)), so P(drop) = thr16/65536.  The same (key, pair index) is recomputed in
// backward, so no mask is stored.  key = kzv_drop_key(seed, site) mixes the step seed with a
// per-call-site id on the host.
__device__ __forceinline__ unsigned kzv_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_bits(unsigned key, unsigned pair_idx) {
    return kzv_hash32(pair_idx * 0x9E3779B9U + key);
}
// keep-multiplier for element `which` (0/1) of a pair
__device__ __forceinline__ float drop_keep(unsigned bits, int which, unsigned thr16, float inv_keep) {
    unsigned r = which ? (bits >> 16) : (bits & 0xffffu);
    return r >= thr16 ? inv_keep : 0.f;
}

// ---- attention-probability dropout: 4 x 4 blocks of one (batch, head)'s [Sq, Sk] matrix ---------------------------
// The forward kernel holds 4 consecutive KEYS of one query per lane, the backward kernel 4 consecutive QUERIES of one key,
// so the generator is built to cost the same in either orientation: ONE 32-bit hash per 4 x 4 block
//     x = mix(block * 0x9E3779B9 + key),   block = ((b * heads + h) * ceil(Sq / 4) + (q >> 2)) * ceil(Sk / 4) + (k >> 2)
// and per element (r = q & 3, c = k & 3) a 16-bit value from one 16-bit half of x (low half if r + c is even):
//     u = ((half ^ C[r][c]) * M[r][c]) mod 2^16        kept iff (int16) u >= thr16 - 32768      (P(drop) = thr16 / 65536)
// A lane computes its 4 elements as two packed pairs (v_pk_mul_lo_u16): ~3.5 VALU slots per element against ~6.5 for the
// 32-bit hash per key pair of the hidden-state sites.  C / M were picked (tools/dev/pick_attn_drop_consts.py) so that the
// joint drop rates of the 8 elements sharing a half stay within sampling noise of independence over all 2^16 halves;
// oracle/attn_dropout.py is the numpy statement, pinned bit-for-bit against kzv_debug_attn_dropout_mask on the GPU.
#define KZV_ATT_GOLD 0x9E3779B9U
__device__ __constant__ const unsigned short kzv_att_c[16] = {0xba79, 0x0e76, 0x9b89, 0x53b0, 0x431d, 0x0cc3, 0xa452, 0x4805,
                                                              0xd3bc, 0xd36a, 0x9c49, 0x9be5, 0xd12f, 0x8ff4, 0x38f5, 0x7f7a};
__device__ __constant__ const unsigned short kzv_att_m[16] = {0x5195, 0x0735, 0xf067, 0x26fb, 0xbafb, 0xee95, 0xe455, 0x0e9d,
                                                              0x1f77, 0xc189, 0x6fa9, 0x4599, 0x31b9, 0x473d, 0xf055, 0xa1a9};
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
__device__ __forceinline__ unsigned att_mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; return x; }
// scalar form (one element): the generic head-dim kernels and the debug / parity entry
__device__ __forceinline__ bool att_keep1(unsigned key, unsigned block, int r, int c, int thr16) {
    const unsigned x = att_mix(block * KZV_ATT_GOLD + key);
    const unsigned half = ((r + c) & 1) ? (x >> 16) : (x & 0xffffu);
    const unsigned u = ((half ^ kzv_att_c[r * 4 + c]) * kzv_att_m[r * 4 + c]) & 0xffffu;
    return (int)(short)u >= thr16 - 32768;
}
// packed form: per-lane constants for the lane's FIXED in-block index (forward: r = q & 3, elements run along c;
// backward: c = key & 3, elements run along r)
struct AttDropLane { unsigned c01, c23, m01, m23, rot; };
__device__ __forceinline__ AttDropLane att_drop_lane(int fixed, bool along_c) {
    unsigned cc[4], mm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = along_c ? fixed * 4 + j : j * 4 + fixed;
        cc[j] = kzv_att_c[idx]; mm[j] = kzv_att_m[idx];
    }
    AttDropLane d;
    d.c01 = cc[0] | (cc[1] << 16); d.c23 = cc[2] | (cc[3] << 16);
    d.m01 = mm[0] | (mm[1] << 16); d.m23 = mm[2] | (mm[3] << 16);
    d.rot = (fixed & 1) ? 16u : 0u;      // odd fixed index: element 0 takes the HIGH half of x
    return d;
}
// the lane's four 16-bit values as two packed pairs (elements 0,1 | 2,3) from the mixed block word
__device__ __forceinline__ void att_drop_u(const AttDropLane& d, unsigned xmixed, unsigned* u01, unsigned* u23) {
    const unsigned xr = __builtin_amdgcn_alignbit(xmixed, xmixed, d.rot);
    *u01 = __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, xr ^ d.c01) * __builtin_bit_cast(u16x2_t, d.m01));
    *u23 = __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2_t, xr ^ d.c23) * __builtin_bit_cast(u16x2_t, d.m23));
}
// 0xffff per kept element, 0 per dropped one; thrm1x2 = two copies of (int16)(thr16 - 32768 - 1)
__device__ __forceinline__ unsigned att_keep_mask(unsigned u2, unsigned thrm1x2) {
    const i16x2_t t = __builtin_elementwise_sub_sat(__builtin_bit_cast(i16x2_t, thrm1x2), __builtin_bit_cast(i16x2_t, u2));
    return __builtin_bit_cast(unsigned, t >> 15);
}

// erf-GELU and its derivative.  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7 before fp32 rounding):
// one v_rcp, one v_exp and 6 FMAs instead of libm's erff (~60 VALU ops), which made the GEMM epilogues
// VALU-bound.  The Gaussian the derivative needs is the same exponential (exp(-(x/sqrt2)^2) = exp(-x^2/2)).
__device__ __forceinline__ void kzv_erf_parts(float x, float* erf_out, float* gauss_out) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    const float e = __expf(-z * z);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float er = fmaf(-poly * t, e, 1.f);
    *erf_out = copysignf(er, x);
    *gauss_out = e;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float er, e;
    kzv_erf_parts(x, &er, &e);
    return 0.5f * x * (1.f + er);
}
// GELU and its derivative from ONE erf / exp evaluation (the forward epilogues store the derivative for backward)
__device__ __forceinline__ void gelu_erf_both(float x, float* y, float* dy) {
    float er, e;
    kzv_erf_parts(x, &er, &e);
    const float cdf = 0.5f * (1.f + er);
    *y = x * cdf;
    *dy = fmaf(x * 0.39894228040143268f, e, cdf);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float er, e;
    kzv_erf_parts(x, &er, &e);
    return fmaf(x * 0.39894228040143268f, e, 0.5f * (1.f + er));
}

// XCD-aware block remap (bijective for any grid size): blocks that share an XCD under round-robin
// dispatch (id % 8) get a contiguous chunk of logical tile ids, so neighbours share L2 lines.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
// (KZV_GLDS_NT, defined by a translation unit before this header: every LDS-DMA of that unit carries the streaming hint)
#ifdef KZV_GLDS_NT
#define KZV_GLDS_AUX 2
#define KZV_GLDS_SFX " nt"
#else
#define KZV_GLDS_AUX 0
#define KZV_GLDS_SFX ""
#endif
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const KZV_GLB void*)g, (KZV_LDS void*)lds_wave_base, 16, 0, KZV_GLDS_AUX);
}
// Same LDS-DMA issued from inline asm: invisible to hipcc's waitcnt pass, so the compiler does not put a
// vmcnt(0) in front of later LDS reads it cannot disambiguate (it does for ds_read_b64_tr_b16 after a
// builtin LDS-DMA, which serialises load and compute).  The CALLER owns the wait: s_waitcnt vmcnt(N), then
// a barrier, before any wave reads the bytes.  lds_dst must be wave-uniform (it is moved into M0).
__device__ __forceinline__ void glds16_asm(const void* g, void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)lds_wave_base));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" KZV_GLDS_SFX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
// Same, leaving M0 clobbered: only for kernels that issue EVERY LDS-DMA through this helper (no builtin LDS-DMA,
// whose M0 the compiler tracks), e.g. gemm_tn.
__device__ __forceinline__ void glds16_asm_m0(const void* g, void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)lds_wave_base));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" KZV_GLDS_SFX : : "v"(g), "s"(dst) : "memory");
}
// Same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (no 64-bit address registers per lane)
__device__ __forceinline__ void glds16_asm_soff(const void* sbase, unsigned voff, void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)lds_wave_base));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" KZV_GLDS_SFX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
// the SGPR-base form with M0 left clobbered (same condition as glds16_asm_m0)
__device__ __forceinline__ void glds16_asm_soff_m0(const void* sbase, unsigned voff, void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)((KZV_LDS char*)lds_wave_base));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" KZV_GLDS_SFX : : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
__device__ __forceinline__ bf16x4 lds_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((KZV_LDS bf16x4*)p);
}
