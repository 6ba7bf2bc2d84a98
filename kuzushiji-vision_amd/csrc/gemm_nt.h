// Shared between the two gemm_nt kernels (gemm.hip: 128x128 two-stage; gemm_nt256.hip: 256x256 eight-phase):
// launch parameters and the fused per-row epilogue.
#pragma once
#include "kzv_common.h"
#include "../../include/kzv.h"

struct NtParams {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; const float* resid; bf16_t* aux;
    const void* zero16;
    int64_t lda, ldb, ldc, ldr, ldaux;
    int M, N, K, n_valid;
    unsigned drop_thr16; float drop_inv_keep; unsigned drop_key;
    int strip;          // tile-walk strip width of the 256x256 kernels (nt_tile_coords)
    // fp8 operands (kzv_nt256p_fp8_launch only; A and B are e4m3 bytes, lda / ldb in elements = bytes):
    // C = (acc * a_scale[m] * b_scale[n]) + bias, then the epilogue as usual
    const float* a_scale; const float* b_scale;
    // GELU epilogue of the fp8 kernel: a second copy of C as e4m3 (the next GEMM's A operand), quantised with the
    // per-tensor multiplier *c8_qscale; the largest |C| seen is folded into *c8_amax (next step's multiplier)
    unsigned char* c8; int64_t ldc8; const float* c8_qscale; float* c8_amax;
    // DGELU epilogue of the fp8 kernel: the e4m3 copy is quantised per ROW with the multipliers c8_rowq[m] (no amax tracking)
    const float* c8_rowq;
};

// One thread finishes 4 consecutive columns n0..n0+3 of output row m: v = accumulator + bias on entry;
// r4 = the residual (RESID), u2 = the saved GELU derivative as 4 bf16 (DGELU), both loaded by the caller.
// Outputs are written once and next read by a later kernel: stream them past L2 (global_store ... nt) so the
// operand panels stay cached and no dirty backlog waits for the end-of-kernel write-back (-3..4 % on the ViT GEMMs).
// Re-tested per epilogue in round 4 (tools/dev/r5_stores.sh, same-box alternations): plain stores for the fc1 activation -0.5 % step, for
// the bf16 outputs -0.1 %, for the DGELU output +-0; for the RESID output (the residual stream) +0.3 %: that one is stored plainly.
#ifndef KZV_NT_PLAIN_STORES
__device__ __forceinline__ void nt_st(uint2* dst, const uint2& v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store((u32x2){v.x, v.y}, (u32x2*)dst);
}
__device__ __forceinline__ void nt_st(float4* dst, const float4& v) { __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, (f32x4*)dst); }
#else
template <typename T> __device__ __forceinline__ void nt_st(T* dst, const T& v) { *dst = v; }
#endif

template <int EPI>
__device__ __forceinline__ void nt_emit(const NtParams& p, int m, int n0, float (&v)[4], const float4& r4, const uint2& u2) {
    if (EPI == KZV_EPI_BF16) {
        nt_st((uint2*)((bf16_t*)p.C + (int64_t)m * p.ldc + n0), make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])));
    } else if (EPI == KZV_EPI_F32) {
        nt_st((float4*)((float*)p.C + (int64_t)m * p.ldc + n0), make_float4(v[0], v[1], v[2], v[3]));
    } else if (EPI == KZV_EPI_GELU || EPI == KZV_EPI_GELU_F32) {
        // aux = gelu'(pre-activation): the erf / exp the activation needs give the derivative for three more FMAs, and the
        // backward epilogue (DGELU) becomes a plain multiply instead of a second erf + exp per element
        float y[4], d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gelu_erf_both(v[r], &y[r], &d[r]);
        nt_st((uint2*)(p.aux + (int64_t)m * p.ldaux + n0), make_uint2(pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])));
        if (EPI == KZV_EPI_GELU) nt_st((uint2*)((bf16_t*)p.C + (int64_t)m * p.ldc + n0), make_uint2(pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])));
        else nt_st((float4*)((float*)p.C + (int64_t)m * p.ldc + n0), make_float4(y[0], y[1], y[2], y[3]));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = y[r];          // the fp8 kernel quantises the activation once more from here
    } else if (EPI == KZV_EPI_RESID) {
        if (p.drop_thr16) {
            const unsigned e = (unsigned)m * (unsigned)p.N + (unsigned)n0;
            const unsigned b0 = drop_bits(p.drop_key, e >> 1), b1 = drop_bits(p.drop_key, (e >> 1) + 1);
            v[0] *= drop_keep(b0, 0, p.drop_thr16, p.drop_inv_keep);
            v[1] *= drop_keep(b0, 1, p.drop_thr16, p.drop_inv_keep);
            v[2] *= drop_keep(b1, 0, p.drop_thr16, p.drop_inv_keep);
            v[3] *= drop_keep(b1, 1, p.drop_thr16, p.drop_inv_keep);
        }
#ifndef KZV_RESID_NT       // the residual stream is read again at once (the LayerNorm behind this GEMM, then the next residual add): a PLAIN store leaves it in
                           // L2 / the Infinity Cache: step 31.02 -> 30.93 ms, family 0.386 -> 0.388 over four same-box alternations (round 4, tools/dev/ab_build.sh)
        *(float4*)((float*)p.C + (int64_t)m * p.ldc + n0) = make_float4(v[0] + r4.x, v[1] + r4.y, v[2] + r4.z, v[3] + r4.w);
#else
        nt_st((float4*)((float*)p.C + (int64_t)m * p.ldc + n0), make_float4(v[0] + r4.x, v[1] + r4.y, v[2] + r4.z, v[3] + r4.w));
#endif
    } else if (EPI == KZV_EPI_DGELU) {
        v[0] *= bf2f((bf16_t)(u2.x & 0xffff));          // u2 = gelu'(pre-activation), stored by the forward GELU epilogue
        v[1] *= bf2f((bf16_t)(u2.x >> 16));
        v[2] *= bf2f((bf16_t)(u2.y & 0xffff));
        v[3] *= bf2f((bf16_t)(u2.y >> 16));
        nt_st((uint2*)((bf16_t*)p.C + (int64_t)m * p.ldc + n0), make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])));
    }
}

// Tile walk of the 256x256 kernels: column STRIPS of `strip` tile columns, row-major inside a strip.  The 32 CUs of an XCD
// run 32 consecutive tile ids at a time; in plain row-major order those span every B (weight) column panel of a wide
// output (N = 2304: 9 panels = 3.5 MB, N = 3072: 4.7 MB next to the A row panels in a 4 MiB L2 -> the weights are
// re-streamed from the Infinity Cache every round); inside a strip they span `strip` panels (3 x 393 KB at K = 768)
// and ~32/strip A row panels, which stream through once per strip.  strip <= 0 or >= tilesN: plain row-major.
__device__ __forceinline__ void nt_tile_coords(int id, int tilesM, int tilesN, int strip, int& tm, int& tn) {
    if (strip <= 0 || strip >= tilesN) { tm = id / tilesN; tn = id - tm * tilesN; return; }
    const int per = tilesM * strip;
    const int st = id / per;
    const int rem = id - st * per;
    const int w = min(strip, tilesN - st * strip);       // the last strip may be narrower (its ids are the tail of the range)
    tm = rem / w;
    tn = st * strip + (rem - tm * w);
}
int kzv_nt_strip();      // KZV_NT_STRIP (default 3)

// gemm_nt256.hip: returns 1 when it took the launch, 0 when the shape is left to the 128x128 kernel.
int kzv_nt256_launch(const NtParams& p, int epilogue, hipStream_t s);
// gemm_nt256p.hip (persistent variant of the same schedule): same contract.
int kzv_nt256p_launch(const NtParams& p, int epilogue, hipStream_t s);
// gemm_nt256f.hip (free-running schedule: two barriers per K-tile, loads hidden under each wave's own MFMAs): same contract.
int kzv_nt256f_launch(const NtParams& p, int epilogue, hipStream_t s);
// gemm_nt256h.hip (four waves, 256x128 tile, two workgroups per CU out of phase, free-running K loop; K % 384 == 0): same contract.
int kzv_nt256h_launch(const NtParams& p, int epilogue, hipStream_t s);
// fp8 (e4m3) operands on the block-scaled MFMA, same persistent schedule (gemm_nt256p.hip); epilogues BF16, GELU, RESID, DGELU.
// Returns KZV_OK or an error: there is no other fp8 kernel to fall back to.
int kzv_nt256p_fp8_launch(const NtParams& p, int epilogue, hipStream_t s);
// gemm_rows.hip (few rows: one wave per 16x64 tile, operands straight from L2): same contract.  Taken inside a KzvRowsScope
// (the generation step, model.cpp) for M <= 4096, elsewhere only below kzv_set_rows_max_m (default 0: never), so that the
// training step and its parity tests keep running the tiled kernels at every size.
int kzv_rows_launch(const NtParams& p, int epilogue, hipStream_t s);
struct KzvRowsScope { KzvRowsScope(); ~KzvRowsScope(); };
// the few-rows kernel with the decoder's LayerNorm folded into its A operand and / or its residual (gemm_rows.hip)
int kzv_rows_ln_launch(const NtParams& p, int epilogue, const float* ln_a, const float* ga, const float* ba, const float* ln_r, const float* gr,
                       const float* br, float eps, hipStream_t s);
