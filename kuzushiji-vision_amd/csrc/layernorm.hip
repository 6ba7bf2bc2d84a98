// LayerNorm forward / backward (fp32 statistics, eps inside the rsqrt, one wave64 per row).
// Replaces nn.LayerNorm(eps=1e-12) at: ViTLayer.layernorm_before/after (HF modeling_vit.py:261-262),
// ViTEncoder.layernorm (src/models/trocr_model.py:152,197), RoBERTa post-LNs
// (HF modeling_roberta.py:333,339,391,397), embeddings LN (:64,120), LM-head LN (:883,890).
//
// HBM-bound: forward reads x once (row kept in registers), writes bf16 (+ optional fp32);
// backward reads dy + x once, writes dx once; gamma/beta gradients are reduced per workgroup in
// registers/LDS and leave with one float atomic per column per workgroup into one of LN_SLOTS partial rows
// (644 workgroups adding to the SAME 1.5k addresses serialised on them: 122 -> 76 us at [41216, 768]), which a
// tiny second kernel folds into dgamma / dbeta.
//
// Row remap (`seq`, `drop_first`): the encoder's final LN output drops the CLS token
// (trocr_model.py:200), so output row = row - row/seq - 1 and rows with row % seq == 0 are skipped.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "kzv_kernels.h"
#include <mutex>
#include <vector>
#include <algorithm>

#define KZV_LN_TRY(expr) do { int rc__ = (expr); if (rc__ != KZV_OK) return rc__; } while (0)
namespace {

constexpr int MAXC = 8;  // float4 chunks per lane -> H <= 2048 (kernels are instantiated per chunk count: registers = occupancy)

struct LnFwd {
    const float* x; const float* gamma; const float* beta; bf16_t* y16; float* y32; float* stats;
    int rows, H, seq, drop_first; float eps;
    unsigned thr16; float inv_keep; unsigned key;
    unsigned char* y8; float* y8_scale;      // F8: e4m3 copy of the output row, quantised by its own largest |y| (y ~ y8 * y8_scale[row])
};

template <int NC, bool F8>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwd p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int nc = p.H >> 2;
    const float4* xr = (const float4*)(p.x + (int64_t)row * p.H);
    float4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) { v[c] = xr[i]; s += v[c].x + v[c].y + v[c].z + v[c].w; }
    }
    const float mean = wave_sum(s) / (float)p.H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
            q += a * a + b * b + cc * cc + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)p.H + p.eps);
    if (p.stats && lane == 0) { p.stats[2 * row] = mean; p.stats[2 * row + 1] = rstd; }
    int orow = row;
    if (p.drop_first) {
        if (row % p.seq == 0) return;
        orow = row - row / p.seq - 1;
    }
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            const float4 gm = ((const float4*)p.gamma)[i], bt = ((const float4*)p.beta)[i];
            float o0 = (v[c].x - mean) * rstd * gm.x + bt.x, o1 = (v[c].y - mean) * rstd * gm.y + bt.y;
            float o2 = (v[c].z - mean) * rstd * gm.z + bt.z, o3 = (v[c].w - mean) * rstd * gm.w + bt.w;
            if (p.thr16) {
                const unsigned e = (unsigned)orow * (unsigned)p.H + 4u * i;
                const unsigned b0 = drop_bits(p.key, e >> 1), b1 = drop_bits(p.key, (e >> 1) + 1);
                o0 *= drop_keep(b0, 0, p.thr16, p.inv_keep); o1 *= drop_keep(b0, 1, p.thr16, p.inv_keep);
                o2 *= drop_keep(b1, 0, p.thr16, p.inv_keep); o3 *= drop_keep(b1, 1, p.thr16, p.inv_keep);
            }
            if (p.y16) ((uint2*)(p.y16 + (int64_t)orow * p.H))[i] = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
            if (p.y32) ((float4*)(p.y32 + (int64_t)orow * p.H))[i] = make_float4(o0, o1, o2, o3);
            if (F8) {
                v[c] = make_float4(o0, o1, o2, o3);
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3))));
            }
        }
    }
    if (F8) {
        amax = wave_max(amax);
        const float qs = amax > 0.f ? KZV_FP8_MAX / amax : 1.f;
        if (lane == 0) p.y8_scale[orow] = amax > 0.f ? amax / KZV_FP8_MAX : 1.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = lane + c * 64;
            if (i < nc) ((unsigned*)(p.y8 + (int64_t)orow * p.H))[i] = pack_fp8x4(v[c].x * qs, v[c].y * qs, v[c].z * qs, v[c].w * qs);
        }
    }
}

struct LnBwd {
    const void* dy; const float* x; const float* stats; const float* gamma;
    float* dx; float* dgamma; float* dbeta;
    int rows, H, seq, drop_first, dy_f32, accumulate;
    unsigned thr16; float inv_keep; unsigned key;
    bf16_t* out16; unsigned o_thr16; float o_inv_keep; unsigned o_key;   // optional: bf16 copy of the TOTAL dx, dropout-masked
    float* partial;                      // [LN_SLOTS][2][H] gamma/beta partial sums (zero on entry, zeroed again by the reduce kernel)
    // fp8 input-gradient path (fast kernel only): beside out16, an e4m3 copy of the masked dx row quantised by its own amax
    // (out8 ~ row / out8_scale[row]) and the multiplier the NEXT gradient tensor of this row will be quantised with, from the
    // bound |dy . w| <= ||dy|| ||w||: out8_rq[row] = 448 / (KZV_F8_BOUND ||row||_2 *rq_wnorm), out8_rqinv = its inverse
    unsigned char* out8; float* out8_scale; float* out8_rq; float* out8_rqinv; const float* rq_wnorm;
};
#define KZV_F8_BOUND 1.4125f             // max gelu' (1.13) x 1.25 for what e4m3 rounding adds to the two norms

constexpr int LN_SLOTS = 32;

constexpr int BWD_ROWS = 64;  // rows per workgroup (4 waves x 16)

template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 waves][2][H] floats
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nc = p.H >> 2;
    float4 dg[NC], db[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { dg[c] = make_float4(0, 0, 0, 0); db[c] = make_float4(0, 0, 0, 0); }
    const float invH = 1.f / (float)p.H;
    float4 gmr[NC];                      // gamma: once per workgroup, not once per row
#pragma unroll
    for (int c = 0; c < NC; ++c) gmr[c] = (lane + c * 64 < nc) ? ((const float4*)p.gamma)[lane + c * 64] : make_float4(0, 0, 0, 0);
#pragma unroll 2
    for (int rr = 0; rr < BWD_ROWS / 4; ++rr) {
        const int row = blockIdx.x * BWD_ROWS + rr * 4 + w;
        if (row >= p.rows) break;
        int drow = row;
        bool has_dy = true;
        if (p.drop_first) {
            if (row % p.seq == 0) has_dy = false;
            drow = row - row / p.seq - 1;
        }
        const float mean = p.stats[2 * row], rstd = p.stats[2 * row + 1];
        const float4* xr = (const float4*)(p.x + (int64_t)row * p.H);
        float4 xh[NC], gy[NC], prev[NC];
        float s1 = 0.f, s2 = 0.f;
        float4* dxr = (float4*)(p.dx + (int64_t)row * p.H);
#pragma unroll
        for (int c = 0; c < NC; ++c) {          // issue the accumulate-into loads with the others (latency overlap)
            const int i = lane + c * 64;
            prev[c] = (p.accumulate && i < nc) ? dxr[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = lane + c * 64;
            if (i < nc) {
                const float4 xv = xr[i];
                float4 d = make_float4(0, 0, 0, 0);
                if (has_dy) {
                    if (p.dy_f32) d = ((const float4*)((const float*)p.dy + (int64_t)drow * p.H))[i];
                    else {
                        const uint2 u = ((const uint2*)((const bf16_t*)p.dy + (int64_t)drow * p.H))[i];
                        d = make_float4(bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16));
                    }
                    if (p.thr16) {
                        const unsigned e = (unsigned)drow * (unsigned)p.H + 4u * i;
                        const unsigned b0 = drop_bits(p.key, e >> 1), b1 = drop_bits(p.key, (e >> 1) + 1);
                        d.x *= drop_keep(b0, 0, p.thr16, p.inv_keep); d.y *= drop_keep(b0, 1, p.thr16, p.inv_keep);
                        d.z *= drop_keep(b1, 0, p.thr16, p.inv_keep); d.w *= drop_keep(b1, 1, p.thr16, p.inv_keep);
                    }
                }
                const float4 gm = gmr[c];
                xh[c] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                gy[c] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
                s1 += gy[c].x + gy[c].y + gy[c].z + gy[c].w;
                s2 += gy[c].x * xh[c].x + gy[c].y * xh[c].y + gy[c].z * xh[c].z + gy[c].w * xh[c].w;
                dg[c].x += d.x * xh[c].x; dg[c].y += d.y * xh[c].y; dg[c].z += d.z * xh[c].z; dg[c].w += d.w * xh[c].w;
                db[c].x += d.x; db[c].y += d.y; db[c].z += d.z; db[c].w += d.w;
            }
        }
        const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = lane + c * 64;
            if (i < nc) {
                float4 o = make_float4(rstd * (gy[c].x - m1 - xh[c].x * m2), rstd * (gy[c].y - m1 - xh[c].y * m2),
                                       rstd * (gy[c].z - m1 - xh[c].z * m2), rstd * (gy[c].w - m1 - xh[c].w * m2));
                o.x += prev[c].x; o.y += prev[c].y; o.z += prev[c].z; o.w += prev[c].w;
                dxr[i] = o;
                if (p.out16) {   // input of the next "dropout(linear)" backward: mask(dx) in bf16 (same indexing as colsum_kernel)
                    if (p.o_thr16) {
                        const unsigned e = (unsigned)row * (unsigned)p.H + 4u * i;
                        const unsigned b0 = drop_bits(p.o_key, e >> 1), b1 = drop_bits(p.o_key, (e >> 1) + 1);
                        o.x *= drop_keep(b0, 0, p.o_thr16, p.o_inv_keep); o.y *= drop_keep(b0, 1, p.o_thr16, p.o_inv_keep);
                        o.z *= drop_keep(b1, 0, p.o_thr16, p.o_inv_keep); o.w *= drop_keep(b1, 1, p.o_thr16, p.o_inv_keep);
                    }
                    ((uint2*)(p.out16 + (int64_t)row * p.H))[i] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
                }
            }
        }
    }
    // workgroup reduction of the gamma/beta partials, then one atomic per column
    float4* sg = (float4*)smem + (w * 2) * nc;
    float4* sb = sg + nc;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) { sg[i] = dg[c]; sb[i] = db[c]; }
    }
    __syncthreads();
    const float* sf = (const float*)smem;
    for (int col = threadIdx.x; col < p.H; col += 256) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { a += sf[(ww * 2) * p.H + col]; b += sf[(ww * 2 + 1) * p.H + col]; }
        float* slot = p.partial + (size_t)(blockIdx.x % LN_SLOTS) * 2 * p.H;
        atomicAdd(slot + col, a);
        atomicAdd(slot + p.H + col, b);
    }
}

// Pipelined variant for H = NC*256 exactly (no lane guards), dy type and accumulate mode fixed at compile time: the
// row loop is straight-line, so hipcc counts its vmcnt waits, and the loads of row r+1 are in flight while row r is
// reduced and stored (two rows of latency overlap per wave instead of one).
template <int NC, bool DYF32, bool ACC, bool F8>
__global__ __launch_bounds__(256) void ln_bwd_fast_kernel(const LnBwd p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 waves][2][H] floats
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int nc = NC * 64;
    float4 dg[NC], db[NC], gmr[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        dg[c] = make_float4(0, 0, 0, 0); db[c] = make_float4(0, 0, 0, 0);
        gmr[c] = ((const float4*)p.gamma)[lane + c * 64];
    }
    const float invH = 1.f / (float)p.H;
    const int row_base = blockIdx.x * BWD_ROWS + w;               // this wave: rows row_base + 4*rr
    const int n = min(BWD_ROWS / 4, (p.rows - row_base + 3) / 4); // valid rows of this wave (wave-uniform, may be <= 0)

    struct Buf { float4 x[NC], prev[NC]; float4 d32[NC]; uint2 d16[NC]; float mean, rstd; };
    auto load = [&](Buf& b, int rr) {
        const int row = row_base + 4 * rr;
        const int drow = p.drop_first ? max(row - row / p.seq - 1, 0) : row;
        const float4* xr = (const float4*)(p.x + (int64_t)row * p.H);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = lane + c * 64;
            b.x[c] = xr[i];
            if (ACC) b.prev[c] = ((const float4*)(p.dx + (int64_t)row * p.H))[i];
            if (DYF32) b.d32[c] = ((const float4*)((const float*)p.dy + (int64_t)drow * p.H))[i];
            else b.d16[c] = ((const uint2*)((const bf16_t*)p.dy + (int64_t)drow * p.H))[i];
        }
        b.mean = p.stats[2 * row]; b.rstd = p.stats[2 * row + 1];
    };
    auto compute = [&](const Buf& b, int rr) {
        const int row = row_base + 4 * rr;
        const bool has_dy = !(p.drop_first && row % p.seq == 0);
        const int drow = p.drop_first ? max(row - row / p.seq - 1, 0) : row;
        const float mean = b.mean, rstd = b.rstd;
        LnBwdTerms tm[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = lane + c * 64;
            float4 d;
            if (DYF32) d = b.d32[c];
            else d = make_float4(bf2f(b.d16[c].x & 0xffff), bf2f(b.d16[c].x >> 16), bf2f(b.d16[c].y & 0xffff), bf2f(b.d16[c].y >> 16));
            if (!has_dy) d = make_float4(0, 0, 0, 0);
            if (p.thr16) {
                const unsigned e = (unsigned)drow * (unsigned)p.H + 4u * i;
                const unsigned b0 = drop_bits(p.key, e >> 1), b1 = drop_bits(p.key, (e >> 1) + 1);
                d.x *= drop_keep(b0, 0, p.thr16, p.inv_keep); d.y *= drop_keep(b0, 1, p.thr16, p.inv_keep);
                d.z *= drop_keep(b1, 0, p.thr16, p.inv_keep); d.w *= drop_keep(b1, 1, p.thr16, p.inv_keep);
            }
            tm[c] = ln_bwd_terms(d, b.x[c], gmr[c], mean, rstd);      // kzv_common.h: the row arithmetic with explicit roundings
            s1 = __fadd_rn(s1, tm[c].s1); s2 = __fadd_rn(s2, tm[c].s2);
            ln_bwd_accum(dg[c], db[c], d, tm[c]);
        }
        const float m1 = __fmul_rn(wave_sum(s1), invH), m2 = __fmul_rn(wave_sum(s2), invH);
        float4* dxr = (float4*)(p.dx + (int64_t)row * p.H);
        float4 om[F8 ? NC : 1];
        float amax = 0.f, ssq = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = lane + c * 64;
            float4 o = ln_bwd_dx(tm[c], m1, m2, rstd);
            if (ACC) { o.x += b.prev[c].x; o.y += b.prev[c].y; o.z += b.prev[c].z; o.w += b.prev[c].w; }
            dxr[i] = o;
            if (p.out16) {
                if (p.o_thr16) {
                    const unsigned e = (unsigned)row * (unsigned)p.H + 4u * i;
                    const unsigned b0 = drop_bits(p.o_key, e >> 1), b1 = drop_bits(p.o_key, (e >> 1) + 1);
                    o.x *= drop_keep(b0, 0, p.o_thr16, p.o_inv_keep); o.y *= drop_keep(b0, 1, p.o_thr16, p.o_inv_keep);
                    o.z *= drop_keep(b1, 0, p.o_thr16, p.o_inv_keep); o.w *= drop_keep(b1, 1, p.o_thr16, p.o_inv_keep);
                }
                ((uint2*)(p.out16 + (int64_t)row * p.H))[i] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
                if (F8) {
                    om[c] = o;
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
                    ssq += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
                }
            }
        }
        if (F8) {
            amax = wave_max(amax);
            ssq = wave_sum(ssq);
            const float qs = amax > 0.f ? KZV_FP8_MAX / amax : 1.f;
            if (lane == 0) {
                p.out8_scale[row] = amax > 0.f ? amax / KZV_FP8_MAX : 1.f;
                const float bound = KZV_F8_BOUND * sqrtf(ssq) * *p.rq_wnorm;
                const float rq = bound > 0.f ? KZV_FP8_MAX / bound : 1.f;
                p.out8_rq[row] = rq; p.out8_rqinv[row] = 1.f / rq;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
                ((unsigned*)(p.out8 + (int64_t)row * p.H))[lane + c * 64] = pack_fp8x4(om[c].x * qs, om[c].y * qs, om[c].z * qs, om[c].w * qs);
        }
    };
    if (n > 0) {
        Buf b0, b1;
        load(b0, 0);
        for (int rr = 0; rr < n; rr += 2) {
            load(b1, min(rr + 1, n - 1));        // past the end: re-load the last row (keeps the loop body branch-free)
            compute(b0, rr);
            load(b0, min(rr + 2, n - 1));
            if (rr + 1 < n) compute(b1, rr + 1);
        }
    }
    float4* sg = (float4*)smem + (w * 2) * nc;
    float4* sb = sg + nc;
#pragma unroll
    for (int c = 0; c < NC; ++c) { sg[lane + c * 64] = dg[c]; sb[lane + c * 64] = db[c]; }
    __syncthreads();
    const float* sf = (const float*)smem;
    float* slot = p.partial + (size_t)(blockIdx.x % LN_SLOTS) * 2 * p.H;
    for (int col = threadIdx.x; col < p.H; col += 256) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { a += sf[(ww * 2) * p.H + col]; b += sf[(ww * 2 + 1) * p.H + col]; }
        atomicAdd(slot + col, a);
        atomicAdd(slot + p.H + col, b);
    }
}

// dgamma / dbeta += sum over the LN_SLOTS partial rows; the partials are left zeroed for the next call.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(float* partial, float* dgamma, float* dbeta, int H) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= 2 * H) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < LN_SLOTS; ++k) { s += partial[(size_t)k * 2 * H + col]; partial[(size_t)k * 2 * H + col] = 0.f; }
    float* out = col < H ? dgamma + col : dbeta + (col - H);
    *out += s;
}

// LN_REGIONS zeroed [LN_SLOTS][2][2048] partial-sum regions per process (calls are stream-ordered).  Region 0 serves a call whose
// fold follows at once; regions 1.. serve the calls of a KzvLnDeferScope (a whole backward pass), whose folds are ONE launch at the
// end of the scope: a fold is 4.9 us of launch latency for 6 KB of work, 45 times per training step.
constexpr int LN_REGIONS = 64;
constexpr size_t LN_REGION_FLOATS = (size_t)LN_SLOTS * 2 * MAXC * 256;
float* ln_partials() {
    static float* buf = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        void* q = nullptr;
        const size_t bytes = LN_REGIONS * LN_REGION_FLOATS * sizeof(float);
        if (hipMalloc(&q, bytes) == hipSuccess && hipMemset(q, 0, bytes) == hipSuccess) buf = (float*)q;
    });
    return buf;
}

struct LnFold { float* partial; float* dgamma; float* dbeta; int H; };
struct LnFoldTable { LnFold e[LN_REGIONS - 1]; };
// the folds of a scope in one launch: blockIdx.y = entry
__global__ __launch_bounds__(256) void ln_bwd_reduce_multi_kernel(const LnFoldTable t) {
    const LnFold& e = t.e[blockIdx.y];
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= 2 * e.H) return;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < LN_SLOTS; ++k) { s += e.partial[(size_t)k * 2 * e.H + col]; e.partial[(size_t)k * 2 * e.H + col] = 0.f; }
    float* out = col < e.H ? e.dgamma + col : e.dbeta + (col - e.H);
    *out += s;
}
int g_ln_defer = 0;
std::vector<LnFold> g_ln_pending;
int ln_flush(hipStream_t s) {
    if (g_ln_pending.empty()) return KZV_OK;
    LnFoldTable t;
    int hmax = 0;
    for (size_t i = 0; i < g_ln_pending.size(); ++i) { t.e[i] = g_ln_pending[i]; hmax = std::max(hmax, g_ln_pending[i].H); }
    hipLaunchKernelGGL(ln_bwd_reduce_multi_kernel, dim3((2 * hmax + 255) / 256, (unsigned)g_ln_pending.size()), dim3(256), 0, s, t);
    g_ln_pending.clear();
    return kzv_check_launch("layernorm_bwd_fold");
}

}  // namespace

int kzv_ln_fwd_ex(const float* x, const float* gamma, const float* beta, void* y16, float* y32, float* stats,
                  int rows, int H, float eps, int seq, int drop_first, float drop_p, uint32_t drop_key, hipStream_t s,
                  void* y8, float* y8_scale) {
    if (!x || !gamma || !beta || rows <= 0) return kzv_fail(KZV_E_ARG, "layernorm_fwd: null/empty");
    if (H % 4 || H > MAXC * 256) return kzv_fail(KZV_E_ARG, "layernorm: H must be a multiple of 4 and <= 2048");
    if ((y8 != nullptr) != (y8_scale != nullptr)) return kzv_fail(KZV_E_ARG, "layernorm_fwd: the fp8 copy needs both y8 and y8_scale");
    LnFwd p{x, gamma, beta, (bf16_t*)y16, y32, stats, rows, H, seq > 0 ? seq : 1, drop_first, eps, 0, 1.f, drop_key,
            (unsigned char*)y8, y8_scale};
    kzv_drop_params(drop_p, &p.thr16, &p.inv_keep);
    const int ncl = (H / 4 + 63) / 64;
    const dim3 grid((rows + 3) / 4);
#define KZV_LN_FWD(NC) do { if (y8) hipLaunchKernelGGL((ln_fwd_kernel<NC, true>), grid, dim3(256), 0, s, p);          \
                            else hipLaunchKernelGGL((ln_fwd_kernel<NC, false>), grid, dim3(256), 0, s, p); } while (0)
    if (ncl <= 1) KZV_LN_FWD(1);
    else if (ncl == 2) KZV_LN_FWD(2);
    else if (ncl == 3) KZV_LN_FWD(3);
    else if (ncl == 4) KZV_LN_FWD(4);
    else KZV_LN_FWD(8);
#undef KZV_LN_FWD
    return kzv_check_launch("layernorm_fwd");
}

int kzv_ln_bwd_ex(const void* dy, int dy_is_f32, const float* x, const float* stats, const float* gamma, float* dx,
                  int accumulate_dx, float* dgamma, float* dbeta, int rows, int H, int seq, int drop_first,
                  float drop_p, uint32_t drop_key, hipStream_t s, bf16_t* out16, float out_drop_p, uint32_t out_drop_key, const KzvLnBwdF8* f8) {
    if (!dy || !x || !stats || !gamma || !dx || !dgamma || !dbeta || rows <= 0) return kzv_fail(KZV_E_ARG, "layernorm_bwd: null/empty");
    if (H % 4 || H > MAXC * 256) return kzv_fail(KZV_E_ARG, "layernorm: H must be a multiple of 4 and <= 2048");
    LnBwd p{dy, x, stats, gamma, dx, dgamma, dbeta, rows, H, seq > 0 ? seq : 1, drop_first, dy_is_f32, accumulate_dx, 0, 1.f, drop_key,
            out16, 0, 1.f, out_drop_key, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (f8) {
        if (!out16 || !f8->out8 || !f8->scale || !f8->rq || !f8->rqinv || !f8->wnorm) return kzv_fail(KZV_E_ARG, "layernorm_bwd: the fp8 copy needs out16 and all of its arrays");
        if (H % 256 || H > 1024) return kzv_fail(KZV_E_ARG, "layernorm_bwd: the fp8 copy needs H = 256, 512, 768 or 1024");
        p.out8 = (unsigned char*)f8->out8; p.out8_scale = f8->scale; p.out8_rq = f8->rq; p.out8_rqinv = f8->rqinv; p.rq_wnorm = f8->wnorm;
    }
    kzv_drop_params(drop_p, &p.thr16, &p.inv_keep);
    kzv_drop_params(out_drop_p, &p.o_thr16, &p.o_inv_keep);
    const int ncl = (H / 4 + 63) / 64;
    const dim3 grid((rows + BWD_ROWS - 1) / BWD_ROWS);
    const size_t lds = 8 * H * sizeof(float);
    p.partial = ln_partials();
    if (!p.partial) return kzv_fail(KZV_E_HIP, "layernorm_bwd: partial-sum buffer unavailable");
    const bool deferred = g_ln_defer > 0;
    if (deferred) {                                  // a region of its own until the scope's fold
        bool same_out = false;                       // the single fold launch adds every entry with a plain `*out += s`: a LayerNorm that
        for (const LnFold& e : g_ln_pending) same_out |= e.dgamma == dgamma || e.dbeta == dbeta;      // appears twice in a scope is folded first
        if ((int)g_ln_pending.size() == LN_REGIONS - 1 || same_out) KZV_LN_TRY(ln_flush(s));
        p.partial += (1 + g_ln_pending.size()) * LN_REGION_FLOATS;
    }
    const bool fast = H == ncl * 256 && ncl <= 4;
#define KZV_LN_FAST2(NC, A, B)                                                                                 \
    do { if (f8) hipLaunchKernelGGL((ln_bwd_fast_kernel<NC, A, B, true>), grid, dim3(256), lds, s, p);          \
         else hipLaunchKernelGGL((ln_bwd_fast_kernel<NC, A, B, false>), grid, dim3(256), lds, s, p); } while (0)
#define KZV_LN_FAST(NC)                                                                                       \
    do {                                                                                                      \
        if (dy_is_f32) { if (accumulate_dx) KZV_LN_FAST2(NC, true, true); else KZV_LN_FAST2(NC, true, false); }  \
        else { if (accumulate_dx) KZV_LN_FAST2(NC, false, true); else KZV_LN_FAST2(NC, false, false); }          \
    } while (0)
    if (fast && ncl == 1) KZV_LN_FAST(1);
    else if (fast && ncl == 2) KZV_LN_FAST(2);
    else if (fast && ncl == 3) KZV_LN_FAST(3);
    else if (fast && ncl == 4) KZV_LN_FAST(4);
    else if (ncl <= 1) hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, dim3(256), lds, s, p);
    else if (ncl == 2) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, dim3(256), lds, s, p);
    else if (ncl == 3) hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, dim3(256), lds, s, p);
    else if (ncl == 4) hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, dim3(256), lds, s, p);
#undef KZV_LN_FAST
#undef KZV_LN_FAST2
    if (deferred) { g_ln_pending.push_back(LnFold{p.partial, dgamma, dbeta, H}); return kzv_check_launch("layernorm_bwd"); }
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * H + 255) / 256), dim3(256), 0, s, p.partial, dgamma, dbeta, H);
    return kzv_check_launch("layernorm_bwd");
}

// A kernel that fuses a LayerNorm backward (decoder_chain.hip's backward segments) accumulates its gamma / beta partial sums the way
// ln_bwd_fast_kernel does (one float atomic per column and workgroup into slot blockIdx % LN_SLOTS of a zeroed [LN_SLOTS][2][H]
// region): this hands out the region, with kzv_ln_bwd_ex's bookkeeping.  Inside a KzvLnDeferScope the fold joins the scope's single
// launch (*fold_now = false); otherwise the caller launches its kernel and then calls kzv_ln_partial_fold.
static_assert(LN_SLOTS == KZV_LN_SLOTS, "kzv_kernels.h: KZV_LN_SLOTS");
float* kzv_ln_partial_region(float* dgamma, float* dbeta, int H, hipStream_t s, bool* fold_now) {
    float* partial = ln_partials();
    if (!partial || !dgamma || !dbeta || H % 4 || H > MAXC * 256) { (void)kzv_fail(KZV_E_ARG, "layernorm partial region: bad argument or no buffer"); return nullptr; }
    *fold_now = g_ln_defer <= 0;
    if (g_ln_defer > 0) {
        bool same_out = false;
        for (const LnFold& e : g_ln_pending) same_out |= e.dgamma == dgamma || e.dbeta == dbeta;
        if (((int)g_ln_pending.size() == LN_REGIONS - 1 || same_out) && ln_flush(s) != KZV_OK) return nullptr;
        partial += (1 + g_ln_pending.size()) * LN_REGION_FLOATS;
        g_ln_pending.push_back(LnFold{partial, dgamma, dbeta, H});
    }
    return partial;
}
int kzv_ln_partial_fold(float* partial, float* dgamma, float* dbeta, int H, hipStream_t s) {
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * H + 255) / 256), dim3(256), 0, s, partial, dgamma, dbeta, H);
    return kzv_check_launch("layernorm_bwd_fold");
}

KzvLnDeferScope::KzvLnDeferScope(hipStream_t stream) : s(stream) { ++g_ln_defer; }
KzvLnDeferScope::~KzvLnDeferScope() { if (--g_ln_defer == 0) (void)ln_flush(s); }

extern "C" int kzv_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                 float* stats, int rows, int H, float eps, void* stream) {
    return kzv_ln_fwd_ex(x, gamma, beta, y_bf16, y_f32, stats, rows, H, eps, 1, 0, 0.f, 0, (hipStream_t)stream);
}

extern "C" int kzv_layernorm_fwd_fp8(const float* x, const float* gamma, const float* beta, void* y_bf16, void* y_fp8, float* y_scale,
                                     float* stats, int rows, int H, float eps, void* stream) {
    if (!y_fp8 || !y_scale) return kzv_fail(KZV_E_ARG, "layernorm_fwd_fp8: y_fp8 and y_scale are required");
    return kzv_ln_fwd_ex(x, gamma, beta, y_bf16, nullptr, stats, rows, H, eps, 1, 0, 0.f, 0, (hipStream_t)stream, y_fp8, y_scale);
}

extern "C" int kzv_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, const float* stats, const float* gamma,
                                 float* dx, int accumulate_dx, float* dgamma, float* dbeta, int rows, int H, void* stream) {
    return kzv_ln_bwd_ex(dy, dy_is_f32, x, stats, gamma, dx, accumulate_dx, dgamma, dbeta, rows, H, 1, 0, 0.f, 0, (hipStream_t)stream);
}
