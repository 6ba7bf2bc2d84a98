// Kernels of the ResNet / BiLSTM / CTC model of ocr_lightning/model.py: everything
// around the MFMA GEMMs of gemm*.hip, which carry the arithmetic (a convolution is im2col + kzv_gemm_nt, its weight gradient
// kzv_gemm_tn on the same column matrix, its input gradient kzv_gemm_nt against the transposed weight + a gather).
//
// Layout: activations are NHWC, rows = pixels (n, h, w), columns = channels -- the row-major [M, C] matrices the GEMMs take.
// Every kernel here is a single streaming pass (HBM-bound, 8- or 16-byte accesses where the channel count allows); none of them
// is on the benchmark path.  Replaces, with the reference line each entry point stands for in include/kzv.h:
//   nn.Conv2d / nn.BatchNorm2d / nn.ReLU / nn.MaxPool2d / BasicBlock residuals of torchvision's resnet34 (model.py:31-32),
//   nn.AdaptiveAvgPool2d((1, 1)) (:34), nn.LSTM over a length-1 sequence (:40-47, :73-75), F.log_softmax + nn.CTCLoss
//   (:51-55, :124-176), nn.SmoothL1Loss over the first min(count, max_boxes) boxes of each sample (:50, :100-122), optim.Adam (:197).
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include <type_traits>

namespace {

inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

// Activation / GEMM-operand element type: bf16 (the engine's MFMA path; default) or fp32 (kzv_ocr_set_precision(1): the reference
// trains this model in fp32, and kzv.OCRModel(precision="fp32") then takes the fp32-operand GEMMs of gemm_f32.hip).  The "bf16"
// buffers of the entry points below are float buffers in that mode; every kernel is instantiated for both.
int g_ocr_f32 = 0;
template <bool F32> __device__ __forceinline__ float ld1(const void* p, int64_t i) {
    if constexpr (F32) return ((const float*)p)[i]; else return bf2f(((const bf16_t*)p)[i]);
}
template <bool F32> __device__ __forceinline__ void st1(void* p, int64_t i, float v) {
    if constexpr (F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f2bf(v);
}
template <bool F32> __device__ __forceinline__ f32x4 ld4(const void* p, int64_t i) {      // i % 4 == 0
    if constexpr (F32) return *(const f32x4*)((const float*)p + i);
    else {
        const uint2 r = *(const uint2*)((const bf16_t*)p + i);
        return (f32x4){bf2f((bf16_t)(r.x & 0xffffu)), bf2f((bf16_t)(r.x >> 16)), bf2f((bf16_t)(r.y & 0xffffu)), bf2f((bf16_t)(r.y >> 16))};
    }
}
template <bool F32> __device__ __forceinline__ void st4(void* p, int64_t i, const f32x4& v) {
    if constexpr (F32) *(f32x4*)((float*)p + i) = v;
    else *(uint2*)((bf16_t*)p + i) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}
#define KZV_OCR_LAUNCH(kernel, grid, block, stream, ...)                                                   \
    do {                                                                                                   \
        if (g_ocr_f32) hipLaunchKernelGGL((kernel<true>), grid, block, 0, stream, __VA_ARGS__);            \
        else hipLaunchKernelGGL((kernel<false>), grid, block, 0, stream, __VA_ARGS__);                     \
    } while (0)

// ---------------------------------------------------------------------------------------------- layout / im2col
// images fp32 [N, C, H, W] (what ocr_collate_fn stacks) -> NHWC bf16
template <bool F32>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, void* __restrict__ out, int N, int C, int H, int W) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * H * W) return;
    const int64_t n = t / ((int64_t)H * W), hw = t - n * H * W;
    for (int c = 0; c < C; ++c) st1<F32>(out, t * C + c, x[(n * C + c) * (int64_t)H * W + hw]);
}

// cols[m][(kh * KW + kw) * C + c] = x[n, ho * s - p + kh, wo * s - p + kw, c] (0 outside), m = (n * Ho + ho) * Wo + wo;
// columns >= KH * KW * C (padding of the GEMM's K to a multiple of 64) are 0.  One thread per 8 columns.
template <bool VEC, bool F32>
__global__ void im2col_kernel(const void* __restrict__ xv, void* __restrict__ colsv, int N, int H, int W, int C, int KH, int KW,
                              int stride, int pad, int Ho, int Wo, int Kp) {
    using E = std::conditional_t<F32, float, bf16_t>;
    const E* x = (const E*)xv; E* cols = (E*)colsv;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kc = Kp >> 3;
    if (t >= (int64_t)N * Ho * Wo * kc) return;
    const int64_t m = t / kc;
    const int k0 = (int)(t - m * kc) * 8;
    const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
    const int64_t n = m / ((int64_t)Wo * Ho);
    const int K = KH * KW * C;
    E v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (E)0;
    if (VEC) {                                   // C % 8 == 0: the 8 columns share one tap
        if (k0 < K) {
            const int tap = k0 / C, c = k0 - tap * C, kh = tap / KW, kw = tap - kh * KW;
            const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
            if (h >= 0 && h < H && w >= 0 && w < W) {
                const E* src = x + ((n * H + h) * W + w) * C + c;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[j];                  // 16 / 32 contiguous bytes: vectorised by the compiler
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            if (k < K) {
                const int tap = k / C, c = k - tap * C, kh = tap / KW, kw = tap - kh * KW;
                const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
                if (h >= 0 && h < H && w >= 0 && w < W) v[j] = x[((n * H + h) * W + w) * C + c];
            }
        }
    }
    E* dst = cols + m * Kp + k0;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = v[j];
}

// input gradient of a convolution from the gradient of its column matrix (gather form, no atomics):
// dx[n, h, w, c] (+)= sum over taps (kh, kw) with (h + p - kh) % s == 0 of dcols[(n, (h + p - kh) / s, (w + p - kw) / s)][(kh, kw, c)]
__global__ void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int N, int H, int W, int C, int KH, int KW,
                              int stride, int pad, int Ho, int Wo, int Kp, int accumulate) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = C >> 2;
    if (t >= (int64_t)N * H * W * c4) return;
    const int64_t px = t / c4;
    const int c = (int)(t - px * c4) * 4;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const int64_t n = px / ((int64_t)W * H);
    f32x4 acc = accumulate ? *(const f32x4*)(dx + px * C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kh = 0; kh < KH; ++kh) {
        const int hh = h + pad - kh;
        if (hh < 0 || hh % stride) continue;
        const int ho = hh / stride;
        if (ho >= Ho) continue;
        for (int kw = 0; kw < KW; ++kw) {
            const int ww = w + pad - kw;
            if (ww < 0 || ww % stride) continue;
            const int wo = ww / stride;
            if (wo >= Wo) continue;
            acc += *(const f32x4*)(dcols + ((n * Ho + ho) * Wo + wo) * Kp + (kh * KW + kw) * C + c);
        }
    }
    *(f32x4*)(dx + px * C + c) = acc;
}

// conv weight fp32 [Cout, Cin, KH, KW] (torch) -> bf16 [Cout, Kp] in (kh, kw, cin) column order, and its transpose [Kp, Cout]
template <bool F32>
__global__ void conv_weight_kernel(const float* __restrict__ w, void* __restrict__ wp, void* __restrict__ wpT, int Cout, int Cin,
                                   int KH, int KW, int Kp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Cout * Kp) return;
    const int o = (int)(t / Kp), k = (int)(t - (int64_t)o * Kp);
    float v = 0.f;
    if (k < KH * KW * Cin) {
        const int tap = k / Cin, c = k - tap * Cin, kh = tap / KW, kw = tap - kh * KW;
        v = w[(((int64_t)o * Cin + c) * KH + kh) * KW + kw];
    }
    st1<F32>(wp, t, v);
    if (wpT) st1<F32>(wpT, (int64_t)k * Cout + o, v);
}
// gradient of the packed weight [Cout, Kp] -> torch layout [Cout, Cin, KH, KW] (accumulated)
__global__ void conv_wgrad_unpack_kernel(const float* __restrict__ gp, float* __restrict__ g, int Cout, int Cin, int KH, int KW, int Kp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Cout * Cin * KH * KW) return;
    const int kw = (int)(t % KW), kh = (int)((t / KW) % KH), c = (int)((t / ((int64_t)KW * KH)) % Cin), o = (int)(t / ((int64_t)KW * KH * Cin));
    g[t] += gp[(int64_t)o * Kp + (kh * KW + kw) * Cin + c];
}

// every convolution of the model in ONE launch each way (a ResNet34 step repacks 36 weights after the optimizer step and unpacks 36
// weight gradients at the end of the backward: 72 launches of ~6 us otherwise): descriptors in the kernel arguments, a block finds its
// convolution by its first element index
struct ConvDesc { const float* w; void* wp; void* wpT; const float* gp; float* g; int Cout, Cin, KH, KW, Kp; unsigned b0; };
constexpr int CONV_MAX = 40, CONV_ROW_MAX = 4672;      // floats of one packed row held in LDS (ResNet34: 512 x 3 x 3 = 4608)
struct ConvTable { ConvDesc d[CONV_MAX]; int n; };
__device__ __forceinline__ int conv_lookup(const ConvTable& t, unsigned blk) {        // block-uniform: scalar loads from the argument block
    int k = 0;
    while (k + 1 < t.n && blk >= t.d[k + 1].b0) ++k;
    return __builtin_amdgcn_readfirstlane(k);
}
// one block per (convolution, output channel): the channel's Cin * KH * KW weights are read as they lie (contiguous), permuted through
// LDS into the (kh, kw, cin) column order of im2col, and written as one contiguous packed row (zero beyond KH * KW * Cin)
template <bool F32>
__global__ __launch_bounds__(256) void conv_weight_multi_kernel(const ConvTable t) {
    __shared__ float row[CONV_ROW_MAX];
    const ConvDesc d = t.d[conv_lookup(t, blockIdx.x)];
    const int o = blockIdx.x - d.b0, KK = d.Cin * d.KH * d.KW;
    for (int i = threadIdx.x; i < KK; i += 256) row[i] = d.w[(int64_t)o * KK + i];
    __syncthreads();
    for (int kk = threadIdx.x; kk < d.Kp; kk += 256) {
        float v = 0.f;
        if (kk < KK) { const int tap = kk / d.Cin, c = kk - tap * d.Cin; v = row[c * d.KH * d.KW + tap]; }       // tap = kh * KW + kw
        st1<F32>(d.wp, (int64_t)o * d.Kp + kk, v);
    }
}
// the transposed packed copies [Kp, Cout] from the packed ones [Cout, Kp]: 64 x 64 tiles through LDS, both sides contiguous;
// blockIdx.x -> (convolution, tile) through the same table (b0 counts tiles here)
template <bool F32>
__global__ __launch_bounds__(256) void conv_weight_transpose_multi_kernel(const ConvTable t) {
    __shared__ float tile[64][65];
    const ConvDesc d = t.d[conv_lookup(t, blockIdx.x)];
    const int tl = blockIdx.x - d.b0, tk = (d.Kp + 63) / 64, to = tl / tk, tkk = tl - to * tk;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = to * 64 + ty + 4 * i, kk = tkk * 64 + tx;
        tile[ty + 4 * i][tx] = (o < d.Cout && kk < d.Kp) ? ld1<F32>(d.wp, (int64_t)o * d.Kp + kk) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int kk = tkk * 64 + ty + 4 * i, o = to * 64 + tx;
        if (kk < d.Kp && o < d.Cout) st1<F32>(d.wpT, (int64_t)kk * d.Cout + o, tile[tx][ty + 4 * i]);
    }
}
// the reverse for the weight gradients: one packed row in, the channel's torch-layout slice accumulated contiguously
__global__ __launch_bounds__(256) void conv_wgrad_unpack_multi_kernel(const ConvTable t) {
    __shared__ float row[CONV_ROW_MAX];
    const ConvDesc d = t.d[conv_lookup(t, blockIdx.x)];
    const int o = blockIdx.x - d.b0, KK = d.Cin * d.KH * d.KW, taps = d.KH * d.KW;
    for (int i = threadIdx.x; i < KK; i += 256) row[i] = d.gp[(int64_t)o * d.Kp + i];
    __syncthreads();
    for (int i = threadIdx.x; i < KK; i += 256) { const int c = i / taps, tap = i - c * taps; d.g[(int64_t)o * KK + i] += row[tap * d.Cin + c]; }
}

// ---------------------------------------------------------------------------------------------- BatchNorm2d
// per-channel sums over the rows of y fp32 [M, C]: pass 0 -> sum(y), pass 1 -> sum((y - mean)^2) with mean = sum0 / M.
// One thread per (row slice, 4 channels); a block's partials meet through LDS and go to row blockIdx.x of `out` [blocks, C];
// bn_rows_reduce_kernel adds the rows in a fixed order (bit-reproducible statistics).
__global__ __launch_bounds__(256) void bn_colsum_kernel(const float* __restrict__ y, int64_t M, int C, const float* __restrict__ sum0,
                                                        float* __restrict__ out, int centered, int rows_per_block) {
    __shared__ float red[256 * 4];
    const int c4n = C >> 2, tpc = 256 / min(c4n, 256);          // threads per channel quad (C >= 4; C = 64..512 -> 16..2)
    const int cq = threadIdx.x % min(c4n, 256), slice = threadIdx.x / min(c4n, 256);
    for (int cb = 0; cb < c4n; cb += 256) {                     // C > 1024 never happens here; loop kept for generality
        const int c = (cb + cq) * 4;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (cb + cq < c4n && slice < tpc) {
            f32x4 mu = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (centered) { mu = *(const f32x4*)(sum0 + c); mu *= 1.f / (float)M; }
            const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
            for (int64_t r = r0 + slice; r < r1; r += tpc) {
                f32x4 v = *(const f32x4*)(y + r * C + c);
                if (centered) { v -= mu; v *= v; }
                acc += v;
            }
        }
        *(f32x4*)(red + threadIdx.x * 4) = acc;
        __syncthreads();
        if (slice == 0 && cb + cq < c4n) {      // this block's partial row: summed in a FIXED order by the finalize kernel (no atomics:
            for (int s2 = 1; s2 < tpc; ++s2) acc += *(const f32x4*)(red + (s2 * min(c4n, 256) + cq) * 4);     // a one-ulp difference of a
            *(f32x4*)(out + (int64_t)blockIdx.x * C + c) = acc;                    // mean flips bf16 roundings of the activations downstream)
        }
        __syncthreads();
    }
}
// tot[c] = sum over the `rows` partial rows of part [rows, C] in a FIXED order: 16 channels per workgroup, 16 row slices per channel
// (slice k adds rows k, k + 16, ... in order; thread 0 of the channel adds the 16 slice sums in order).  FINAL: `part` holds the
// centred squares, `sum0` the plain sums: mean / rstd and the running statistics come out of the same launch (nn.BatchNorm2d:
// momentum 0.1, UNBIASED variance into running_var).
template <bool FINAL>
__global__ __launch_bounds__(256) void bn_rows_reduce_kernel(const float* __restrict__ part, float* __restrict__ tot, int rows, int C,
                                                             const float* __restrict__ sum0, float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var, int64_t M, float eps, float momentum) {
    __shared__ float red[16][17];
    const int ch = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + ch;
    float sacc = 0.f;
    if (c < C) for (int r = slice; r < rows; r += 16) sacc += part[(int64_t)r * C + c];
    red[slice][ch] = sacc;
    __syncthreads();
    if (slice != 0 || c >= C) return;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][ch];
    if constexpr (!FINAL) tot[c] = t;
    else {
        const float mu = sum0[c] / (float)M, var = t / (float)M;
        mean[c] = mu; rstd[c] = rsqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * ((float)M / (float)max((int64_t)1, M - 1));
        }
    }
}
// mean / rstd from the two sums (train) or from the running statistics (eval); train also updates the running statistics the
// way nn.BatchNorm2d does (momentum 0.1, UNBIASED variance into running_var)
__global__ void bn_finalize_kernel(const float* sum0, const float* sum1, float* mean, float* rstd, float* run_mean, float* run_var,
                                   int C, int64_t M, float eps, float momentum, int train) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (train) {
        const float mu = sum0[c] / (float)M, var = sum1[c] / (float)M;
        mean[c] = mu; rstd[c] = rsqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * ((float)M / (float)max((int64_t)1, M - 1));
        }
    } else {
        mean[c] = run_mean[c]; rstd[c] = rsqrtf(run_var[c] + eps);
    }
}
// a = [relu](gamma * (y - mean) * rstd + beta [+ resid]) -> bf16 (the next GEMM operand / pooling input)
template <bool F32>
__global__ void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta, const void* __restrict__ resid,
                                void* __restrict__ out, int64_t M, int C, int relu) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4n = C >> 2;
    if (t >= M * c4n) return;
    const int c = (int)(t % c4n) * 4;
    const f32x4 v = *(const f32x4*)(y + t * 4), mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c);
    const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
    f32x4 z = (v - mu) * rs * ga + be;
    if (resid) z += ld4<F32>(resid, t * 4);
    if (relu) { z[0] = fmaxf(z[0], 0.f); z[1] = fmaxf(z[1], 0.f); z[2] = fmaxf(z[2], 0.f); z[3] = fmaxf(z[3], 0.f); }
    st4<F32>(out, t * 4, z);
}
// backward, pass 1: dz = da (* (a > 0) if relu); dbeta += sum dz, dgamma += sum dz * xhat; dz written (fp32) for pass 2 and for
// the residual branch.  Same block decomposition as bn_colsum_kernel.
template <bool F32>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ da, const void* __restrict__ a, const float* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dz,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t M, int C, int relu,
                                                            int rows_per_block) {
    __shared__ float red[256 * 8];
    const int c4n = C >> 2, nq = min(c4n, 256), tpc = 256 / nq;
    const int cq = threadIdx.x % nq, slice = threadIdx.x / nq;
    const int c = cq * 4;
    f32x4 sg = (f32x4){0.f, 0.f, 0.f, 0.f}, sb = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (cq < c4n && slice < tpc) {
        const f32x4 mu = *(const f32x4*)(mean + c), rs = *(const f32x4*)(rstd + c);
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
        for (int64_t r = r0 + slice; r < r1; r += tpc) {
            f32x4 g = *(const f32x4*)(da + r * C + c);
            if (relu) {
                const f32x4 av = ld4<F32>(a, r * C + c);                 // the ReLU output: gradient flows where it is non-zero
#pragma unroll
                for (int q = 0; q < 4; ++q) if (av[q] == 0.f) g[q] = 0.f;
            }
            *(f32x4*)(dz + r * C + c) = g;
            const f32x4 xh = (*(const f32x4*)(y + r * C + c) - mu) * rs;
            sb += g; sg += g * xh;
        }
    }
    *(f32x4*)(red + threadIdx.x * 8) = sg; *(f32x4*)(red + threadIdx.x * 8 + 4) = sb;
    __syncthreads();
    if (slice == 0 && cq < c4n) {
        for (int s2 = 1; s2 < tpc; ++s2) { sg += *(const f32x4*)(red + (s2 * nq + cq) * 8); sb += *(const f32x4*)(red + (s2 * nq + cq) * 8 + 4); }
        *(f32x4*)(dgamma + (int64_t)blockIdx.x * C + c) = sg;       // partial rows [blocks, C] (dgamma / dbeta here = scratch)
        *(f32x4*)(dbeta + (int64_t)blockIdx.x * C + c) = sb;
    }
}
// totals of this launch (read by the second pass) and their accumulation into the gradient buffers; fixed summation order
// (bn_rows_reduce_kernel's decomposition)
__global__ __launch_bounds__(256) void bn_bwd_totals_kernel(const float* __restrict__ pg, const float* __restrict__ pb, float* __restrict__ tot,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C) {
    __shared__ float red[2][16][17];
    const int ch = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + ch;
    float g = 0.f, b = 0.f;
    if (c < C) for (int r = slice; r < rows; r += 16) { g += pg[(int64_t)r * C + c]; b += pb[(int64_t)r * C + c]; }
    red[0][slice][ch] = g; red[1][slice][ch] = b;
    __syncthreads();
    if (slice != 0 || c >= C) return;
    g = 0.f; b = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { g += red[0][k][ch]; b += red[1][k][ch]; }
    tot[c] = g; tot[C + c] = b;
    dgamma[c] += g; dbeta[c] += b;
}
// pass 2: dy = gamma * rstd * (dz - dbeta / M - xhat * dgamma / M) (train) or gamma * rstd * dz (eval) -> bf16 (GEMM operand)
template <bool F32>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, void* __restrict__ dy, int64_t M, int C, int train) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4n = C >> 2;
    if (t >= M * c4n) return;
    const int c = (int)(t % c4n) * 4;
    const f32x4 g = *(const f32x4*)(dz + t * 4), rs = *(const f32x4*)(rstd + c), ga = *(const f32x4*)(gamma + c);
    f32x4 r = g;
    if (train) {
        const f32x4 xh = (*(const f32x4*)(y + t * 4) - *(const f32x4*)(mean + c)) * rs;
        const float im = 1.f / (float)M;
        r = g - *(const f32x4*)(dbeta + c) * im - xh * *(const f32x4*)(dgamma + c) * im;
    }
    r = r * ga * rs;
    st4<F32>(dy, t * 4, r);
}

// ---------------------------------------------------------------------------------------------- pooling
// MaxPool2d(3, stride 2, padding 1) on NHWC bf16; idx = winning tap (kh * 3 + kw), first maximum like torch
template <bool F32>
__global__ void maxpool_fwd_kernel(const void* __restrict__ x, void* __restrict__ out, unsigned char* __restrict__ idx, int N, int H, int W,
                                   int C, int Ho, int Wo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * Ho * Wo * C) return;
    const int c = (int)(t % C);
    const int64_t m = t / C;
    const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
    const int64_t n = m / ((int64_t)Wo * Ho);
    float best = -INFINITY; int bi = 0;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int h = ho * 2 - 1 + kh, w = wo * 2 - 1 + kw;
            if (h < 0 || h >= H || w < 0 || w >= W) continue;
            const float v = ld1<F32>(x, ((n * H + h) * W + w) * C + c);
            if (v > best) { best = v; bi = kh * 3 + kw; }
        }
    st1<F32>(out, t, best); idx[t] = (unsigned char)bi;
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ idx, float* __restrict__ dx, int N, int H, int W,
                                   int C, int Ho, int Wo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * H * W * C) return;
    const int c = (int)(t % C);
    const int64_t px = t / C;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const int64_t n = px / ((int64_t)W * H);
    float acc = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + 1 - kh;
        if (hh < 0 || (hh & 1) || (hh >> 1) >= Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int ww = w + 1 - kw;
            if (ww < 0 || (ww & 1) || (ww >> 1) >= Wo) continue;
            const int64_t o = (((n * Ho + (hh >> 1)) * Wo + (ww >> 1)) * C) + c;
            if (idx[o] == kh * 3 + kw) acc += dout[o];
        }
    }
    dx[t] = acc;
}
// AdaptiveAvgPool2d((1, 1)) + flatten: feat[n][c] = mean over the HW pixels; fp32 and bf16 copies
template <bool F32>
__global__ void avgpool_fwd_kernel(const void* __restrict__ x, float* __restrict__ f32, void* __restrict__ f16, int N, int HW, int C) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * C) return;
    const int n = t / C, c = t - n * C;
    float s = 0.f;
    for (int i = 0; i < HW; ++i) s += ld1<F32>(x, ((int64_t)n * HW + i) * C + c);
    s /= (float)HW;
    f32[t] = s; st1<F32>(f16, t, s);
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dfeat, float* __restrict__ dx, int N, int HW, int C) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * HW * C) return;
    const int c = (int)(t % C);
    const int64_t n = t / ((int64_t)HW * C);
    dx[t] = dfeat[n * C + c] / (float)HW;
}

// ---------------------------------------------------------------------------------------------- LSTM cell, one time step, zero state
// gates fp32 [B, 4H] = x W_ih^T + b_ih (b_hh [4H] is added here) in torch's order (i, f, g, o); h0 = c0 = 0 (the reference feeds a length-1 sequence,
// model.py:73-75), so c = sigmoid(i) * tanh(g), h = sigmoid(o) * tanh(c); W_hh never sees a non-zero operand.
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
template <bool F32>
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ bhh, float* __restrict__ h32, void* __restrict__ h16,
                                     int64_t ldh, int B, int Hh) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * Hh) return;
    const int b = t / Hh, j = t - b * Hh;
    const float* g = gates + (int64_t)b * 4 * Hh;
    const float c = sigm(g[j] + bhh[j]) * tanhf(g[2 * Hh + j] + bhh[2 * Hh + j]);
    const float h = sigm(g[3 * Hh + j] + bhh[3 * Hh + j]) * tanhf(c);
    h32[(int64_t)b * ldh + j] = h; st1<F32>(h16, (int64_t)b * ldh + j, h);
}
// dgates (bf16, GEMM operand) from dh: the forget gate receives no gradient (c0 = 0)
template <bool F32>
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ bhh, const float* __restrict__ dh, int64_t lddh,
                                     void* __restrict__ dg, int B, int Hh) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * Hh) return;
    const int b = t / Hh, j = t - b * Hh;
    const float* g = gates + (int64_t)b * 4 * Hh;
    const float i = sigm(g[j] + bhh[j]), gg = tanhf(g[2 * Hh + j] + bhh[2 * Hh + j]), o = sigm(g[3 * Hh + j] + bhh[3 * Hh + j]);
    const float c = i * gg, tc = tanhf(c), d = dh[(int64_t)b * lddh + j];
    const float dc = d * o * (1.f - tc * tc);
    const int64_t q = (int64_t)b * 4 * Hh;
    st1<F32>(dg, q + j, dc * gg * i * (1.f - i));
    st1<F32>(dg, q + Hh + j, 0.f);
    st1<F32>(dg, q + 2 * Hh + j, dc * i * (1.f - gg * gg));
    st1<F32>(dg, q + 3 * Hh + j, d * tc * o * (1.f - o));
}

// ---------------------------------------------------------------------------------------------- log_softmax + CTC
// one wave per row: lp = x - logsumexp(x)
__global__ __launch_bounds__(64) void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ lp, int rows, int C) {
    const int r = blockIdx.x;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 64) mx = fmaxf(mx, x[(int64_t)r * C + c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) s += __expf(x[(int64_t)r * C + c] - mx);
    s = wave_sum(s);
    const float l = mx + __logf(s);
    for (int c = threadIdx.x; c < C; c += 64) lp[(int64_t)r * C + c] = x[(int64_t)r * C + c] - l;
}
__device__ __forceinline__ float lse2(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = fmaxf(a, b);
    return m + __logf(__expf(a - m) + __expf(b - m));
}
// nn.CTCLoss on log-probabilities lp [T, B, C]: one workgroup per sample, one thread per extended-label state s (S = 2 L + 1);
// alpha and beta tables [T][S] in the caller's scratch.  Writes nll[b] (0 where infinite and zero_infinity) and the gradient with
// respect to the LOGITS the log-probabilities came from (ATen's ctc_loss backward returns exp(lp) - exp(log sum alpha beta + nll - lp),
// which log_softmax's backward leaves unchanged because it sums to 0 over the classes), scaled by gscale[b].
__global__ __launch_bounds__(1024) void ctc_kernel(const float* __restrict__ lp, const int64_t* __restrict__ targets, int64_t ldt,
                                                   const int64_t* __restrict__ in_len, const int64_t* __restrict__ tg_len, int T, int B, int C,
                                                   int blank, int zero_inf, float* __restrict__ alpha, float* __restrict__ beta,
                                                   float* __restrict__ nll, const float* __restrict__ gscale, float* __restrict__ dlogits, int Smax) {
    const int b = blockIdx.x, s = threadIdx.x;
    const int Ti = (int)in_len[b], L = (int)tg_len[b], S = 2 * L + 1;
    float* al = alpha + (int64_t)b * T * Smax;
    float* be = beta + (int64_t)b * T * Smax;
    // labels are clamped into the class range: a bad target must not become an out-of-bounds read (ATen asserts instead)
    const int lab = (s < S) ? ((s & 1) ? min(max((int)targets[(int64_t)b * ldt + (s >> 1)], 0), C - 1) : blank) : blank;
    const bool skip_ok = s < S && (s & 1) && s >= 3 && lab != (int)targets[(int64_t)b * ldt + (s >> 1) - 1];       // may come from s - 2
    __shared__ float res;
    if (Ti <= 0 || L > Ti) {                 // no valid path at all (also covers the empty input)
        if (s == 0) res = INFINITY;
    } else {
        for (int t = 0; t < Ti; ++t) {
            float v = -INFINITY;
            if (s < S) {
                const float e = lp[((int64_t)t * B + b) * C + lab];
                if (t == 0) v = (s <= 1) ? e : -INFINITY;
                else {
                    const float* pa = al + (int64_t)(t - 1) * Smax;
                    float a = pa[s];
                    if (s >= 1) a = lse2(a, pa[s - 1]);
                    if (skip_ok) a = lse2(a, pa[s - 2]);
                    v = a + e;
                }
            }
            __syncthreads();
            if (s < S) al[(int64_t)t * Smax + s] = v;
            __syncthreads();
        }
        for (int t = Ti - 1; t >= 0; --t) {
            float v = -INFINITY;
            if (s < S) {
                const float e = lp[((int64_t)t * B + b) * C + lab];
                if (t == Ti - 1) v = (s >= S - 2) ? e : -INFINITY;
                else {
                    const float* pb = be + (int64_t)(t + 1) * Smax;
                    float a = pb[s];
                    if (s + 1 < S) a = lse2(a, pb[s + 1]);
                    // s -> s + 2 when s is a label whose successor label differs
                    if ((s & 1) && s + 2 < S && lab != (int)targets[(int64_t)b * ldt + (s >> 1) + 1]) a = lse2(a, pb[s + 2]);
                    v = a + e;
                }
            }
            __syncthreads();
            if (s < S) be[(int64_t)t * Smax + s] = v;
            __syncthreads();
        }
        if (s == 0) {
            const float* la = al + (int64_t)(Ti - 1) * Smax;
            res = -lse2(la[S - 1], S >= 2 ? la[S - 2] : -INFINITY);
        }
    }
    __syncthreads();
    const float n = res;
    const bool bad = !(n < INFINITY);
    if (s == 0) nll[b] = (bad && zero_inf) ? 0.f : n;
    if (!dlogits) return;
    const float gs = gscale[b];
    for (int t = 0; t < T; ++t) {
        for (int c = s; c < C; c += blockDim.x) {
            float g = 0.f;
            if (!bad && t < Ti) {
                const float l = lp[((int64_t)t * B + b) * C + c];
                float acc = -INFINITY;                             // log sum over the states carrying class c of alpha * beta
                for (int s2 = 0; s2 < S; ++s2) {
                    const int lb = (s2 & 1) ? min(max((int)targets[(int64_t)b * ldt + (s2 >> 1)], 0), C - 1) : blank;
                    if (lb == c) acc = lse2(acc, al[(int64_t)t * Smax + s2] + be[(int64_t)t * Smax + s2]);
                }
                g = (__expf(l) - __expf(acc + n - l)) * gs;
            }
            dlogits[((int64_t)t * B + b) * C + c] = g;
        }
    }
}

// ---------------------------------------------------------------------------------------------- SmoothL1 box loss
// model.py:100-122: per sample i with n_i = min(count_i, max_boxes) > 0 the MEAN SmoothL1 (beta = 1) over its first n_i boxes,
// then the mean over those samples.  out[0] += loss; dpred receives the gradient (zeros elsewhere).  One workgroup per sample.
__global__ __launch_bounds__(64) void smooth_l1_kernel(const float* __restrict__ pred, int max_boxes, const float* __restrict__ gt, int gt_boxes,
                                                       const int* __restrict__ counts, int B, float* __restrict__ out, float* __restrict__ dpred) {
    const int i = blockIdx.x;
    __shared__ int nvalid_s;
    if (threadIdx.x == 0) {
        int nv = 0;
        for (int j = 0; j < B; ++j) nv += min(counts[j], max_boxes) > 0;
        nvalid_s = nv;
    }
    __syncthreads();
    const int n = min(counts[i], max_boxes), nv = nvalid_s;
    for (int e = threadIdx.x; e < max_boxes * 4; e += 64) dpred[(int64_t)i * max_boxes * 4 + e] = 0.f;
    if (n <= 0 || nv <= 0) return;
    __syncthreads();
    float acc = 0.f;
    const float w = 1.f / ((float)(n * 4) * (float)nv);
    for (int e = threadIdx.x; e < n * 4; e += 64) {
        const float d = pred[(int64_t)i * max_boxes * 4 + e] - gt[(int64_t)i * gt_boxes * 4 + e];
        const float ad = fabsf(d);
        acc += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
        dpred[(int64_t)i * max_boxes * 4 + e] = (ad < 1.f ? d : copysignf(1.f, d)) * w;
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) atomicAdd(out, acc * w);
}

// ---------------------------------------------------------------------------------------------- optim.Adam
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                            float b1, float b2, float eps, float bc1, float sqrt_bc2) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float gr = g[t];
    const float mm = b1 * m[t] + (1.f - b1) * gr;
    const float vv = b2 * v[t] + (1.f - b2) * gr * gr;
    m[t] = mm; v[t] = vv;
    p[t] -= (lr / bc1) * mm / (sqrtf(vv) / sqrt_bc2 + eps);
}
// the same update with the step's bias corrections read from device memory (bc[0] = 1 - beta1^t, bc[1] = sqrt(1 - beta2^t)): a step
// replayed from a captured graph has no step-dependent kernel argument
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                float b1, float b2, float eps, const float* __restrict__ bc) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float bc1 = bc[0], sqrt_bc2 = bc[1];
    const float gr = g[t];
    const float mm = b1 * m[t] + (1.f - b1) * gr;
    const float vv = b2 * v[t] + (1.f - b2) * gr * gr;
    m[t] = mm; v[t] = vv;
    p[t] -= (lr / bc1) * mm / (sqrtf(vv) / sqrt_bc2 + eps);
}
template <bool F32>
__global__ void cast_bf16_kernel(const float* __restrict__ x, void* __restrict__ out, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) st1<F32>(out, t, x[t]);
}
// [R, Cc] fp32 -> its bf16 transpose [Cc, R] (input-gradient GEMMs of the small Linear / LSTM weights)
template <bool F32>
__global__ void cast_transpose_kernel(const float* __restrict__ x, void* __restrict__ out, int R, int Cc) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)R * Cc) return;
    const int r = (int)(t / Cc), c = (int)(t - (int64_t)r * Cc);
    st1<F32>(out, (int64_t)c * R + r, x[t]);
}

}  // namespace

#define KZV_OCR_NULL(cond, what) do { if (cond) return kzv_fail(KZV_E_ARG, what); } while (0)

extern "C" int kzv_ocr_set_precision(int fp32) { g_ocr_f32 = fp32 != 0; return KZV_OK; }

extern "C" int kzv_ocr_nchw_to_nhwc(const float* x, void* out_bf16, int N, int C, int H, int W, void* stream) {
    KZV_OCR_NULL(!x || !out_bf16 || N <= 0 || C <= 0 || H <= 0 || W <= 0, "ocr_nchw_to_nhwc: bad argument");
    KZV_OCR_LAUNCH(nchw_to_nhwc_kernel, dim3(nblk((int64_t)N * H * W, 256)), dim3(256), (hipStream_t)stream, x, out_bf16, N, C, H, W);
    return kzv_check_launch("ocr_nchw_to_nhwc");
}
extern "C" int kzv_ocr_im2col(const void* x, void* cols, int N, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp, void* stream) {
    KZV_OCR_NULL(!x || !cols || N <= 0 || Kp % 8 || Kp < KH * KW * C || stride <= 0, "ocr_im2col: bad argument (Kp multiple of 8, >= KH*KW*C)");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    KZV_OCR_NULL(Ho <= 0 || Wo <= 0, "ocr_im2col: empty output");
    const int64_t total = (int64_t)N * Ho * Wo * (Kp / 8);
#define KZV_IM2COL(V, F) hipLaunchKernelGGL((im2col_kernel<V, F>), dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, x, cols, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kp)
    if (C % 8 == 0) { if (g_ocr_f32) KZV_IM2COL(true, true); else KZV_IM2COL(true, false); }
    else { if (g_ocr_f32) KZV_IM2COL(false, true); else KZV_IM2COL(false, false); }
#undef KZV_IM2COL
    return kzv_check_launch("ocr_im2col");
}
extern "C" int kzv_ocr_col2im(const float* dcols, float* dx, int N, int H, int W, int C, int KH, int KW, int stride, int pad, int Kp, int accumulate, void* stream) {
    KZV_OCR_NULL(!dcols || !dx || C % 4 || Kp % 4 || Kp < KH * KW * C, "ocr_col2im: bad argument (C, Kp multiples of 4)");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    hipLaunchKernelGGL(col2im_kernel, dim3(nblk((int64_t)N * H * W * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, dcols, dx, N, H, W, C, KH, KW, stride, pad, Ho, Wo, Kp, accumulate);
    return kzv_check_launch("ocr_col2im");
}
extern "C" int kzv_ocr_conv_weight(const float* w, void* wp, void* wpT, int Cout, int Cin, int KH, int KW, int Kp, void* stream) {
    KZV_OCR_NULL(!w || !wp || Kp < KH * KW * Cin, "ocr_conv_weight: bad argument");
    KZV_OCR_LAUNCH(conv_weight_kernel, dim3(nblk((int64_t)Cout * Kp, 256)), dim3(256), (hipStream_t)stream, w, wp, wpT, Cout, Cin, KH, KW, Kp);
    return kzv_check_launch("ocr_conv_weight");
}
extern "C" int kzv_ocr_conv_wgrad_unpack(const float* gp, float* g, int Cout, int Cin, int KH, int KW, int Kp, void* stream) {
    KZV_OCR_NULL(!gp || !g, "ocr_conv_wgrad_unpack: null");
    hipLaunchKernelGGL(conv_wgrad_unpack_kernel, dim3(nblk((int64_t)Cout * Cin * KH * KW, 256)), dim3(256), 0, (hipStream_t)stream, gp, g, Cout, Cin, KH, KW, Kp);
    return kzv_check_launch("ocr_conv_wgrad_unpack");
}
// the same for n convolutions at once (n <= 40; one launch): arrays of n pointers / geometries on the HOST
extern "C" int kzv_ocr_conv_weight_multi(int n, const float* const* w, void* const* wp, void* const* wpT, const int32_t* geom /* [n][5]: Cout Cin KH KW Kp */, void* stream) {
    KZV_OCR_NULL(n < 1 || n > CONV_MAX || !w || !wp || !wpT || !geom, "ocr_conv_weight_multi: 1..40 convolutions");
    ConvTable t, tt; unsigned b = 0, bt = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t* g = geom + 5 * i;
        KZV_OCR_NULL(!w[i] || !wp[i] || g[4] < g[2] * g[3] * g[1] || g[4] > CONV_ROW_MAX, "ocr_conv_weight_multi: bad entry (packed rows of <= 4672 columns)");
        t.d[i] = ConvDesc{w[i], wp[i], wpT[i], nullptr, nullptr, g[0], g[1], g[2], g[3], g[4], b};
        b += (unsigned)g[0];
        tt.d[i] = t.d[i]; tt.d[i].b0 = bt;
        if (wpT[i]) bt += (unsigned)(((g[0] + 63) / 64) * ((g[4] + 63) / 64));
    }
    t.n = tt.n = n;
    KZV_OCR_LAUNCH(conv_weight_multi_kernel, dim3(b), dim3(256), (hipStream_t)stream, t);
    bool all_t = true;
    for (int i = 0; i < n; ++i) all_t &= wpT[i] != nullptr;
    KZV_OCR_NULL(!all_t && bt, "ocr_conv_weight_multi: transposed copies for all convolutions or for none");
    if (bt) KZV_OCR_LAUNCH(conv_weight_transpose_multi_kernel, dim3(bt), dim3(256), (hipStream_t)stream, tt);
    return kzv_check_launch("ocr_conv_weight_multi");
}
extern "C" int kzv_ocr_conv_wgrad_unpack_multi(int n, const float* const* gp, float* const* g, const int32_t* geom, void* stream) {
    KZV_OCR_NULL(n < 1 || n > CONV_MAX || !gp || !g || !geom, "ocr_conv_wgrad_unpack_multi: 1..40 convolutions");
    ConvTable t; unsigned b = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t* q = geom + 5 * i;
        KZV_OCR_NULL(!gp[i] || !g[i] || q[4] > CONV_ROW_MAX, "ocr_conv_wgrad_unpack_multi: bad entry");
        t.d[i] = ConvDesc{nullptr, nullptr, nullptr, gp[i], g[i], q[0], q[1], q[2], q[3], q[4], b};
        b += (unsigned)q[0];
    }
    t.n = n;
    hipLaunchKernelGGL(conv_wgrad_unpack_multi_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, t);
    return kzv_check_launch("ocr_conv_wgrad_unpack_multi");
}
// d_scratch: kzv_ocr_bn_scratch_floats(M, C) floats
// rows per workgroup of the BatchNorm reductions: ~512 workgroups per launch (two rounds of the chip) between 32 and 256 rows each
static inline int bn_rpb(int64_t M) { const int64_t r = (M + 511) / 512; return r < 32 ? 32 : (r > 256 ? 256 : (int)r); }
extern "C" int64_t kzv_ocr_bn_scratch_floats(int64_t M, int C) { return (2 * (int64_t)nblk(M, bn_rpb(M)) + 2) * C; }
extern "C" int kzv_ocr_bn_fwd(const float* y, int64_t M, int C, const float* gamma, const float* beta, float* run_mean, float* run_var,
                              float* mean, float* rstd, const void* resid_bf16, void* out_bf16, int relu, int train, float eps, float momentum,
                              float* d_scratch, void* stream) {
    KZV_OCR_NULL(!y || !gamma || !beta || !mean || !rstd || !out_bf16 || M <= 0 || C < 4 || C % 4 || C > 1024, "ocr_bn_fwd: bad argument (C multiple of 4, <= 1024)");
    KZV_OCR_NULL(!train && (!run_mean || !run_var), "ocr_bn_fwd: eval mode needs the running statistics");
    hipStream_t s = (hipStream_t)stream;
    if (train) {
        KZV_OCR_NULL(!d_scratch, "ocr_bn_fwd: scratch");
        const int rpb = bn_rpb(M), nb = (int)nblk(M, rpb);
        float* part = d_scratch + 2 * C;                       // [nb, C] partial rows, reused by both passes
        hipLaunchKernelGGL(bn_colsum_kernel, dim3(nb), dim3(256), 0, s, y, M, C, nullptr, part, 0, rpb);
        hipLaunchKernelGGL(bn_rows_reduce_kernel<false>, dim3(nblk(C, 16)), dim3(256), 0, s, part, d_scratch, nb, C, nullptr, nullptr, nullptr, nullptr, nullptr, M, eps, momentum);
        hipLaunchKernelGGL(bn_colsum_kernel, dim3(nb), dim3(256), 0, s, y, M, C, d_scratch, part, 1, rpb);
        hipLaunchKernelGGL(bn_rows_reduce_kernel<true>, dim3(nblk(C, 16)), dim3(256), 0, s, part, d_scratch + C, nb, C, d_scratch, mean, rstd, run_mean, run_var, M, eps, momentum);
    } else
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(nblk(C, 256)), dim3(256), 0, s, nullptr, nullptr, mean, rstd, run_mean, run_var, C, M, eps, momentum, 0);
    KZV_OCR_LAUNCH(bn_apply_kernel, dim3(nblk(M * (C / 4), 256)), dim3(256), s, y, mean, rstd, gamma, beta, resid_bf16, out_bf16, M, C, relu);
    return kzv_check_launch("ocr_bn_fwd");
}
extern "C" int kzv_ocr_bn_bwd(const float* da, const void* a_bf16, const float* y, int64_t M, int C, const float* mean, const float* rstd,
                              const float* gamma, float* dz, float* dgamma, float* dbeta, void* dy_bf16, int relu, int train, float* d_scratch,
                              void* stream) {
    KZV_OCR_NULL(!da || !y || !mean || !rstd || !gamma || !dz || !dgamma || !dbeta || !dy_bf16 || !d_scratch || (relu && !a_bf16) || C % 4 || C > 1024, "ocr_bn_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int rpb = bn_rpb(M), nb = (int)nblk(M, rpb);
    float* tot = d_scratch;                                    // [2, C]: this launch's dgamma | dbeta (the second pass needs them complete)
    float* pg = d_scratch + 2 * C; float* pb = pg + (int64_t)nb * C;
    KZV_OCR_LAUNCH(bn_bwd_reduce_kernel, dim3(nb), dim3(256), s, da, a_bf16, y, mean, rstd, dz, pg, pb, M, C, relu, rpb);
    hipLaunchKernelGGL(bn_bwd_totals_kernel, dim3(nblk(C, 16)), dim3(256), 0, s, pg, pb, tot, dgamma, dbeta, nb, C);
    KZV_OCR_LAUNCH(bn_bwd_apply_kernel, dim3(nblk(M * (C / 4), 256)), dim3(256), s, dz, y, mean, rstd, gamma, tot, tot + C, dy_bf16, M, C, train);
    return kzv_check_launch("ocr_bn_bwd");
}
extern "C" int kzv_ocr_maxpool_fwd(const void* x, void* out, unsigned char* idx, int N, int H, int W, int C, void* stream) {
    KZV_OCR_NULL(!x || !out || !idx, "ocr_maxpool_fwd: null");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    KZV_OCR_LAUNCH(maxpool_fwd_kernel, dim3(nblk((int64_t)N * Ho * Wo * C, 256)), dim3(256), (hipStream_t)stream, x, out, idx, N, H, W, C, Ho, Wo);
    return kzv_check_launch("ocr_maxpool_fwd");
}
extern "C" int kzv_ocr_maxpool_bwd(const float* dout, const unsigned char* idx, float* dx, int N, int H, int W, int C, void* stream) {
    KZV_OCR_NULL(!dout || !idx || !dx, "ocr_maxpool_bwd: null");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblk((int64_t)N * H * W * C, 256)), dim3(256), 0, (hipStream_t)stream, dout, idx, dx, N, H, W, C, Ho, Wo);
    return kzv_check_launch("ocr_maxpool_bwd");
}
extern "C" int kzv_ocr_avgpool_fwd(const void* x, float* f32, void* f16, int N, int HW, int C, void* stream) {
    KZV_OCR_NULL(!x || !f32 || !f16 || HW <= 0, "ocr_avgpool_fwd: bad argument");
    KZV_OCR_LAUNCH(avgpool_fwd_kernel, dim3(nblk((int64_t)N * C, 256)), dim3(256), (hipStream_t)stream, x, f32, f16, N, HW, C);
    return kzv_check_launch("ocr_avgpool_fwd");
}
extern "C" int kzv_ocr_avgpool_bwd(const float* dfeat, float* dx, int N, int HW, int C, void* stream) {
    KZV_OCR_NULL(!dfeat || !dx, "ocr_avgpool_bwd: null");
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(nblk((int64_t)N * HW * C, 256)), dim3(256), 0, (hipStream_t)stream, dfeat, dx, N, HW, C);
    return kzv_check_launch("ocr_avgpool_bwd");
}
extern "C" int kzv_ocr_lstm_cell_fwd(const float* gates, const float* b_hh, float* h32, void* h16, int64_t ldh, int B, int Hh, void* stream) {
    KZV_OCR_NULL(!gates || !b_hh || !h32 || !h16 || B <= 0 || Hh <= 0, "ocr_lstm_cell_fwd: bad argument");
    KZV_OCR_LAUNCH(lstm_cell_fwd_kernel, dim3(nblk((int64_t)B * Hh, 256)), dim3(256), (hipStream_t)stream, gates, b_hh, h32, h16, ldh, B, Hh);
    return kzv_check_launch("ocr_lstm_cell_fwd");
}
extern "C" int kzv_ocr_lstm_cell_bwd(const float* gates, const float* b_hh, const float* dh, int64_t lddh, void* dgates_bf16, int B, int Hh, void* stream) {
    KZV_OCR_NULL(!gates || !b_hh || !dh || !dgates_bf16, "ocr_lstm_cell_bwd: null");
    KZV_OCR_LAUNCH(lstm_cell_bwd_kernel, dim3(nblk((int64_t)B * Hh, 256)), dim3(256), (hipStream_t)stream, gates, b_hh, dh, lddh, dgates_bf16, B, Hh);
    return kzv_check_launch("ocr_lstm_cell_bwd");
}
extern "C" int kzv_ocr_log_softmax(const float* x, float* lp, int rows, int C, void* stream) {
    KZV_OCR_NULL(!x || !lp || rows <= 0 || C <= 0, "ocr_log_softmax: bad argument");
    hipLaunchKernelGGL(log_softmax_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, x, lp, rows, C);
    return kzv_check_launch("ocr_log_softmax");
}
extern "C" int kzv_ocr_ctc(const float* lp, const int64_t* targets, int64_t ld_targets, const int64_t* input_lengths, const int64_t* target_lengths,
                           int T, int B, int C, int blank, int zero_infinity, int max_target_len, float* d_scratch, float* d_nll,
                           const float* d_gscale, float* d_dlogits, void* stream) {
    KZV_OCR_NULL(!lp || !input_lengths || !target_lengths || !d_scratch || !d_nll || T <= 0 || B <= 0 || C <= 0, "ocr_ctc: bad argument");
    KZV_OCR_NULL(max_target_len > 0 && !targets, "ocr_ctc: targets");
    KZV_OCR_NULL(d_dlogits && !d_gscale, "ocr_ctc: the gradient needs per-sample scales");
    // a label longer than its input has no alignment: the kernel gives it loss inf (0 with zero_infinity) and a zero gradient before
    // it looks at a state, so the state tables only ever hold labels of <= T characters, whatever the longest label of the batch is
    // (ocr_lightning's whole-page texts against its length-1 sequence: nn.CTCLoss(zero_infinity=True) returns 0 there too)
    const int Smax = 2 * (max_target_len < T ? max_target_len : T) + 1;
    KZV_OCR_NULL(Smax > 1024, "ocr_ctc: more than 511 time steps are not supported (one thread per extended-label state)");
    int threads = 64;
    while (threads < Smax) threads <<= 1;
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(threads), 0, (hipStream_t)stream, lp, targets, ld_targets, input_lengths, target_lengths, T, B, C, blank,
                       zero_infinity, d_scratch, d_scratch + (int64_t)B * T * Smax, d_nll, d_gscale, d_dlogits, Smax);
    return kzv_check_launch("ocr_ctc");
}
extern "C" int kzv_ocr_smooth_l1_boxes(const float* pred, int max_boxes, const float* gt, int gt_boxes, const int32_t* counts, int B, float* d_loss,
                                       float* d_dpred, void* stream) {
    KZV_OCR_NULL(!pred || !counts || !d_loss || !d_dpred || B <= 0 || max_boxes <= 0 || (gt_boxes > 0 && !gt), "ocr_smooth_l1_boxes: bad argument");
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pred, max_boxes, gt, gt_boxes, counts, B, d_loss, d_dpred);
    return kzv_check_launch("ocr_smooth_l1_boxes");
}
extern "C" int kzv_ocr_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step, void* stream) {
    KZV_OCR_NULL(!p || !g || !m || !v || n <= 0 || step <= 0, "ocr_adam: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
    return kzv_check_launch("ocr_adam");
}
extern "C" int kzv_ocr_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, const float* d_bc, void* stream) {
    KZV_OCR_NULL(!p || !g || !m || !v || n <= 0 || !d_bc, "ocr_adam_dev: bad argument");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, d_bc);
    return kzv_check_launch("ocr_adam_dev");
}
extern "C" int kzv_ocr_cast_bf16(const float* x, void* out, int64_t n, void* stream) {
    KZV_OCR_NULL(!x || !out || n <= 0, "ocr_cast_bf16: bad argument");
    KZV_OCR_LAUNCH(cast_bf16_kernel, dim3(nblk(n, 256)), dim3(256), (hipStream_t)stream, x, out, n);
    return kzv_check_launch("ocr_cast_bf16");
}
extern "C" int kzv_ocr_cast_transpose(const float* x, void* out, int R, int Cc, void* stream) {
    KZV_OCR_NULL(!x || !out || R <= 0 || Cc <= 0, "ocr_cast_transpose: bad argument");
    KZV_OCR_LAUNCH(cast_transpose_kernel, dim3(nblk((int64_t)R * Cc, 256)), dim3(256), (hipStream_t)stream, x, out, R, Cc);
    return kzv_check_launch("ocr_cast_transpose");
}
