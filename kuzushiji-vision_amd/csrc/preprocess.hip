// N2: the reference's image transform on the device (include/kzv.h, "input pipeline on the device").
//
// ResizeWithPadding (src/data/trocr_dataset.py:24-53) = Pillow Image.resize(LANCZOS) + centred paste on white, then
// ToTensor + Normalize(0.5, 0.5) (:97-104).  Pillow's resample (src/libImaging/Resample.c, ImagingResample) is two
// separable 8-bit passes: out = clip8((2^21 + sum_k in[xmin + k] * kk[k]) >> 22) with per-output-pixel windows and 22-bit
// fixed-point Lanczos-3 weights.  The weights are produced on the HOST by kzv_lanczos_coeffs with the same double
// arithmetic and libm sin() as Pillow (a device sin() is not bit-identical), the two passes are integer work on the GPU:
// byte-exact with Pillow by construction, pinned by tests/golden/resize_kat.npz.
//
// HBM-bound byte work: pass 1 reads the crop once (neighbouring outputs share their window in L1/L2) and writes the
// [in_h, new_w] uint8 intermediate; pass 2 reads it once and writes the fp32 CHW crop (12 B per output pixel, the
// dominant term: 64x640 crops -> 491 KB each).
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include <cmath>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// one unaligned 32-bit load fetches the three channel bytes of a pixel (+1 byte of its neighbour): the kernels are bound
// by address processing of byte gathers, not by bytes, so this is 3x fewer memory instructions.  The packed crop buffer
// and the intermediate carry 4 bytes of slack at the end for the last pixel.
typedef unsigned u32_unaligned __attribute__((aligned(1)));
__device__ __forceinline__ unsigned load_rgbx(const uint8_t* px) { return *(const u32_unaligned*)px; }

__device__ __forceinline__ int clip8(int v) { v >>= PRECISION_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// pass 1: tmp[y][xo][c] for y < in_h, xo < new_w (a plain copy when the width is unchanged: Pillow skips the pass)
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ rgb, const kzv_line_desc* __restrict__ desc,
                                                         const int32_t* __restrict__ coef, uint8_t* __restrict__ tmp) {
    const kzv_line_desc d = desc[blockIdx.y];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)d.in_h * d.new_w) return;
    const int y = (int)(idx / d.new_w), xo = (int)(idx - (int64_t)y * d.new_w);
    const uint8_t* row = rgb + d.src_off + (int64_t)y * d.in_w * 3;
    uint8_t* o = tmp + d.tmp_off + ((int64_t)y * d.new_w + xo) * 3;
    if (d.new_w == d.in_w) { o[0] = row[xo * 3]; o[1] = row[xo * 3 + 1]; o[2] = row[xo * 3 + 2]; return; }
    const int xmin = coef[d.hb_off + 2 * xo], cnt = coef[d.hb_off + 2 * xo + 1];
    const int32_t* k = coef + d.hk_off + xo;            // horizontal weights are stored TRANSPOSED [hk_size][new_w]: coalesced over xo
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll 4
    for (int i = 0; i < cnt; ++i) {
        const int w = k[(int64_t)i * d.new_w];
        const unsigned v = load_rgbx(row + (xmin + i) * 3);
        s0 += (int)(v & 0xff) * w; s1 += (int)((v >> 8) & 0xff) * w; s2 += (int)((v >> 16) & 0xff) * w;
    }
    o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

// pass 2 + paste + ToTensor/Normalize: one thread per output pixel (3 channels), white outside the pasted crop
__global__ __launch_bounds__(256) void resample_v_kernel(const kzv_line_desc* __restrict__ desc, const int32_t* __restrict__ coef,
                                                         const uint8_t* __restrict__ tmp, const float* __restrict__ lut,
                                                         float* __restrict__ out, int target_h, int target_w) {
    const kzv_line_desc d = desc[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= target_h * target_w) return;
    const int Y = idx / target_w, X = idx - Y * target_w;
    const int yo = Y - d.paste_y, xo = X - d.paste_x;
    int v0 = 255, v1 = 255, v2 = 255;
    if (yo >= 0 && yo < d.new_h && xo >= 0 && xo < d.new_w) {
        const uint8_t* col = tmp + d.tmp_off + (int64_t)xo * 3;
        const int64_t rs = (int64_t)d.new_w * 3;
        if (d.new_h == d.in_h) {
            const uint8_t* px = col + yo * rs;
            v0 = px[0]; v1 = px[1]; v2 = px[2];
        } else {
            const int ymin = coef[d.vb_off + 2 * yo], cnt = coef[d.vb_off + 2 * yo + 1];
            const int32_t* k = coef + d.vk_off + (int64_t)yo * d.vk_size;
            int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll 4
            for (int i = 0; i < cnt; ++i) {
                const int w = k[i];
                const unsigned v = load_rgbx(col + (ymin + i) * rs);
                s0 += (int)(v & 0xff) * w; s1 += (int)((v >> 8) & 0xff) * w; s2 += (int)((v >> 16) & 0xff) * w;
            }
            v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
        }
    }
    const int64_t plane = (int64_t)target_h * target_w;
    float* o = out + (int64_t)blockIdx.y * 3 * plane + idx;
    o[0] = lut[v0]; o[plane] = lut[v1]; o[2 * plane] = lut[v2];
}

}  // namespace

// ---- host: Pillow's precompute_coeffs + normalize_coeffs_8bpc, expression for expression (double, libm sin) ----
#pragma clang fp contract(off)
static double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

extern "C" int kzv_lanczos_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int* ksize_out) {
    if (in_size <= 0 || out_size <= 0 || !ksize_out) return kzv_fail(KZV_E_ARG, "lanczos_coeffs: sizes must be positive");
    const double scale0 = (double)in_size / out_size;
    double filterscale = scale0, scale = scale0;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if (!kk || !bounds) return KZV_OK;                     // size query
    const double ss = 1.0 / filterscale;
    double* pre = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = lanczos_filter((x + xmin - center + 0.5) * ss);
            pre[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) pre[x] /= ww;
        int32_t* k = kk + (int64_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            if (x >= xmax) { k[x] = 0; continue; }
            if (pre[x] < 0) k[x] = (int32_t)(-0.5 + pre[x] * (1 << PRECISION_BITS));
            else k[x] = (int32_t)(0.5 + pre[x] * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    delete[] pre;
    return KZV_OK;
}

extern "C" int kzv_preprocess_lines(const uint8_t* rgb, const kzv_line_desc* desc, const int32_t* coef, int n, int target_h,
                                    int target_w, int64_t max_tmp_pixels, const float* lut256, uint8_t* tmp, float* out, void* stream) {
    if (!rgb || !desc || !coef || !lut256 || !tmp || !out) return kzv_fail(KZV_E_ARG, "preprocess_lines: null operand");
    if (n <= 0 || n > 65535 || target_h <= 0 || target_w <= 0 || max_tmp_pixels <= 0)
        return kzv_fail(KZV_E_ARG, "preprocess_lines: batch 1..65535, positive target and max_tmp_pixels");
    hipStream_t s = (hipStream_t)stream;
    // blockIdx.y = crop; blockIdx.x covers the largest crop's work, threads beyond a crop's own extent exit at once
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((max_tmp_pixels + 255) / 256), n), dim3(256), 0, s, rgb, desc, coef, tmp);
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((target_h * target_w + 255) / 256), n), dim3(256), 0, s, desc, coef, tmp,
                       lut256, out, target_h, target_w);
    return kzv_check_launch("preprocess_lines");
}
