// HBM-bound glue kernels of the TrOCR training path (all vectorised 8/16 B per lane).
//   im2row            -- the data movement half of nn.Conv2d(k=s=patch) (src/models/trocr_model.py:77,89-90)
//   embed_assemble    -- cat(cls, patches) + position_embeddings, dropout (trocr_model.py:183-190) (+ backward)
//   cast_drop_colsum  -- backward of "dropout(linear(x))": masked bf16 copy of the upstream grad + bias grad
//   colsum_bf16       -- bias gradient of a bf16 gradient matrix
//   dec_prepare / embed_gather / embed_scatter_bwd -- RobertaEmbeddings (HF modeling_roberta.py:75-155)
//   ce_fwd_bwd        -- nn.CrossEntropyLoss(ignore_index=pad) forward + dlogits (trocr_model.py:256,292)
//   cast_weights      -- fp32 master -> bf16 W and W^T compute copies (autocast's weight cast)
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "kzv_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------ im2row
// out[(b*np + gy*gw + gx)][c*ph*pw + i*pw + j] = px[b][c][gy*ph + i][gx*pw + j]; one thread = 8 j's.
__global__ void im2row_kernel(const float* __restrict__ px, bf16_t* __restrict__ out, int B, int C, int H, int W,
                              int ph, int pw, int64_t total8) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total8) return;
    const int gw = W / pw, gh = H / ph, pw8 = pw / 8;
    const int kdim = C * ph * pw;
    int64_t r = t;
    const int j8 = r % pw8; r /= pw8;
    const int i = r % ph; r /= ph;
    const int c = r % C; r /= C;
    const int gx = r % gw; r /= gw;
    const int gy = r % gh; r /= gh;
    const int b = (int)r;
    const float* src = px + (((int64_t)b * C + c) * H + gy * ph + i) * W + gx * pw + j8 * 8;
    const float4 a = ((const float4*)src)[0], d = ((const float4*)src)[1];
    bf16_t* dst = out + ((int64_t)b * gh * gw + gy * gw + gx) * kdim + (c * ph + i) * pw + j8 * 8;
    *(uint4*)dst = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(d.x, d.y), pack_bf2(d.z, d.w));
}

// -------------------------------------------------------------------------------- embed assemble (+bwd)
// position row of token s (0 = CLS): patch p = s - 1 of a gw-wide grid sits at (p / gw, p % gw) of the gw_max-wide table
__device__ __forceinline__ int pos_row(int s, int gw, int gw_max) {
    if (s == 0 || gw == gw_max) return s;
    const int pch = s - 1;
    return 1 + (pch / gw) * gw_max + pch % gw;
}

__global__ void embed_assemble_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                      const float* __restrict__ pos, float* __restrict__ x0, int B, int np, int He,
                                      unsigned thr16, float inv_keep, unsigned key, int gw, int gw_max) {
    const int S = np + 1, h4 = He / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * S * h4) return;
    const int c = t % h4;
    const int64_t row = t / h4;
    const int s = row % S, b = (int)(row / S);
    const float4 a = s == 0 ? ((const float4*)cls)[c] : ((const float4*)(pe + ((int64_t)b * np + s - 1) * He))[c];
    const float4 p = ((const float4*)(pos + (int64_t)pos_row(s, gw, gw_max) * He))[c];
    float o0 = a.x + p.x, o1 = a.y + p.y, o2 = a.z + p.z, o3 = a.w + p.w;
    if (thr16) {
        const unsigned e = (unsigned)row * (unsigned)He + 4u * c;
        const unsigned b0 = drop_bits(key, e >> 1), b1 = drop_bits(key, (e >> 1) + 1);
        o0 *= drop_keep(b0, 0, thr16, inv_keep); o1 *= drop_keep(b0, 1, thr16, inv_keep);
        o2 *= drop_keep(b1, 0, thr16, inv_keep); o3 *= drop_keep(b1, 1, thr16, inv_keep);
    }
    ((float4*)(x0 + row * He))[c] = make_float4(o0, o1, o2, o3);
}

// Backward of the embedding assembly in three small launches, every sum in a fixed order (no atomics: bit-reproducible).
//   1. thread = (token s, 4 columns, batch chunk of EAB_CHUNK samples): masks dx0, writes dpatch (bf16), and its partial sum over the
//      chunk -> part[chunk][s][He].  (One thread per (s, columns) looping over the WHOLE batch -- the first form -- was 30k threads
//      with 256 dependent-address loads each: 145 us for 190 MB; slices of the batch with atomic sums were tried at 126 - 289 us: the
//      float atomics of many adders on the 768 bias addresses.)
//   2. thread = (s, 4 columns): adds the chunks' partials in order -> dpos (+=), dcls (+=, s = 0), and the token's total -> tok[s][He]
//   3. thread = 4 columns: patch-bias gradient += sum over the patch tokens of tok, in order
constexpr int EAB_CHUNK = 32;
__global__ void embed_assemble_bwd_part_kernel(const float* __restrict__ dx0, bf16_t* __restrict__ dpatch, float* __restrict__ part, int B, int np, int He,
                                               unsigned thr16, float inv_keep, unsigned key) {
    const int S = np + 1, h4 = He / 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * h4) return;
    const int c = t % h4, s = t / h4, chunk = blockIdx.y;
    const int b0 = chunk * EAB_CHUNK, b1 = min(B, b0 + EAB_CHUNK);
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
    for (int b = b0; b < b1; ++b) {
        const int64_t row = (int64_t)b * S + s;
        float4 d = ((const float4*)(dx0 + row * He))[c];
        if (thr16) {
            const unsigned e = (unsigned)row * (unsigned)He + 4u * c;
            const unsigned q0 = drop_bits(key, e >> 1), q1 = drop_bits(key, (e >> 1) + 1);
            d.x *= drop_keep(q0, 0, thr16, inv_keep); d.y *= drop_keep(q0, 1, thr16, inv_keep);
            d.z *= drop_keep(q1, 0, thr16, inv_keep); d.w *= drop_keep(q1, 1, thr16, inv_keep);
        }
        a0 += d.x; a1 += d.y; a2 += d.z; a3 += d.w;
        if (s > 0) ((uint2*)(dpatch + ((int64_t)b * np + s - 1) * He))[c] = make_uint2(pack_bf2(d.x, d.y), pack_bf2(d.z, d.w));
    }
    ((float4*)(part + ((int64_t)chunk * S + s) * He))[c] = make_float4(a0, a1, a2, a3);
}
__global__ void embed_assemble_bwd_tok_kernel(const float* __restrict__ part, float* __restrict__ tok, float* dcls, float* dpos, int nchunk, int np, int He,
                                              int gw, int gw_max) {
    const int S = np + 1, h4 = He / 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * h4) return;
    const int c = t % h4, s = t / h4;
    float4 a = make_float4(0, 0, 0, 0);
    for (int k = 0; k < nchunk; ++k) {
        const float4 v = ((const float4*)(part + ((int64_t)k * S + s) * He))[c];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    ((float4*)(tok + (int64_t)s * He))[c] = a;
    float* dp = dpos + (int64_t)pos_row(s, gw, gw_max) * He + 4 * c;
    dp[0] += a.x; dp[1] += a.y; dp[2] += a.z; dp[3] += a.w;
    if (s == 0) {
        float* dc = dcls + 4 * c;
        dc[0] += a.x; dc[1] += a.y; dc[2] += a.z; dc[3] += a.w;
    }
}
__global__ void embed_assemble_bwd_bias_kernel(const float* __restrict__ tok, float* dpbias, int np, int He) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= He / 4) return;
    float4 a = make_float4(0, 0, 0, 0);
    for (int s0 = 1; s0 <= np; s0 += 16) {          // 16 loads in flight, added in token order
        float4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ((const float4*)(tok + (int64_t)min(s0 + j, np) * He))[c];
#pragma unroll
        for (int j = 0; j < 16; ++j) if (s0 + j <= np) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    float* db = dpbias + 4 * c;
    db[0] += a.x; db[1] += a.y; db[2] += a.z; db[3] += a.w;
}

// ------------------------------------------------------------------------------- cast/drop + column sums
// block = 64 column-lanes (4 columns each) x 4 row-lanes; grid.y strides 64-row chunks.
template <bool IN_F32>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ gin, int64_t ld, bf16_t* __restrict__ out,
                                                     float* __restrict__ dbias, int M, int N, unsigned thr16,
                                                     float inv_keep, unsigned key, const bf16_t* __restrict__ gelu_pre) {
    __shared__ float red[4][256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cl) * 4;
    const int r0 = blockIdx.y * 64;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (col < N) {
        for (int r = r0 + rl; r < min(M, r0 + 64); r += 4) {
            float4 d;
            if (IN_F32) d = *(const float4*)((const float*)gin + (int64_t)r * ld + col);
            else {
                const uint2 u = *(const uint2*)((const bf16_t*)gin + (int64_t)r * ld + col);
                d = make_float4(bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16));
            }
            if (IN_F32) {
                if (thr16) {
                    const unsigned e = (unsigned)r * (unsigned)N + (unsigned)col;
                    const unsigned b0 = drop_bits(key, e >> 1), b1 = drop_bits(key, (e >> 1) + 1);
                    d.x *= drop_keep(b0, 0, thr16, inv_keep); d.y *= drop_keep(b0, 1, thr16, inv_keep);
                    d.z *= drop_keep(b1, 0, thr16, inv_keep); d.w *= drop_keep(b1, 1, thr16, inv_keep);
                }
                if (gelu_pre) {   // backward of GELU: multiply by the derivative the forward epilogue saved
                    const uint2 u = *(const uint2*)(gelu_pre + (int64_t)r * N + col);
                    d.x *= bf2f(u.x & 0xffff); d.y *= bf2f(u.x >> 16);
                    d.z *= bf2f(u.y & 0xffff); d.w *= bf2f(u.y >> 16);
                }
                const uint2 pk = make_uint2(pack_bf2(d.x, d.y), pack_bf2(d.z, d.w));
                *(uint2*)(out + (int64_t)r * N + col) = pk;
                // bias grad sums the values the GEMMs will actually see (bf16-rounded)
                d = make_float4(bf2f(pk.x & 0xffff), bf2f(pk.x >> 16), bf2f(pk.y & 0xffff), bf2f(pk.y >> 16));
            }
            a0 += d.x; a1 += d.y; a2 += d.z; a3 += d.w;
        }
    }
    red[rl][cl * 4 + 0] = a0; red[rl][cl * 4 + 1] = a1; red[rl][cl * 4 + 2] = a2; red[rl][cl * 4 + 3] = a3;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (dbias && c < N) atomicAdd(dbias + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------- decoder embeddings
// position ids: cumsum(ids != pad) * (ids != pad) + pad over the decoder INPUT ids labels[:, :-1];
// count = number of targets labels[:, 1:] != pad.  One thread per batch row (T <= 127).
__global__ void dec_prepare_kernel(const int64_t* __restrict__ labels, int B, int L, int T, int pad, int max_pos,
                                   int* __restrict__ posids, float* count, int* err) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int run = 0, cnt = 0;
    for (int t = 0; t < T; ++t) {
        const bool np = labels[(int64_t)b * L + t] != pad;
        run += np;
        int pid = np ? run + pad : pad;
        if (pid >= max_pos) { *err = 1; pid = max_pos - 1; }   // HF would raise an index error here
        posids[b * T + t] = pid;
        cnt += labels[(int64_t)b * L + t + 1] != pad;
    }
    if (cnt) atomicAdd(count, (float)cnt);
}

__global__ void embed_gather_kernel(const int64_t* __restrict__ labels, int L, const int* __restrict__ posids,
                                    const float* __restrict__ word, const float* __restrict__ type0,
                                    const float* __restrict__ postab, float* __restrict__ out, int B, int T, int Hd) {
    const int h4 = Hd / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * T * h4) return;
    const int c = t % h4;
    const int64_t row = t / h4;
    const int tt = row % T, b = (int)(row / T);
    const int64_t id = labels[(int64_t)b * L + tt];
    const int pid = posids[row];
    const float4 w = ((const float4*)(word + id * Hd))[c], ty = ((const float4*)type0)[c];
    const float4 p = ((const float4*)(postab + (int64_t)pid * Hd))[c];
    ((float4*)(out + row * Hd))[c] = make_float4(w.x + ty.x + p.x, w.y + ty.y + p.y, w.z + ty.z + p.z, w.w + ty.w + p.w);
}

// one wave per 16 token rows; lanes over columns.  word/pos rows with padding_idx get no gradient
// (nn.Embedding(padding_idx=pad) for BOTH tables: modeling_roberta.py:61,72-74).
__global__ __launch_bounds__(256) void embed_scatter_bwd_kernel(const float* __restrict__ dsum, const int64_t* __restrict__ labels,
                                                                int L, const int* __restrict__ posids, float* dword,
                                                                float* dtype0, float* dpostab, int B, int T, int Hd, int pad) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t rbase = ((int64_t)blockIdx.x * 4 + w) * 16;
    const int64_t rows = (int64_t)B * T;
    for (int c = lane; c < Hd; c += 64) {
        float acc = 0.f;
        for (int r = 0; r < 16; ++r) {
            const int64_t row = rbase + r;
            if (row >= rows) break;
            const float v = dsum[row * Hd + c];
            acc += v;
            const int tt = row % T, b = (int)(row / T);
            const int64_t id = labels[(int64_t)b * L + tt];
            const int pid = posids[row];
            if (id != pad) atomicAdd(dword + id * Hd + c, v);
            if (pid != pad) atomicAdd(dpostab + (int64_t)pid * Hd + c, v);
        }
        atomicAdd(dtype0 + c, acc);
    }
}

// ------------------------------------------------------------------------------------------ cross entropy
// one workgroup per token row; target = labels[b][t+1]; rows whose target is pad contribute nothing and
// get a zero dlogits row.  dlogits = (softmax - onehot) / count, bf16, pad columns [V, ldl) zeroed.
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ labels,
                                                 int L, int T, int V, int pad, const float* __restrict__ count,
                                                 float* loss, bf16_t* __restrict__ dlogits) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = row / T, t = row - b * T;
    const int64_t tgt = labels[(int64_t)b * L + t + 1];
    const float* lr = logits + (int64_t)row * ldl;
    bf16_t* dr = dlogits ? dlogits + (int64_t)row * ldl : nullptr;
    const int n4 = (int)(ldl / 4);
    if (tgt == pad) {
        if (dr) for (int i = tid; i < n4; i += 256) ((uint2*)dr)[i] = make_uint2(0, 0);
        return;
    }
    // the row lives in registers (ldl / 4 <= 256 * RV float4 pieces): read once, 16 bytes per lane
    constexpr int RV = 5;                                  // 5 * 256 * 4 = 5120 columns
    float4 v[RV];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < RV; ++j) {
        const int i = tid + j * 256;
        v[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (i < n4) {
            v[j] = ((const float4*)lr)[i];
            if (4 * i + 1 > V) v[j].x = -INFINITY;      // pad columns [V, ldl) hold zeros, not logits
            if (4 * i + 2 > V) v[j].y = -INFINITY;
            if (4 * i + 3 > V) v[j].z = -INFINITY;
            if (4 * i + 4 > V) v[j].w = -INFINITY;
        }
        mx = fmaxf(fmaxf(mx, fmaxf(v[j].x, v[j].y)), fmaxf(v[j].z, v[j].w));
    }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < RV; ++j) {
        v[j].x = __expf(v[j].x - mx); v[j].y = __expf(v[j].y - mx); v[j].z = __expf(v[j].z - mx); v[j].w = __expf(v[j].w - mx);
        se += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    se = wave_sum(se);
    if (lane == 0) red[4 + w] = se;
    __syncthreads();
    se = red[4] + red[5] + red[6] + red[7];
    const float lse = mx + __logf(se);
    const float inv_cnt = 1.f / *count;
    if (tid == 0) atomicAdd(loss, (lse - lr[tgt]) * inv_cnt);
    if (dr) {
        const float sc = inv_cnt / se;                     // softmax = exp(x - mx) / se
#pragma unroll
        for (int j = 0; j < RV; ++j) {
            const int i = tid + j * 256;
            if (i < n4) {
                float o[4] = {v[j].x * sc, v[j].y * sc, v[j].z * sc, v[j].w * sc};
                const int64_t d = tgt - 4 * (int64_t)i;
                if (d >= 0 && d < 4) o[d] -= inv_cnt;
                ((uint2*)dr)[i] = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}

// the same for rows wider than the registers of ce_kernel hold (> 5120 columns): three passes over the (L2-resident) row
// one workgroup per token row; target = labels[b][t+1]; rows whose target is pad contribute nothing and
// get a zero dlogits row.  dlogits = (softmax - onehot) / count, bf16, pad columns [V, ldl) zeroed.
__global__ __launch_bounds__(256) void ce_generic_kernel(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ labels,
                                                 int L, int T, int V, int pad, const float* __restrict__ count,
                                                 float* loss, bf16_t* __restrict__ dlogits) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = row / T, t = row - b * T;
    const int64_t tgt = labels[(int64_t)b * L + t + 1];
    const float* lr = logits + (int64_t)row * ldl;
    bf16_t* dr = dlogits ? dlogits + (int64_t)row * ldl : nullptr;
    const int n4 = (int)(ldl / 4);
    if (tgt == pad) {
        if (dr) for (int i = tid; i < n4; i += 256) ((uint2*)dr)[i] = make_uint2(0, 0);
        return;
    }
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, lr[i]);
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
    for (int i = tid; i < V; i += 256) se += __expf(lr[i] - mx);
    se = wave_sum(se);
    if (lane == 0) red[4 + w] = se;
    __syncthreads();
    se = red[4] + red[5] + red[6] + red[7];
    const float lse = mx + __logf(se);
    const float inv_cnt = 1.f / *count;
    if (tid == 0) atomicAdd(loss, (lse - lr[tgt]) * inv_cnt);
    if (dr) {
        for (int i = tid; i < n4; i += 256) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 4 * i + r;
                v[r] = col < V ? (__expf(lr[col] - lse) - (col == tgt ? 1.f : 0.f)) * inv_cnt : 0.f;
            }
            ((uint2*)dr)[i] = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
    }
}

__global__ void copy_logits_kernel(const float* __restrict__ logits, int64_t ldl, float* __restrict__ out, int64_t total, int V) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t row = t / V;
    const int c = (int)(t - row * V);
    out[t] = logits[row * ldl + c];
}

// ------------------------------------------------------------------------------------------ weight casts
// 64x64 tile per workgroup; binary search of the tile index in the descriptor table.
__global__ __launch_bounds__(256) void cast_weights_kernel(const KzvCastDesc* __restrict__ desc, int ndesc) {
    __shared__ float tile[64][65];
    int lo = 0, hi = ndesc - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].tile0 <= bid) lo = mid; else hi = mid - 1;
    }
    const KzvCastDesc d = desc[lo];
    const int tl = bid - d.tile0;
    const int tr = tl / d.tiles_c, tc = tl - tr * d.tiles_c;
    const int r0 = tr * 64, c0 = tc * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
        float4 v = make_float4(0, 0, 0, 0);
        if (r < d.rows && c < d.cols) {           // cols % 4 == 0 is checked on the host
            v = *(const float4*)(d.src + (int64_t)r * d.cols + c);
            *(uint2*)(d.dst + (int64_t)r * d.cols + c) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
        }
        tile[ty + 16 * i][tx * 4 + 0] = v.x; tile[ty + 16 * i][tx * 4 + 1] = v.y;
        tile[ty + 16 * i][tx * 4 + 2] = v.z; tile[ty + 16 * i][tx * 4 + 3] = v.w;
    }
    if (!d.dstT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 16 * i, r = r0 + tx * 4;   // output row = source column
        if (c < d.cols && r < d.rows) {                     // rows % 4 == 0 checked on the host
            const int lc = ty + 16 * i, lr = tx * 4;
            *(uint2*)(d.dstT + (int64_t)c * d.ldT + r) =
                make_uint2(pack_bf2(tile[lr][lc], tile[lr + 1][lc]), pack_bf2(tile[lr + 2][lc], tile[lr + 3][lc]));
        }
    }
}

// ------------------------------------------------------------------------------------------- fp8 path
// One wave per row: amax, then e4m3(x * 448 / amax) and scale = amax / 448 (a zero row keeps scale 1).  `desc` == nullptr:
// the single matrix `one`; otherwise the row -> matrix table (row0 ascending), binary-searched like cast_weights.
__global__ __launch_bounds__(256) void quant_rows_kernel(const KzvQuantDesc* __restrict__ desc, int ndesc, const KzvQuantDesc one, int total_rows) {
    const int lane = threadIdx.x & 63;
    const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grow >= total_rows) return;
    KzvQuantDesc d = one;
    if (desc) {
        int lo = 0, hi = ndesc - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (desc[mid].row0 <= grow) lo = mid; else hi = mid - 1;
        }
        d = desc[lo];
    }
    const int r = grow - d.row0;
    const int nc = d.cols >> 2;                       // cols % 4 == 0 is checked on the host
    const float4* x = d.src ? (const float4*)(d.src + (int64_t)r * d.cols) : nullptr;
    const uint2* x16 = d.src ? nullptr : (const uint2*)(d.src16 + (int64_t)r * d.ld16);
    auto at = [&](int i) {
        if (x) return x[i];
        const uint2 u = x16[i];
        return make_float4(bf2f(u.x & 0xffff), bf2f(u.x >> 16), bf2f(u.y & 0xffff), bf2f(u.y >> 16));
    };
    float amax = 0.f, ssq = 0.f;
    for (int i = lane; i < nc; i += 64) {
        const float4 v = at(i);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        ssq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    amax = wave_max(amax);
    const float qs = amax > 0.f ? KZV_FP8_MAX / amax : 1.f;
    if (d.normmax) {
        ssq = wave_sum(ssq);
        if (lane == 0) atomicMax((unsigned*)d.normmax, __float_as_uint(sqrtf(ssq)));     // floats >= 0 order as integers
    }
    if (lane == 0) d.scale[r] = amax > 0.f ? amax / KZV_FP8_MAX : 1.f;
    unsigned* q = (unsigned*)(d.dst + (int64_t)r * d.cols);
    for (int i = lane; i < nc; i += 64) {
        const float4 v = at(i);                       // second read of a row this wave just streamed: L2 / L1 hit
        q[i] = pack_fp8x4(v.x * qs, v.y * qs, v.z * qs, v.w * qs);
    }
}

__global__ void fp8_roll_kernel(float* qscale, float* amax, float* row_scales, int sites, int rows) {
    const int site = blockIdx.y;
    float q = qscale[site];
    const float a = amax[site];
    if (a > 0.f) q = exp2f(floorf(log2f(KZV_FP8_MAX / a)) - 1.f);
    const float inv = 1.f / q;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) row_scales[(int64_t)site * rows + r] = inv;
}
// publishing the new multiplier in the kernel that still reads the old one would race across blocks: a second tiny kernel
__global__ void fp8_roll_publish_kernel(float* qscale, float* amax, int sites) {
    const int site = blockIdx.x * blockDim.x + threadIdx.x;
    if (site >= sites) return;
    const float a = amax[site];
    if (a > 0.f) qscale[site] = exp2f(floorf(log2f(KZV_FP8_MAX / a)) - 1.f);
    amax[site] = 0.f;
}

// debug / parity: the multiplier every fused dropout applies to element index row * ld + col
__global__ void dropout_mask_kernel(float* __restrict__ out, int64_t rows, int64_t cols, int64_t ld, unsigned thr16, float inv_keep, unsigned key) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * cols) return;
    const int64_t r = t / cols, c = t - r * cols;
    const unsigned e = (unsigned)r * (unsigned)ld + (unsigned)c;
    out[t] = thr16 ? drop_keep(drop_bits(key, e >> 1), e & 1, thr16, inv_keep) : 1.f;
}

// same for an attention-probability site: rows = (batch * heads + head) * Sq + q, the 4 x 4-block generator of kzv_common.h
__global__ void attn_dropout_mask_kernel(float* __restrict__ out, int64_t pairs, int Sq, int Sk, unsigned thr16, float inv_keep, unsigned key) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pairs * Sq * Sk) return;
    const int64_t row = t / Sk;
    const int k = (int)(t - row * Sk), q = (int)(row % Sq);
    const unsigned pair = (unsigned)(row / Sq);
    const unsigned block = (pair * ((unsigned)(Sq + 3) >> 2) + ((unsigned)q >> 2)) * ((unsigned)(Sk + 3) >> 2) + ((unsigned)k >> 2);
    out[t] = !thr16 ? 1.f : att_keep1(key, block, q & 3, k & 3, (int)thr16) ? inv_keep : 0.f;
}

inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

}  // namespace

int kzv_im2row(const float* px, bf16_t* out, int B, int C, int H, int W, int ph, int pw, hipStream_t s) {
    if (pw % 8 || W % pw || H % ph) return kzv_fail(KZV_E_ARG, "im2row: patch width must be a multiple of 8 and divide the image");
    const int64_t total8 = (int64_t)B * C * H * W / 8;
    hipLaunchKernelGGL(im2row_kernel, dim3(nblk(total8, 256)), dim3(256), 0, s, px, out, B, C, H, W, ph, pw, total8);
    return kzv_check_launch("im2row");
}

int kzv_embed_assemble(const float* pe, const float* cls, const float* pos, float* x0, int B, int np, int He,
                       float drop_p, uint32_t key, hipStream_t s, int gw, int gw_max) {
    unsigned thr; float ik; kzv_drop_params(drop_p, &thr, &ik);
    const int64_t total = (int64_t)B * (np + 1) * (He / 4);
    if (gw <= 0 || gw_max <= 0) gw = gw_max = 1;
    hipLaunchKernelGGL(embed_assemble_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, pe, cls, pos, x0, B, np, He, thr, ik, key, gw, gw_max);
    return kzv_check_launch("embed_assemble");
}

int kzv_embed_assemble_bwd(const float* dx0, bf16_t* dpatch, float* dcls, float* dpos, float* dpbias, int B, int np, int He,
                           float drop_p, uint32_t key, hipStream_t s, int gw, int gw_max) {
    unsigned thr; float ik; kzv_drop_params(drop_p, &thr, &ik);
    const int S = np + 1, total = S * (He / 4), nchunk = (B + EAB_CHUNK - 1) / EAB_CHUNK;
    if (gw <= 0 || gw_max <= 0) gw = gw_max = 1;
    // partial sums [nchunk + 1][S][He] fp32: a process-global scratch grown on demand (calls are stream-ordered)
    static float* scratch = nullptr; static size_t scratch_floats = 0;
    const size_t need = (size_t)(nchunk + 1) * S * He;
    if (need > scratch_floats) {
        if (scratch) { (void)hipDeviceSynchronize(); (void)hipFree(scratch); scratch = nullptr; scratch_floats = 0; }
        if (hipMalloc((void**)&scratch, need * sizeof(float)) != hipSuccess) return kzv_fail(KZV_E_HIP, "embed_assemble_bwd: scratch (%zu bytes)", need * sizeof(float));
        scratch_floats = need;
    }
    float* tok = scratch + (size_t)nchunk * S * He;
    hipLaunchKernelGGL(embed_assemble_bwd_part_kernel, dim3(nblk(total, 64), nchunk), dim3(64), 0, s, dx0, dpatch, scratch, B, np, He, thr, ik, key);
    hipLaunchKernelGGL(embed_assemble_bwd_tok_kernel, dim3(nblk(total, 64)), dim3(64), 0, s, scratch, tok, dcls, dpos, nchunk, np, He, gw, gw_max);
    hipLaunchKernelGGL(embed_assemble_bwd_bias_kernel, dim3(nblk(He / 4, 64)), dim3(64), 0, s, tok, dpbias, np, He);
    return kzv_check_launch("embed_assemble_bwd");
}

int kzv_cast_drop_colsum(const float* g, bf16_t* out, float* dbias, int M, int N, float drop_p, uint32_t key, hipStream_t s,
                         const bf16_t* gelu_pre) {
    if (N % 4) return kzv_fail(KZV_E_ARG, "cast_drop_colsum: N %% 4");
    unsigned thr; float ik; kzv_drop_params(drop_p, &thr, &ik);
    hipLaunchKernelGGL(colsum_kernel<true>, dim3((N + 255) / 256, (M + 63) / 64), dim3(256), 0, s, (const void*)g, (int64_t)N, out, dbias, M, N, thr, ik, key, gelu_pre);
    return kzv_check_launch("cast_drop_colsum");
}

int kzv_colsum_bf16(const bf16_t* g, int64_t ld, float* dbias, int M, int N, hipStream_t s) {
    if (N % 4 || ld % 4) return kzv_fail(KZV_E_ARG, "colsum_bf16: N, ld %% 4");
    hipLaunchKernelGGL(colsum_kernel<false>, dim3((N + 255) / 256, (M + 63) / 64), dim3(256), 0, s, (const void*)g, ld, (bf16_t*)nullptr, dbias, M, N, 0u, 1.f, 0u, (const bf16_t*)nullptr);
    return kzv_check_launch("colsum_bf16");
}

int kzv_dec_prepare(const int64_t* labels, int B, int L, int T, int pad, int max_pos, int* posids, float* count, int* err, hipStream_t s) {
    hipLaunchKernelGGL(dec_prepare_kernel, dim3(nblk(B, 64)), dim3(64), 0, s, labels, B, L, T, pad, max_pos, posids, count, err);
    return kzv_check_launch("dec_prepare");
}

int kzv_embed_gather(const int64_t* labels, int L, const int* posids, const float* word, const float* type0,
                     const float* postab, float* out, int B, int T, int Hd, hipStream_t s) {
    const int64_t total = (int64_t)B * T * (Hd / 4);
    hipLaunchKernelGGL(embed_gather_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, labels, L, posids, word, type0, postab, out, B, T, Hd);
    return kzv_check_launch("embed_gather");
}

int kzv_embed_scatter_bwd(const float* dsum, const int64_t* labels, int L, const int* posids, float* dword, float* dtype0,
                          float* dpostab, int B, int T, int Hd, int pad, hipStream_t s) {
    const int64_t rows = (int64_t)B * T;
    hipLaunchKernelGGL(embed_scatter_bwd_kernel, dim3(nblk(rows, 64)), dim3(256), 0, s, dsum, labels, L, posids, dword, dtype0, dpostab, B, T, Hd, pad);
    return kzv_check_launch("embed_scatter_bwd");
}

int kzv_ce_fwd_bwd(const float* logits, int64_t ldl, const int64_t* labels, int L, int B, int T, int V, int pad,
                   const float* count, float* loss, bf16_t* dlogits, hipStream_t s) {
    if (ldl % 4) return kzv_fail(KZV_E_ARG, "ce: ldl %% 4");
    if (ldl <= 5120) hipLaunchKernelGGL(ce_kernel, dim3(B * T), dim3(256), 0, s, logits, ldl, labels, L, T, V, pad, count, loss, dlogits);
    else hipLaunchKernelGGL(ce_generic_kernel, dim3(B * T), dim3(256), 0, s, logits, ldl, labels, L, T, V, pad, count, loss, dlogits);
    return kzv_check_launch("ce_fwd_bwd");
}

int kzv_copy_logits(const float* logits, int64_t ldl, float* out, int rows, int V, hipStream_t s) {
    const int64_t total = (int64_t)rows * V;
    hipLaunchKernelGGL(copy_logits_kernel, dim3(nblk(total, 256)), dim3(256), 0, s, logits, ldl, out, total, V);
    return kzv_check_launch("copy_logits");
}

int kzv_cast_weights(const KzvCastDesc* d_desc, int ndesc, int total_tiles, hipStream_t s) {
    hipLaunchKernelGGL(cast_weights_kernel, dim3(total_tiles), dim3(256), 0, s, d_desc, ndesc);
    return kzv_check_launch("cast_weights");
}

int kzv_quant_rows(const KzvQuantDesc* d_desc, int ndesc, int total_rows, hipStream_t s) {
    if (!d_desc || ndesc <= 0 || total_rows <= 0) return kzv_fail(KZV_E_ARG, "quant_rows: empty table");
    hipLaunchKernelGGL(quant_rows_kernel, dim3((total_rows + 3) / 4), dim3(256), 0, s, d_desc, ndesc, KzvQuantDesc{}, total_rows);
    return kzv_check_launch("quant_rows");
}

extern "C" int kzv_quant_rows_fp8(const float* x, int64_t rows, int64_t cols, void* q, float* scale, void* stream) {
    if (!x || !q || !scale || rows <= 0 || cols <= 0 || rows > 0x7fffffff) return kzv_fail(KZV_E_ARG, "quant_rows_fp8: null/empty");
    if (cols % 4 || ((uintptr_t)x & 15) || ((uintptr_t)q & 3)) return kzv_fail(KZV_E_ARG, "quant_rows_fp8: cols must be a multiple of 4, x 16-byte aligned");
    const KzvQuantDesc one{x, (unsigned char*)q, scale, (int)rows, (int)cols, 0, nullptr, 0, nullptr};
    hipLaunchKernelGGL(quant_rows_kernel, dim3(((int)rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const KzvQuantDesc*)nullptr, 0, one, (int)rows);
    return kzv_check_launch("quant_rows_fp8");
}

int kzv_fp8_roll(float* qscale, float* amax, float* row_scales, int sites, int rows, hipStream_t s) {
    if (!qscale || !amax || !row_scales || sites <= 0 || rows <= 0) return kzv_fail(KZV_E_ARG, "fp8_roll: null/empty");
    hipLaunchKernelGGL(fp8_roll_kernel, dim3(nblk(rows, 256) < 64 ? nblk(rows, 256) : 64, sites), dim3(256), 0, s, qscale, amax, row_scales, sites, rows);
    hipLaunchKernelGGL(fp8_roll_publish_kernel, dim3(nblk(sites, 64)), dim3(64), 0, s, qscale, amax, sites);
    return kzv_check_launch("fp8_roll");
}

extern "C" int kzv_debug_attn_dropout_mask(uint32_t key, float p, int64_t pairs, int32_t Sq, int32_t Sk, float* d_out, void* stream) {
    if (!d_out || pairs < 0 || Sq <= 0 || Sk <= 0) return kzv_fail(KZV_E_ARG, "debug_attn_dropout_mask: bad shape");
    if (pairs == 0) return KZV_OK;
    unsigned thr; float ik; kzv_drop_params(p, &thr, &ik);
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(nblk(pairs * Sq * Sk, 256)), dim3(256), 0, (hipStream_t)stream, d_out, pairs, Sq, Sk, thr, ik, key);
    return kzv_check_launch("debug_attn_dropout_mask");
}

extern "C" int kzv_debug_dropout_mask(uint32_t key, float p, int64_t rows, int64_t cols, int64_t ld_index, float* d_out, void* stream) {
    if (!d_out || rows < 0 || cols < 0 || ld_index < cols) return kzv_fail(KZV_E_ARG, "debug_dropout_mask: bad shape");
    if (rows * cols == 0) return KZV_OK;
    unsigned thr; float ik; kzv_drop_params(p, &thr, &ik);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(nblk(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, d_out, rows, cols, ld_index, thr, ik, key);
    return kzv_check_launch("debug_dropout_mask");
}
