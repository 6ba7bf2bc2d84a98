// N1: the KV-cached generation step of the whole decoder in ONE launch.
//
// Replaces, per new token, the embedding gather + 6 x (QKV, cached self-attention, output projection, cross query, cross-attention,
// output projection, fc1 + GELU, fc2) + the LM head's dense layer of HF RobertaForCausalLM with `use_cache=True`
// (modeling_roberta.py:75-122, 186-326, 421-464, 877-893 as driven by src/models/trocr_model.py:306-316, num_beams = 4): 50 launches
// of 5-15 us became one.
//
// Every operation of a decoder step is local to a sequence -- the linears are row-wise, a sequence attends to its own cache and to
// its own image's patch keys -- so ONE workgroup owns the G sequences of one image (G = beams) through all layers and never talks
// to another workgroup: no grid barrier, no flags.  Hidden size 256, 4 heads of 64, FFN 768 (the reference decoder), <= 128 cached
// keys, <= 160 patch keys; other geometries keep the launch-per-operation path (model.cpp).
//
// What bounds it is ONE CU's load path (~60 GB/s from L2): per token a workgroup streams all 9.4 MB of decoder weights plus its
// sequences' attention rows.  So the eight waves are split by what they LOAD, and each stream stays in flight across the barriers:
//   * waves 0-3, the linears: the 16-row MFMA tile holds the G rows (lanes of rows >= G repeat row G - 1); wave w owns the
//     16-column blocks w, w + 4, ... and reads its weight fragments in MFMA fragment order (pack_frag_kernel: a wave-instruction is one
//     contiguous KiB), a window of 16 fragments always in flight and refilled with the NEXT linear's first fragments as a linear ends.
//   * waves 4-7, attention, one head each: key / value rows go to registers (a wave-instruction = 8 rows x 128 B) and are requested
//     one phase early -- the cached rows of the first sequence while the QKV projection runs, the image's patch keys / values
//     (shared by the beams) while the output projection, LayerNorm and cross query run -- so the HBM latency of one stream hides
//     behind the L2 stream of the other; between the two they also fetch the next layer's biases and LayerNorm weights into LDS.
//   * a vector-memory counter retires in issue order, which is why the two kinds of load cannot share a wave: a weight fragment
//     issued behind an HBM row would wait for it.
#include "kzv_common.h"
#include "../../include/kzv.h"
#include "kzv_host.h"
#include "kzv_kernels.h"
#include <cmath>

namespace {

constexpr int HD = 256, NH = 4, FD = 768, TMAX = 128, NPMAX = 160;
constexpr int LDX = HD + 4;          // fp32 rows of the residual stream in LDS
constexpr int LDH = HD + 8;          // bf16 rows, 256 wide (16-byte aligned, 4 banks apart)
constexpr int LDW = FD + 8;          // bf16 rows, 768 wide (qkv / FFN activation)
// one layer's biases and LayerNorm weights in LDS (floats)
constexpr int PB_QKV = 0, PB_O = 768, PB_CQ = 1024, PB_CO = 1280, PB_FC1 = 1536, PB_FC2 = 2304;
constexpr int PL1W = 2560, PL1B = 2816, PL2W = 3072, PL2B = 3328, PL3W = 3584, PL3B = 3840, PAR = 4096;
constexpr int NR = 40;               // role registers: 40 fragments (160 VGPRs): the weight window, or key + value rows
constexpr int WIN = 24;
constexpr int NIS = TMAX / 8, NIC = NPMAX / 8;

// the lane id, re-derived per phase: hipcc otherwise hoists every lane-derived offset of all phases out of the layer loop and spills them
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// Workgroup barrier for LDS hand-offs that leaves global loads in flight: __syncthreads() is a workgroup-scope release, and on gfx9
// that means s_waitcnt vmcnt(0) -- it would drain both streams at every phase boundary.  Nothing a phase writes to global memory
// is read by another wave of the same launch.
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sums / maxima over lane groups without the LDS crossbar (ds_bpermute, what __shfl_xor compiles to): DPP moves for the steps inside a
// 16-lane row, the crossbar only for the two steps across rows.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 8 lanes of a key row: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (the other quad's sum)
__device__ __forceinline__ float sum8(float a) { a += dpp_mov<0xB1>(a); a += dpp_mov<0x4E>(a); a += dpp_mov<0x141>(a); return a; }
// sum over the 8 key rows of a wave-instruction (lanes with equal lane & 7): row_ror:8, then the two cross-row steps
__device__ __forceinline__ float sum_rows(float a) { a += dpp_mov<0x128>(a); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64); return a; }
__device__ __forceinline__ float wave_sum_d(float a) { return sum_rows(sum8(a)); }
__device__ __forceinline__ float wave_max_d(float a) {
    a = fmaxf(a, dpp_mov<0xB1>(a)); a = fmaxf(a, dpp_mov<0x4E>(a)); a = fmaxf(a, dpp_mov<0x141>(a)); a = fmaxf(a, dpp_mov<0x128>(a));
    a = fmaxf(a, __shfl_xor(a, 16, 64)); a = fmaxf(a, __shfl_xor(a, 32, 64));
    return a;
}

// q . k over this lane's 8 dimensions in fp32 (products of bf16 pairs are exact, eight fp32 accumulations); qp = the query pre-scaled
// by 1 / 8 (exact in bf16), two dimensions per dword.  NOT v_dot2c_f32_bf16: four of those per key are 4x fewer instructions, and
// moved the step's logits 4 - 6x further from the teacher-forced recompute (3e-2 against 5e-3 twelve layers deep,
// tests/test_decode_fused_gpu.py) -- its accumulation is not an fp32 FMA chain.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ float dot8(const bf16x8& k, const u32x4_t& qp) {
    const u32x4_t kd = __builtin_bit_cast(u32x4_t, k);
    float a = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a += __builtin_bit_cast(float, kd[e] << 16) * __builtin_bit_cast(float, qp[e] << 16);
        a += __builtin_bit_cast(float, kd[e] & 0xffff0000u) * __builtin_bit_cast(float, qp[e] & 0xffff0000u);
    }
    return a;
}
__device__ __forceinline__ u32x4_t scaled_query(const bf16_t* q) {
    const bf16x8 q8 = *(const bf16x8*)q;
    return (u32x4_t){pack_bf2(bf2f((bf16_t)q8[0]) * 0.125f, bf2f((bf16_t)q8[1]) * 0.125f), pack_bf2(bf2f((bf16_t)q8[2]) * 0.125f, bf2f((bf16_t)q8[3]) * 0.125f),
                     pack_bf2(bf2f((bf16_t)q8[4]) * 0.125f, bf2f((bf16_t)q8[5]) * 0.125f), pack_bf2(bf2f((bf16_t)q8[6]) * 0.125f, bf2f((bf16_t)q8[7]) * 0.125f)};
}

#ifdef KZV_STAMPS      // diagnostic build (tools/dev/stamps_decode.py): the phase timeline of workgroup 0, layer 0 and the last layer
__device__ long long g_df_stamps[64];
#define DF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_df_stamps[k] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define DF_STAMP_L(k) do { if (li == 0) DF_STAMP(2 + (k)); else if (li == p.nlayers - 1) DF_STAMP(20 + (k)); } while (0)
#define DF_STAMP_A(k) do { if (blockIdx.x == 0 && threadIdx.x == 256 && li == 1) g_df_stamps[44 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define DF_STAMP_A(k)
#define DF_STAMP(k)
#define DF_STAMP_L(k)
#endif

template <int G>
struct Lds {
    float xs[G][LDX];                // x = LN(s): the residual stream (fp32)
    float ss[G][LDX];                // a sub-layer's output before its LayerNorm
    bf16_t ab[G][LDH];               // x as the next GEMM's A operand
    bf16_t wide[G][LDW];             // q | k | v of this step, then the cross query, then the FFN activation
    bf16_t ctx[G][LDH];              // attention output
    float par[2][PAR];               // biases + LayerNorm weights of this layer / the next one
    int rt[G][TMAX];                 // beam row table of the G sequences (cache row of the ancestor that wrote key j)
    unsigned char vf[G][TMAX];       // usable-key flags
};

// ---- the weight stream (waves 0-3) --------------------------------------------------------------------------------------------
// acc[c][r] = sum_k W[n][k] * a[m][k] for m = lane & 15 (rows >= G repeat row G - 1), n = (w + 4 c) * 16 + 4 (lane >> 4) + r.
// Wp = the weight in FRAGMENT ORDER: the 64 lanes' 16-byte pieces of (16-column block nb, 32-deep k-step ks) are 1 KiB of contiguous
// memory at ((nb * KS + ks) * 64 + lane) * 16 bytes.  (Read from the row-major [N, K] copy, the 64 lanes of a fragment are 64
// separate 16-byte requests -- 16 rows x 4 pieces -- and the address coalescer, at one request per cycle, held the stream at 30 GB/s
// per CU: 13 us for the 393 KB of a QKV projection against 6.6.)  Fragment i of a CB x KS GEMM is (k-step i / CB, column block i % CB);
// slot i % WIN of the window is refilled right behind the MFMA that consumed it, with this GEMM's fragment i + WIN or, past its end,
// with the NEXT GEMM's first fragments.
template <int CB, int KS>
__device__ __forceinline__ const char* wave_frags(const bf16_t* Wp, int w) { return (const char*)(Wp + (int64_t)w * KS * 512); }
template <int CB, int KS>
__device__ __forceinline__ bf16x8 ld_frag(const char* wb, unsigned wo, int i) {
    const int ks = i / CB, c = i % CB;
    return *(const bf16x8*)(wb + ((int64_t)(4 * c) * KS + ks) * 1024 + wo);
}
template <int CB, int KS>
__device__ __forceinline__ void fill_window(bf16x8 (&R)[NR], const char* wb, int lane) {
    const unsigned wo = (unsigned)lane * 16u;
#pragma unroll
    for (int i = 0; i < WIN; ++i) R[i] = ld_frag<CB, KS>(wb, wo, i);
}
template <int CB, int KS, int NCB, int NKS, int OFF, int G>      // OFF: window slot of this GEMM's fragment 0 (the stream's fragments take the slots in turn)
__device__ __forceinline__ void rows_gemm(bf16x8 (&R)[NR], const char* wb, const char* next, const bf16_t* a_lds, int lda, int lane, f32x4 (&acc)[CB]) {
    constexpr int F = CB * KS;
    static_assert(F >= WIN && NCB * NKS >= WIN && WIN <= NR, "rows_gemm: window");
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned wo = (unsigned)lane * 16u;
    const bf16_t* ap = a_lds + min(l15, G - 1) * lda + g * 8;
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa;
#pragma unroll
    for (int i = 0; i < F; ++i) {
        const int ks = i / CB, c = i % CB, slot = (OFF + i) % WIN;
        if (c == 0) fa = *(const bf16x8*)(ap + ks * 32);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(R[slot], fa, acc[c], 0, 0, 0);
        if (i + WIN < F) R[slot] = ld_frag<CB, KS>(wb, wo, i + WIN);
        else if (next) R[slot] = ld_frag<NCB, NKS>(next, wo, i + WIN - F);
    }
}

// W [N, K] row-major -> fragment order (see rows_gemm); one thread per 16-byte piece
__global__ void pack_frag_kernel(const uint4* __restrict__ W, uint4* __restrict__ out, int N, int K) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N * K / 8) return;
    const int lane = t & 63, f = t >> 6, KS = K / 32;
    const int nb = f / KS, ks = f - nb * KS;
    out[t] = W[((int64_t)(nb * 16 + (lane & 15)) * K + ks * 32 + (lane >> 4) * 8) / 8];
}

// LayerNorm of one row (fp32, LDS) by one wave: the normalised row goes to xs (fp32) and ab (bf16); gamma / beta from LDS
__device__ __forceinline__ void ln_row(const float* src, const float* gamma, const float* beta, float eps, float* xs, bf16_t* ab, int lane) {
    const float4 v = *(const float4*)(src + lane * 4);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / HD);
    const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
    const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) * (1.f / HD) + eps);
    const float4 ga = *(const float4*)(gamma + lane * 4), be = *(const float4*)(beta + lane * 4);
    const float y0 = a * rstd * ga.x + be.x, y1 = b * rstd * ga.y + be.y, y2 = c * rstd * ga.z + be.z, y3 = d * rstd * ga.w + be.w;
    *(float4*)(xs + lane * 4) = make_float4(y0, y1, y2, y3);
    *(uint2*)(ab + lane * 4) = make_uint2(pack_bf2(y0, y1), pack_bf2(y2, y3));
}

struct FusedLayer {
    const bf16_t *wqkv, *wo, *wcq, *wco, *wfc1, *wfc2;
    const float *bqkv, *bo, *bcq, *bco, *bfc1, *bfc2;
    const float *ln1w, *ln1b, *ln2w, *ln2b, *ln3w, *ln3b;
};
struct FusedP {
    FusedLayer L[KZV_DECODE_FUSED_MAX_LAYERS];
    int nlayers;
    const int64_t* tokens; const int* posids;
    const float *word, *type0, *postab, *elnw, *elnb;
    const bf16_t* whd; const float* bhd; float* hd_out;         // LM head dense + GELU -> fp32 [B, 256]
    bf16_t* cache; int64_t plane;                               // self-attention K / V planes [2 * layer (+1)][B][head][T][64]
    const bf16_t* ckv; int64_t plane2;                          // cross K / V planes [2 * layer (+1)][image][head][npa][64]
    const unsigned char* valid; int64_t ldvalid;
    const int* tptr; int t, T, npa, B;
    int* rows;                                                  // beam row table [B][T] or null
    float eps;
};

// ---- the attention streams (waves 4-7, head h each) ------------------------------------------------------------------------------
// lane = (row r = lane >> 3 of an 8-key group, 16-byte piece c = lane & 7); iteration i holds key 8 i + r.
// cached keys -> R[0 .. NIS), values -> R[NIS .. 2 NIS) for sequence g.  Every load is unconditional and clamped to a written
// position: this step's own key / value comes from LDS and positions past it are dropped in self_compute.
template <int G, bool KEYS, bool VALUES>
__device__ __forceinline__ void self_load(bf16x8 (&R)[NR], const Lds<G>& sm, const FusedP& p, int li, int g, int h, int b, int tdev, int lane) {
    const int r = lane >> 3, c = lane & 7;
    const unsigned kb = (unsigned)p.T * HD;
    const char* Kl = (const char*)(p.cache + (int64_t)(2 * li) * p.plane + (int64_t)h * p.T * 64);
    const char* Vl = (const char*)(p.cache + (int64_t)(2 * li + 1) * p.plane + (int64_t)h * p.T * 64);
#pragma unroll
    for (int i = 0; i < NIS; ++i) {
        const int j = 8 * i + r;
        const int rowi = (p.rows && j < tdev) ? sm.rt[g][j] : b;
        const unsigned off = ((unsigned)rowi * kb + (unsigned)min(j, tdev) * 64u + (unsigned)c * 8u) * 2u;     // < 2^32: checked on the host
        if (KEYS) R[i] = *(const bf16x8*)(Kl + off);
        if (VALUES) R[NIS + i] = *(const bf16x8*)(Vl + off);
    }
}
template <int G>
__device__ __forceinline__ void self_compute(bf16x8 (&R)[NR], Lds<G>& sm, const FusedP& p, int li, int g, int h, int b, int tdev, int lane, bool more) {
    const int r = lane >> 3, c = lane & 7;
    const bf16_t* fresh_k = &sm.wide[g][HD + h * 64];
    const bf16_t* fresh_v = &sm.wide[g][2 * HD + h * 64];
    {   // this step's key / value join the cache (used from LDS below, never read back from there in this launch)
        bf16_t* Kl = p.cache + (int64_t)(2 * li) * p.plane + (int64_t)h * p.T * 64;
        bf16_t* Vl = p.cache + (int64_t)(2 * li + 1) * p.plane + (int64_t)h * p.T * 64;
        const int64_t at = (int64_t)b * p.T * HD + (int64_t)tdev * 64 + lane;
        Kl[at] = fresh_k[lane]; Vl[at] = fresh_v[lane];
        if (p.rows && lane == 0 && h == 0) p.rows[(int64_t)b * p.T + tdev] = b;
    }
    const bf16x8 fk8 = *(const bf16x8*)(fresh_k + c * 8), fv8 = *(const bf16x8*)(fresh_v + c * 8);
    const u32x4_t qp = scaled_query(&sm.wide[g][h * 64 + c * 8]);
    float sc[NIS];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NIS; ++i) {
        const int j = 8 * i + r;
        bf16x8 kk = R[i];
        if (j == tdev) kk = fk8;
        const float a = sum8(dot8(kk, qp));
        sc[i] = (j <= tdev && sm.vf[g][j]) ? a : -INFINITY;
        mx = fmaxf(mx, sc[i]);
    }
    if (more) self_load<G, true, false>(R, sm, p, li, g + 1, h, b + 1, tdev, lane);       // the next sequence's keys: the key registers are free
    mx = wave_max_d(mx);
    const bool dead = mx == -INFINITY;               // no usable key (a finished, all-pad row): output zeros
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NIS; ++i) { sc[i] = dead ? 0.f : __expf(sc[i] - mx); sum += sc[i]; }
    sum = wave_sum_d(sum) * 0.125f;                  // every key is counted by the 8 lanes of its row
    const float inv = dead ? 0.f : 1.f / sum;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NIS; ++i) {
        const int j = 8 * i + r;
        bf16x8 vv = R[NIS + i];
        if (j == tdev) vv = fv8;
        if (j > tdev) vv = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += sc[i] * bf2f((bf16_t)vv[e]);
    }
    if (more) self_load<G, false, true>(R, sm, p, li, g + 1, h, b + 1, tdev, lane);       // ... and its values
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = sum_rows(o[e]);
    if (r == 0) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        *(u32x4*)(&sm.ctx[g][h * 64 + c * 8]) = (u32x4){pack_bf2(o[0] * inv, o[1] * inv), pack_bf2(o[2] * inv, o[3] * inv), pack_bf2(o[4] * inv, o[5] * inv), pack_bf2(o[6] * inv, o[7] * inv)};
    }
}
// the image's patch keys of head h -> R[0 .. NIC), values -> R[NIC .. 2 NIC) (shared by the G sequences)
__device__ __forceinline__ void cross_load(bf16x8 (&R)[NR], const FusedP& p, int li, int img, int h, int lane) {
    const int r = lane >> 3, c = lane & 7;
    const char* Kc = (const char*)(p.ckv + (int64_t)(2 * li) * p.plane2 + ((int64_t)img * NH + h) * p.npa * 64);
    const char* Vc = (const char*)(p.ckv + (int64_t)(2 * li + 1) * p.plane2 + ((int64_t)img * NH + h) * p.npa * 64);
#pragma unroll
    for (int i = 0; i < NIC; ++i) {
        const unsigned off = ((unsigned)min(8 * i + r, p.npa - 1) * 64u + (unsigned)c * 8u) * 2u;       // past the last key: a real row, probability 0
        R[i] = *(const bf16x8*)(Kc + off);
        R[NIC + i] = *(const bf16x8*)(Vc + off);
    }
}
template <int G>
__device__ __forceinline__ void cross_compute(bf16x8 (&R)[NR], Lds<G>& sm, const FusedP& p, int h, int lane) {
    const int r = lane >> 3, c = lane & 7;
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        // the rows are the same for every sequence: without this hipcc hoists their 320 bf16 -> fp32 conversions out of the loop (and spills them)
#pragma unroll
        for (int i = 0; i < 2 * NIC; ++i) asm volatile("" : "+v"(R[i]));
        const u32x4_t qp = scaled_query(&sm.wide[g][h * 64 + c * 8]);
        float sc[NIC];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NIC; ++i) {
            const float a = sum8(dot8(R[i], qp));
            sc[i] = (8 * i + r < p.npa) ? a : -INFINITY;
            mx = fmaxf(mx, sc[i]);
        }
        mx = wave_max_d(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NIC; ++i) { sc[i] = __expf(sc[i] - mx); sum += sc[i]; }
        const float inv = 1.f / (wave_sum_d(sum) * 0.125f);
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NIC; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += sc[i] * bf2f((bf16_t)R[NIC + i][e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sum_rows(o[e]);
        if (r == 0) {
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            *(u32x4*)(&sm.ctx[g][h * 64 + c * 8]) = (u32x4){pack_bf2(o[0] * inv, o[1] * inv), pack_bf2(o[2] * inv, o[3] * inv), pack_bf2(o[4] * inv, o[5] * inv), pack_bf2(o[6] * inv, o[7] * inv)};
        }
    }
}
// biases + LayerNorm weights of one layer -> LDS (256 threads; the loads go out together, one wait, then the writes)
__device__ __forceinline__ void params_load(float* par, const FusedLayer& L, int t256_) {
    const int t256 = opaque(t256_);
    const float* src[12] = {L.bqkv, L.bo, L.bcq, L.bco, L.bfc1, L.bfc2, L.ln1w, L.ln1b, L.ln2w, L.ln2b, L.ln3w, L.ln3b};
    const int dst[12] = {PB_QKV, PB_O, PB_CQ, PB_CO, PB_FC1, PB_FC2, PL1W, PL1B, PL2W, PL2B, PL3W, PL3B};
    const int n4[12] = {192, 64, 64, 64, 192, 64, 64, 64, 64, 64, 64, 64};
    float4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = ((const float4*)src[k])[min(t256, n4[k] - 1)];
#pragma unroll
    for (int k = 0; k < 12; ++k) if (t256 < n4[k]) *(float4*)(par + dst[k] + t256 * 4) = v[k];
}

// ---- the two roles: each runs its own layer loop; the barrier sequences match (1 + 11 per layer) -------------------------------
template <int G>
__device__ __forceinline__ void linear_role(Lds<G>& sm, const FusedP& p, int w, int lane0, int b0) {
    bf16x8 R[NR];                                     // only R[0 .. WIN) is used here: the weight window
    {
        // embeddings: x0 = LN(word[token] + type[0] + position[posid]) (modeling_roberta.py:75-122); requested BEFORE the window
        const int lane = opaque(lane0);
        float4 wv = make_float4(0, 0, 0, 0), ty = wv, pv = wv, ga = wv, be = wv;
        if (w < G) {
            const int64_t id = p.tokens[b0 + w];
            const int pid = p.posids[b0 + w];
            wv = *(const float4*)(p.word + id * HD + lane * 4); ty = *(const float4*)(p.type0 + lane * 4);
            pv = *(const float4*)(p.postab + (int64_t)pid * HD + lane * 4);
            ga = *(const float4*)(p.elnw + lane * 4); be = *(const float4*)(p.elnb + lane * 4);
        }
        fill_window<6, 8>(R, wave_frags<6, 8>(p.L[0].wqkv, w), lane);
        if (w < G) {
            const float4 v = make_float4(wv.x + ty.x + pv.x, wv.y + ty.y + pv.y, wv.z + ty.z + pv.z, wv.w + ty.w + pv.w);
            const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / HD);
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) * (1.f / HD) + p.eps);
            const float y0 = a * rstd * ga.x + be.x, y1 = b * rstd * ga.y + be.y, y2 = c * rstd * ga.z + be.z, y3 = d * rstd * ga.w + be.w;
            *(float4*)(&sm.xs[w][lane * 4]) = make_float4(y0, y1, y2, y3);
            *(uint2*)(&sm.ab[w][lane * 4]) = make_uint2(pack_bf2(y0, y1), pack_bf2(y2, y3));
        }
    }
    wg_barrier();
    DF_STAMP(1);
    for (int li = 0; li < p.nlayers; ++li) {
        const FusedLayer& L = p.L[li];
        const float* par = sm.par[li & 1];
        DF_STAMP_L(0);
        {   // q | k | v = x Wqkv^T + b, in two column halves (24 accumulator registers instead of 48: the window gets them)
            const int lane = opaque(lane0), l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 acc[6];
                if (half == 0) rows_gemm<6, 8, 6, 8, 0, G>(R, wave_frags<6, 8>(L.wqkv, w), wave_frags<6, 8>(L.wqkv, w + 24), &sm.ab[0][0], LDH, lane, acc);
                else rows_gemm<6, 8, 4, 8, 48 % WIN, G>(R, wave_frags<6, 8>(L.wqkv, w + 24), wave_frags<4, 8>(L.wo, w), &sm.ab[0][0], LDH, lane, acc);
                if (l15 < G) {
#pragma unroll
                    for (int cb = 0; cb < 6; ++cb) {
                        const int n = (w + 4 * (cb + 6 * half)) * 16 + 4 * g4;
                        const float4 bb = *(const float4*)(par + PB_QKV + n);
                        *(uint2*)(&sm.wide[l15][n]) = make_uint2(pack_bf2(acc[cb][0] + bb.x, acc[cb][1] + bb.y), pack_bf2(acc[cb][2] + bb.z, acc[cb][3] + bb.w));
                    }
                }
            }
        }
        wg_barrier();            // B1
        DF_STAMP_L(1);
        wg_barrier();            // B2: self-attention done
        DF_STAMP_L(2);
        {   // s1 = ctx Wo^T + b + x
            const int lane = opaque(lane0), l15 = lane & 15, g4 = lane >> 4;
            f32x4 acc[4];
            rows_gemm<4, 8, 4, 8, (12 * 8) % WIN, G>(R, wave_frags<4, 8>(L.wo, w), wave_frags<4, 8>(L.wcq, w), &sm.ctx[0][0], LDH, lane, acc);
            if (l15 < G) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const int n = (w + 4 * cb) * 16 + 4 * g4;
                    const float4 bb = *(const float4*)(par + PB_O + n), x4 = *(const float4*)(&sm.xs[l15][n]);
                    *(float4*)(&sm.ss[l15][n]) = make_float4(acc[cb][0] + bb.x + x4.x, acc[cb][1] + bb.y + x4.y, acc[cb][2] + bb.z + x4.z, acc[cb][3] + bb.w + x4.w);
                }
            }
        }
        wg_barrier();            // B3
        DF_STAMP_L(3);
        if (w < G) ln_row(sm.ss[w], par + PL1W, par + PL1B, p.eps, sm.xs[w], sm.ab[w], opaque(lane0));
        wg_barrier();            // B4
        DF_STAMP_L(4);
        {   // cross query
            const int lane = opaque(lane0), l15 = lane & 15, g4 = lane >> 4;
            f32x4 acc[4];
            rows_gemm<4, 8, 4, 8, (12 * 8 + 32) % WIN, G>(R, wave_frags<4, 8>(L.wcq, w), wave_frags<4, 8>(L.wco, w), &sm.ab[0][0], LDH, lane, acc);
            if (l15 < G) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const int n = (w + 4 * cb) * 16 + 4 * g4;
                    const float4 bb = *(const float4*)(par + PB_CQ + n);
                    *(uint2*)(&sm.wide[l15][n]) = make_uint2(pack_bf2(acc[cb][0] + bb.x, acc[cb][1] + bb.y), pack_bf2(acc[cb][2] + bb.z, acc[cb][3] + bb.w));
                }
            }
        }
        wg_barrier();            // B5
        DF_STAMP_L(5);
        wg_barrier();            // B6: cross-attention done
        DF_STAMP_L(6);
        {   // s2 = cctx Wco^T + b + x
            const int lane = opaque(lane0), l15 = lane & 15, g4 = lane >> 4;
            f32x4 acc[4];
            rows_gemm<4, 8, 6, 8, (12 * 8 + 64) % WIN, G>(R, wave_frags<4, 8>(L.wco, w), wave_frags<6, 8>(L.wfc1, w), &sm.ctx[0][0], LDH, lane, acc);
            if (l15 < G) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const int n = (w + 4 * cb) * 16 + 4 * g4;
                    const float4 bb = *(const float4*)(par + PB_CO + n), x4 = *(const float4*)(&sm.xs[l15][n]);
                    *(float4*)(&sm.ss[l15][n]) = make_float4(acc[cb][0] + bb.x + x4.x, acc[cb][1] + bb.y + x4.y, acc[cb][2] + bb.z + x4.z, acc[cb][3] + bb.w + x4.w);
                }
            }
        }
        wg_barrier();            // B7
        DF_STAMP_L(7);
        if (w < G) ln_row(sm.ss[w], par + PL2W, par + PL2B, p.eps, sm.xs[w], sm.ab[w], opaque(lane0));
        wg_barrier();            // B8
        DF_STAMP_L(8);
        {   // FFN: act = gelu(x Wfc1^T + b), in two column halves
            const int lane = opaque(lane0), l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 acc[6];
                if (half == 0) rows_gemm<6, 8, 6, 8, (96 + 96) % WIN, G>(R, wave_frags<6, 8>(L.wfc1, w), wave_frags<6, 8>(L.wfc1, w + 24), &sm.ab[0][0], LDH, lane, acc);
                else rows_gemm<6, 8, 4, 24, (96 + 96 + 48) % WIN, G>(R, wave_frags<6, 8>(L.wfc1, w + 24), wave_frags<4, 24>(L.wfc2, w), &sm.ab[0][0], LDH, lane, acc);
                if (l15 < G) {
#pragma unroll
                    for (int cb = 0; cb < 6; ++cb) {
                        const int n = (w + 4 * (cb + 6 * half)) * 16 + 4 * g4;
                        const float4 bb = *(const float4*)(par + PB_FC1 + n);
                        float y[4], d;
                        gelu_erf_both(acc[cb][0] + bb.x, &y[0], &d); gelu_erf_both(acc[cb][1] + bb.y, &y[1], &d);
                        gelu_erf_both(acc[cb][2] + bb.z, &y[2], &d); gelu_erf_both(acc[cb][3] + bb.w, &y[3], &d);
                        *(uint2*)(&sm.wide[l15][n]) = make_uint2(pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]));
                    }
                }
            }
        }
        wg_barrier();            // B9
        DF_STAMP_L(9);
        {   // s3 = act Wfc2^T + b + x
            const int lane = opaque(lane0), l15 = lane & 15, g4 = lane >> 4;
            f32x4 acc[4];
            if (li + 1 < p.nlayers) rows_gemm<4, 24, 6, 8, (12 * 8 + 96 + 96) % WIN, G>(R, wave_frags<4, 24>(L.wfc2, w), wave_frags<6, 8>(p.L[li + 1].wqkv, w), &sm.wide[0][0], LDW, lane, acc);
            else rows_gemm<4, 24, 4, 8, (12 * 8 + 96 + 96) % WIN, G>(R, wave_frags<4, 24>(L.wfc2, w), wave_frags<4, 8>(p.whd, w), &sm.wide[0][0], LDW, lane, acc);
            if (l15 < G) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const int n = (w + 4 * cb) * 16 + 4 * g4;
                    const float4 bb = *(const float4*)(par + PB_FC2 + n), x4 = *(const float4*)(&sm.xs[l15][n]);
                    *(float4*)(&sm.ss[l15][n]) = make_float4(acc[cb][0] + bb.x + x4.x, acc[cb][1] + bb.y + x4.y, acc[cb][2] + bb.z + x4.z, acc[cb][3] + bb.w + x4.w);
                }
            }
        }
        wg_barrier();            // B10
        DF_STAMP_L(10);
        if (w < G) ln_row(sm.ss[w], par + PL3W, par + PL3B, p.eps, sm.xs[w], sm.ab[w], opaque(lane0));
        wg_barrier();            // B11
        DF_STAMP_L(11);
    }
    DF_STAMP(40);
    {   // LM head, first half: gelu(x Wd^T + b) in fp32 (its LayerNorm and the vocabulary GEMM are the next launch)
        const int lane = opaque(lane0), l15 = lane & 15, g4 = lane >> 4;
        const float* par = sm.par[p.nlayers & 1];
        f32x4 acc[4];
        rows_gemm<4, 8, 4, 8, 0, G>(R, wave_frags<4, 8>(p.whd, w), nullptr, &sm.ab[0][0], LDH, lane, acc);
        if (l15 < G) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const int n = (w + 4 * cb) * 16 + 4 * g4;
                const float4 bb = *(const float4*)(par + n);
                float y[4], d;
                gelu_erf_both(acc[cb][0] + bb.x, &y[0], &d); gelu_erf_both(acc[cb][1] + bb.y, &y[1], &d);
                gelu_erf_both(acc[cb][2] + bb.z, &y[2], &d); gelu_erf_both(acc[cb][3] + bb.w, &y[3], &d);
                *(float4*)(p.hd_out + (int64_t)(b0 + l15) * HD + n) = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    }
    DF_STAMP(41);
}

template <int G>
__device__ __forceinline__ void attention_role(Lds<G>& sm, const FusedP& p, int h, int lane0, int t256, int img, int b0, int tdev) {
    bf16x8 R[NR];                                     // key + value rows
    {   // layer 0's biases / LayerNorm weights, the sequences' row table and usable-key flags -> LDS
        const int lane = opaque(lane0);
        params_load(sm.par[0], p.L[0], t256);
        for (int g = h; g < G; g += 4) {
            const int b = b0 + g;
#pragma unroll
            for (int q = 0; q < TMAX / 64; ++q) {
                const int j = lane + 64 * q;
                sm.rt[g][j] = p.rows ? p.rows[(int64_t)b * p.T + min(j, p.T - 1)] : b;
                sm.vf[g][j] = p.valid ? p.valid[(int64_t)b * p.ldvalid + min(j, (int)p.ldvalid - 1)] : (unsigned char)1;
            }
        }
    }
    wg_barrier();
    self_load<G, true, true>(R, sm, p, 0, 0, h, b0, tdev, opaque(lane0));       // in flight while the first QKV projection runs
    for (int li = 0; li < p.nlayers; ++li) {
        wg_barrier();            // B1: q | k | v of this step are in LDS
        DF_STAMP_A(0);
        {   // cached self-attention of head h for every sequence; then the image's patch keys / values are requested
            const int lane = opaque(lane0);
#pragma unroll 1
            for (int g = 0; g < G; ++g) {             // sequence g + 1's rows are requested while sequence g's are being used
                self_compute<G>(R, sm, p, li, g, h, b0 + g, tdev, lane, g + 1 < G);
                DF_STAMP_A(1);
            }
            DF_STAMP_A(2);
        }
        wg_barrier();            // B2: the attention output is in LDS; the image's patch keys / values are requested behind it
        cross_load(R, p, li, img, h, opaque(lane0));
        DF_STAMP_A(3);
        wg_barrier();            // B3
        wg_barrier();            // B4
        wg_barrier();            // B5: the cross query is in LDS
        DF_STAMP_A(4);
        {   // cross-attention of head h for every sequence; then the next layer's parameters and its first cached rows
            const int lane = opaque(lane0);
            cross_compute<G>(R, sm, p, h, lane);
            DF_STAMP_A(5);
        }
        wg_barrier();            // B6: the cross-attention output is in LDS; behind it, the next layer's parameters and first cached rows
        if (li + 1 < p.nlayers) {
            params_load(sm.par[(li + 1) & 1], p.L[li + 1], t256);
            DF_STAMP_A(6);
            self_load<G, true, true>(R, sm, p, li + 1, 0, h, b0, tdev, opaque(lane0));
            DF_STAMP_A(7);
        } else if (opaque(t256) < 64) {
            *(float4*)(sm.par[(li + 1) & 1] + opaque(t256) * 4) = ((const float4*)p.bhd)[opaque(t256)];      // the LM head dense layer's bias
        }
        wg_barrier();            // B7
        wg_barrier();            // B8
        wg_barrier();            // B9
        wg_barrier();            // B10
        wg_barrier();            // B11
    }
}

template <int G>
__global__ __launch_bounds__(512) void decode_fused_kernel(const FusedP p) {
    __shared__ Lds<G> sm;
    const int tid = threadIdx.x, lane0 = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.x, b0 = img * G;
    const int tdev = p.tptr ? min(*p.tptr, p.T - 1) : p.t;
    static_assert((12 * 8 + 3 * 32 + 2 * 96) % WIN == 0, "a layer's fragments must fill the window a whole number of times");
    DF_STAMP(0);
    if (w < 4) linear_role<G>(sm, p, w, lane0, b0);
    else attention_role<G>(sm, p, w - 4, lane0, tid - 256, img, b0, tdev);
}

}  // namespace

#ifdef KZV_STAMPS
extern "C" int kzv_debug_decode_stamps(long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_df_stamps), sizeof(long long) * (n < 64 ? n : 64)) == hipSuccess ? 0 : 1;
}
#endif

int kzv_decode_fused_supported(int Hd, int heads, int Fd, int layers, int group, int T, int npa) {
    return Hd == HD && heads == NH && Fd == FD && layers >= 1 && layers <= KZV_DECODE_FUSED_MAX_LAYERS && (group == 1 || group == 2 || group == 4) &&
           T >= 1 && T <= TMAX && npa >= 1 && npa <= NPMAX;
}

int kzv_pack_frag(const bf16_t* W, bf16_t* out, int N, int K, hipStream_t s) {
    if (N % 16 || K % 32) return kzv_fail(KZV_E_ARG, "pack_frag: N %% 16, K %% 32");
    hipLaunchKernelGGL(pack_frag_kernel, dim3((N * K / 8 + 255) / 256), dim3(256), 0, s, (const uint4*)W, (uint4*)out, N, K);
    return kzv_check_launch("pack_frag");
}

int kzv_decode_fused_launch(const KzvDecodeFused& a, hipStream_t s) {
    if (!kzv_decode_fused_supported(HD, NH, FD, a.nlayers, a.group, a.T, a.npa)) return kzv_fail(KZV_E_ARG, "decode_fused: geometry not instantiated");
    if (a.B % a.group) return kzv_fail(KZV_E_ARG, "decode_fused: rows must be a multiple of the group");
    if (a.rows && (int64_t)a.B * a.T * HD >= (1ll << 31)) return kzv_fail(KZV_E_ARG, "decode_fused: cache too large");
    FusedP p;
    for (int i = 0; i < a.nlayers; ++i) {
        const KzvDecodeFusedLayer& l = a.layers[i];
        p.L[i] = FusedLayer{l.wqkv, l.wo, l.wcq, l.wco, l.wfc1, l.wfc2, l.bqkv, l.bo, l.bcq, l.bco, l.bfc1, l.bfc2, l.ln1w, l.ln1b, l.ln2w, l.ln2b, l.ln3w, l.ln3b};
    }
    for (int i = a.nlayers; i < KZV_DECODE_FUSED_MAX_LAYERS; ++i) p.L[i] = p.L[0];
    p.nlayers = a.nlayers; p.tokens = a.tokens; p.posids = a.posids; p.word = a.word; p.type0 = a.type0; p.postab = a.postab; p.elnw = a.elnw; p.elnb = a.elnb;
    p.whd = a.whd; p.bhd = a.bhd; p.hd_out = a.hd_out; p.cache = a.cache; p.plane = a.plane; p.ckv = a.ckv; p.plane2 = a.plane2;
    p.valid = a.valid; p.ldvalid = a.ldvalid; p.tptr = a.tptr; p.t = a.t; p.T = a.T; p.npa = a.npa; p.B = a.B; p.rows = a.rows; p.eps = a.eps;
    const int images = a.B / a.group;
    if (a.group == 1) hipLaunchKernelGGL(decode_fused_kernel<1>, dim3(images), dim3(512), 0, s, p);
    else if (a.group == 2) hipLaunchKernelGGL(decode_fused_kernel<2>, dim3(images), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(decode_fused_kernel<4>, dim3(images), dim3(512), 0, s, p);
    return kzv_check_launch("decode_fused");
}
