// Model handle: parameter table, workspace plan, forward / backward schedules of the TrOCR training step.
//
// forward  = TrOCRModel.forward training branch (src/models/trocr_model.py:258-297):
//            ViTEncoder.forward (:169-202, 12 x HF ViTLayer pre-LN) -> encoder_decoder_proj (:269) ->
//            RobertaForCausalLM teacher-forced (HF modeling_roberta.py:75-122, 421-464, 877-893) -> CE (:292)
// backward = what loss.backward() does for that graph, hand-derived (no autograd), every matmul on the
//            MFMA GEMMs of gemm.hip, everything else fused into their epilogues or the HBM-bound kernels.
//
// Numeric policy (scripts/train_trocr.py:165-176 "bf16-mixed"): fp32 master weights and gradients, fp32
// residual stream / LayerNorm / softmax / loss, bf16 GEMM operands with fp32 accumulation.
#include "kzv_host.h"
#include "kzv_kernels.h"
#include "../../include/kzv.h"
#include "gemm_nt.h"
#include "gemm_tn.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct PEntry { std::string name; int64_t off, rows, cols; };

inline int64_t align_up(int64_t n, int64_t a) { return (n + a - 1) / a * a; }

struct EncLayerP { int64_t ln1w, ln1b, qkvw, qkvb, ow, ob, ln2w, ln2b, fc1w, fc1b, fc2w, fc2b; };
struct DecLayerP { int64_t qkvw, qkvb, ow, ob, ln1w, ln1b, cqw, cqb, cow, cob, ln2w, ln2b, fc1w, fc1b, fc2w, fc2b, ln3w, ln3b; };
struct W16 { bf16_t* w; bf16_t* wt; int64_t ldt; };   // bf16 copy [N,K] and transposed copy [K, ldt]
struct W8 { unsigned char* w; float* scale; };         // fp8 path: e4m3 copy [N,K] quantised per output row, scale [N]

struct EncAct {
    float *x_in, *x_mid, *st1, *st2, *lse;
    bf16_t *ln1, *qkv, *ctx, *ln2, *pre, *act;
};
struct DecAct {
    float *s1, *x1, *s2, *x2, *s3, *x3, *st1, *st2, *st3, *lse_sa, *lse_ca;
    bf16_t *qkv, *ctx, *x1h, *cq, *cctx, *x2h, *pre, *act, *x3h;
    // backward (segment path): this layer's six "dY" operands stay alive until ONE grouped weight-gradient launch behind the last layer
    bf16_t *g_dy = nullptr, *g_dbig = nullptr, *g_dy2 = nullptr, *g_dq = nullptr, *g_dy3 = nullptr, *g_dqkv = nullptr;
};

}  // namespace

struct kzv_model {
    kzv_config c;
    int np, Se, PD, He, Fe, Hd, Fd, V, Vp, Le, Ld;
    int npa = 0, Sa = 0, img_w = 0;   // ACTIVE geometry (kzv_set_image_width): img_w <= c.image_w, npa patches, Sa = npa + 1 tokens
    bool has_proj;
    std::vector<PEntry> table;
    int64_t total = 0;
    // parameter offsets
    int64_t patch_w, patch_b, cls, pos, lnf_w, lnf_b, proj_w, proj_b, word, dpos, dtype, eln_w, eln_b, ckv_w, ckv_b,
        hd_w, hd_b, hln_w, hln_b, hbias;
    std::vector<EncLayerP> ep;
    std::vector<DecLayerP> dp;
    // bound state
    float* P = nullptr; float* G = nullptr; char* ws = nullptr; int64_t ws_bytes = 0;
    int B = 0, L = 0, T = 0, Ta = 0;
    int Be = 0;              // images the encoder states / cross-attention K/V currently hold (B after kzv_forward_loss; fewer after kzv_encode_images)
    bool bound = false, have_fwd = false, have_enc = false, train = false;
    uint64_t seed = 0;
    const int64_t* labels = nullptr;
    // workspace pointers
    KzvCastDesc* d_desc = nullptr; int ndesc = 0, cast_tiles = 0;
    std::vector<KzvCastDesc> h_desc;
    W16 w_patch, w_proj, w_word, w_ckv, w_hd;
    std::vector<W16> w_eqkv, w_eo, w_efc1, w_efc2, w_dqkv, w_do, w_dcq, w_dco, w_dfc1, w_dfc2;
    bf16_t *patches, *enc_out, *proj_out, *crosskv, *xd0h, *hd_pre, *hd_ln, *dlogits;
    float *pe32, *x_last, *stf, *emb_sum, *emb_st, *xd0, *hd_gelu, *hd_st, *logits, *count, *loss_acc;
    int *posids, *err;
    std::vector<EncAct> ea;
    std::vector<DecAct> da;
    // backward scratch
    float *dx_e, *dx_d, *dsum_d;
    bf16_t *dy_e2;           // second dy_e (overlap mode 2: the fc2 weight gradient still reads dy_e while LayerNorm-2 backward writes its output)
    bf16_t *dy_e, *dbig_e, *dh_e, *dqkv_e, *dctx_e, *dpatch, *denc_out, *denc, *dckv, *dy_d, *dbig_d, *dqkv_d, *dctx_d, *dq_d, *dhln;
    bf16_t *dy_d2, *dy_d3;   // the decoder's three "dropout(linear)" gradients of a layer stay alive until its grouped weight-gradient launch
    std::vector<kzv_gemm_tn_args> wbatch;
    // weight-gradient GEMMs run on an internal side stream so they overlap the input-gradient chain on the
    // caller's stream (their tails and epilogues fill each other's idle workgroup slots)
    hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr; hipEvent_t ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
    bool pending[4] = {false, false, false, false}; bool use_side = false, join_each_segment = true;
    // KV cache of the generation path (kzv_decode_step): two copies [2*Ld][B][T][Hd] (beam re-ordering gathers from one into the other)
    bf16_t* kvc[2] = {nullptr, nullptr}; int kv_cur = 0, kvB = 0, kvT = 0;
    // beam re-parenting by indirection: rowtab[x][b][j] = cache row holding key j of sequence b; rt_cur = -1: identity (no table)
    int* rowtab[2] = {nullptr, nullptr}; int rt_cur = -1;
    // cross-attention K/V re-laid out for the generation steps ([layer][K|V][image][head][key][64]); rebuilt when the encoder ran
    bf16_t* ckv_dec = nullptr; size_t ckv_dec_bytes = 0; bool ckv_dec_ok = false;
    // the decoder's bf16 weights in MFMA fragment order (decode_fused.hip), refreshed after every weight change
    bf16_t* dec_pack = nullptr; bool dec_pack_ok = false; int64_t head_pack_off = 0, tpack_off = 0, head_tpack_off = 0;
    bool dhln_fused = false;     // the last training forward's head_ce launch already wrote dhln (the LM head's input gradient)
    // graph-replayed decode step (kzv_decode_step_graph): device-side step index + one instantiated graph per cache copy
    int* d_t = nullptr;
    hipGraphExec_t dgraph[3] = {nullptr, nullptr, nullptr};          // one per row table in use: none, rowtab[0], rowtab[1]
    const void* dg_key[3][6] = {};
    int64_t dg_ld[3] = {0, 0, 0};
    // fp8 weight path (kzv_set_fp8; BASELINE configs[4]): the encoder's QKV, fc1 and fc2 FORWARD GEMMs read e4m3 operands.
    // Weights: one e4m3 copy per matrix, quantised per output row from the fp32 master at kzv_model_sync_weights.  Activations:
    // LayerNorm writes an e4m3 copy of its output beside the bf16 one, quantised per token row (x8, x8_scale); the fc1 GELU
    // epilogue writes an e4m3 copy of the activation with a per-tensor multiplier (f8_q[layer]) derived from the largest |value|
    // the previous forward saw (f8_amax[layer]; "delayed scaling").  Backward and the output projection stay bf16.
    int fp8 = 0;
    std::vector<W8> w8_qkv, w8_fc1, w8_fc2;
    KzvQuantDesc* d_qdesc = nullptr; int nqdesc = 0, qrows = 0;
    std::vector<KzvQuantDesc> h_qdesc;
    unsigned char *x8 = nullptr, *act8 = nullptr;
    float *x8_scale = nullptr, *f8_q = nullptr, *f8_amax = nullptr, *f8_rows = nullptr;
    int64_t f8_stride = 0;
    // mode 2: also the MLP's two INPUT-GRADIENT GEMMs (d fc2 with the DGELU epilogue, d fc1).  Transposed e4m3 weight copies
    // (quantised per row from the bf16 transposed copies); the masked gradient rows arriving at fc2 come from the LayerNorm
    // backward above them as e4m3 with their own amax (dy8, dy8_scale); the gradient of the GELU input is quantised per row
    // in the DGELU epilogue with a multiplier from the bound ||dy row|| * max ||W2 column|| (rq / rqinv; f8_wnorm[layer]).
    std::vector<W8> w8t_fc1, w8t_fc2;
    unsigned char *dy8 = nullptr, *dbig8 = nullptr;
    float *dy8_scale = nullptr, *dy8_rq = nullptr, *dy8_rqinv = nullptr, *f8_wnorm = nullptr;
    bool side_ok = false;    // mode 2: set only inside the encoder-layer schedule (everything else stays on the caller's stream)
    int side_mode = 0;       // 0 off, 1 free-running wgrads, 2 wgrads only under the HBM-bound kernels (LayerNorm / attention backward)
};

namespace {

int64_t add_param(kzv_model* m, const std::string& name, int64_t rows, int64_t cols) {
    const int64_t off = m->total;
    m->table.push_back({name, off, rows, cols});
    m->total += align_up(rows * cols, 64);
    return off;
}

// MUST match kzv/params.py::param_table (tests/test_capi_cpu.py checks name/offset/shape equality)
void build_param_table(kzv_model* m) {
    const int He = m->He, Fe = m->Fe, Hd = m->Hd, Fd = m->Fd;
    m->patch_w = add_param(m, "enc.patch.w", He, m->PD);
    m->patch_b = add_param(m, "enc.patch.b", He, 1);
    m->cls = add_param(m, "enc.cls", He, 1);
    m->pos = add_param(m, "enc.pos", m->Se, He);
    m->ep.resize(m->Le);
    for (int i = 0; i < m->Le; ++i) {
        const std::string p = "enc." + std::to_string(i) + ".";
        EncLayerP& e = m->ep[i];
        e.ln1w = add_param(m, p + "ln1.w", He, 1); e.ln1b = add_param(m, p + "ln1.b", He, 1);
        e.qkvw = add_param(m, p + "qkv.w", 3 * He, He); e.qkvb = add_param(m, p + "qkv.b", 3 * He, 1);
        e.ow = add_param(m, p + "o.w", He, He); e.ob = add_param(m, p + "o.b", He, 1);
        e.ln2w = add_param(m, p + "ln2.w", He, 1); e.ln2b = add_param(m, p + "ln2.b", He, 1);
        e.fc1w = add_param(m, p + "fc1.w", Fe, He); e.fc1b = add_param(m, p + "fc1.b", Fe, 1);
        e.fc2w = add_param(m, p + "fc2.w", He, Fe); e.fc2b = add_param(m, p + "fc2.b", He, 1);
    }
    m->lnf_w = add_param(m, "enc.lnf.w", He, 1); m->lnf_b = add_param(m, "enc.lnf.b", He, 1);
    if (m->has_proj) { m->proj_w = add_param(m, "proj.w", Hd, He); m->proj_b = add_param(m, "proj.b", Hd, 1); }
    m->word = add_param(m, "dec.word", m->V, Hd);
    m->dpos = add_param(m, "dec.pos", m->c.max_pos, Hd);
    m->dtype = add_param(m, "dec.type", m->c.type_vocab, Hd);
    m->eln_w = add_param(m, "dec.emb_ln.w", Hd, 1); m->eln_b = add_param(m, "dec.emb_ln.b", Hd, 1);
    m->ckv_w = add_param(m, "dec.cross_kv.w", (int64_t)m->Ld * 2 * Hd, Hd);
    m->ckv_b = add_param(m, "dec.cross_kv.b", (int64_t)m->Ld * 2 * Hd, 1);
    m->dp.resize(m->Ld);
    for (int i = 0; i < m->Ld; ++i) {
        const std::string p = "dec." + std::to_string(i) + ".";
        DecLayerP& d = m->dp[i];
        d.qkvw = add_param(m, p + "sa_qkv.w", 3 * Hd, Hd); d.qkvb = add_param(m, p + "sa_qkv.b", 3 * Hd, 1);
        d.ow = add_param(m, p + "sa_o.w", Hd, Hd); d.ob = add_param(m, p + "sa_o.b", Hd, 1);
        d.ln1w = add_param(m, p + "sa_ln.w", Hd, 1); d.ln1b = add_param(m, p + "sa_ln.b", Hd, 1);
        d.cqw = add_param(m, p + "ca_q.w", Hd, Hd); d.cqb = add_param(m, p + "ca_q.b", Hd, 1);
        d.cow = add_param(m, p + "ca_o.w", Hd, Hd); d.cob = add_param(m, p + "ca_o.b", Hd, 1);
        d.ln2w = add_param(m, p + "ca_ln.w", Hd, 1); d.ln2b = add_param(m, p + "ca_ln.b", Hd, 1);
        d.fc1w = add_param(m, p + "fc1.w", Fd, Hd); d.fc1b = add_param(m, p + "fc1.b", Fd, 1);
        d.fc2w = add_param(m, p + "fc2.w", Hd, Fd); d.fc2b = add_param(m, p + "fc2.b", Hd, 1);
        d.ln3w = add_param(m, p + "out_ln.w", Hd, 1); d.ln3b = add_param(m, p + "out_ln.b", Hd, 1);
    }
    m->hd_w = add_param(m, "head.dense.w", Hd, Hd); m->hd_b = add_param(m, "head.dense.b", Hd, 1);
    m->hln_w = add_param(m, "head.ln.w", Hd, 1); m->hln_b = add_param(m, "head.ln.b", Hd, 1);
    m->hbias = add_param(m, "head.bias", m->V, 1);
}

// ---- workspace bump allocator: pass 1 (base == nullptr) only measures ------------------------------
struct Bump {
    char* base; int64_t off = 0;
    template <class T> T* take(int64_t n) {
        off = align_up(off, 256);
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * (int64_t)sizeof(T);
        return p;
    }
};

W16 take_w(kzv_model* m, Bump& b, int64_t woff, int64_t N, int64_t K, bool need_t) {
    W16 w;
    w.w = b.take<bf16_t>(N * K);
    w.ldt = align_up(N, 64);
    w.wt = need_t ? b.take<bf16_t>(K * w.ldt) : nullptr;
    KzvCastDesc d;
    d.src = m->P ? m->P + woff : nullptr; d.dst = w.w; d.dstT = w.wt; d.rows = (int)N; d.cols = (int)K; d.ldT = w.ldt;
    d.tiles_c = (int)((K + 63) / 64);
    d.tile0 = m->cast_tiles;
    m->cast_tiles += (int)((N + 63) / 64) * d.tiles_c;
    m->h_desc.push_back(d);
    return w;
}

W8 take_w8t(kzv_model* m, Bump& b, const bf16_t* wt, int64_t ldt, int64_t rows, int64_t cols, float* normmax) {
    W8 w;
    w.w = b.take<unsigned char>(rows * cols);
    w.scale = b.take<float>(rows);
    KzvQuantDesc d{nullptr, w.w, w.scale, (int)rows, (int)cols, m->qrows, wt, ldt, normmax};
    m->h_qdesc.push_back(d);
    m->qrows += (int)rows;
    return w;
}

W8 take_w8(kzv_model* m, Bump& b, int64_t woff, int64_t N, int64_t K) {
    W8 w;
    w.w = b.take<unsigned char>(N * K);
    w.scale = b.take<float>(N);
    m->h_qdesc.push_back(KzvQuantDesc{m->P ? m->P + woff : nullptr, w.w, w.scale, (int)N, (int)K, m->qrows});
    m->qrows += (int)N;
    return w;
}

int64_t plan(kzv_model* m, char* base, int B, int L) {
    Bump b{base};
    const int T = L - 1;
    const int64_t Me = (int64_t)B * m->Se, Mp = (int64_t)B * m->np, Md = (int64_t)B * T;
    const int He = m->He, Fe = m->Fe, Hd = m->Hd, Fd = m->Fd;
    m->h_desc.clear(); m->cast_tiles = 0;
    // bf16 weight copies first: this whole region is zeroed at bind (transposed-copy padding stays zero)
    m->w_patch = take_w(m, b, m->patch_w, He, m->PD, false);
    auto vec = [&](std::vector<W16>& v, int n) { v.resize(n); };
    vec(m->w_eqkv, m->Le); vec(m->w_eo, m->Le); vec(m->w_efc1, m->Le); vec(m->w_efc2, m->Le);
    for (int i = 0; i < m->Le; ++i) {
        m->w_eqkv[i] = take_w(m, b, m->ep[i].qkvw, 3 * He, He, true);
        m->w_eo[i] = take_w(m, b, m->ep[i].ow, He, He, true);
        m->w_efc1[i] = take_w(m, b, m->ep[i].fc1w, Fe, He, true);
        m->w_efc2[i] = take_w(m, b, m->ep[i].fc2w, He, Fe, true);
    }
    if (m->has_proj) m->w_proj = take_w(m, b, m->proj_w, Hd, He, true);
    m->w_word = take_w(m, b, m->word, m->V, Hd, true);
    m->w_ckv = take_w(m, b, m->ckv_w, (int64_t)m->Ld * 2 * Hd, Hd, true);
    vec(m->w_dqkv, m->Ld); vec(m->w_do, m->Ld); vec(m->w_dcq, m->Ld); vec(m->w_dco, m->Ld); vec(m->w_dfc1, m->Ld); vec(m->w_dfc2, m->Ld);
    for (int i = 0; i < m->Ld; ++i) {
        m->w_dqkv[i] = take_w(m, b, m->dp[i].qkvw, 3 * Hd, Hd, true);
        m->w_do[i] = take_w(m, b, m->dp[i].ow, Hd, Hd, true);
        m->w_dcq[i] = take_w(m, b, m->dp[i].cqw, Hd, Hd, true);
        m->w_dco[i] = take_w(m, b, m->dp[i].cow, Hd, Hd, true);
        m->w_dfc1[i] = take_w(m, b, m->dp[i].fc1w, Fd, Hd, true);
        m->w_dfc2[i] = take_w(m, b, m->dp[i].fc2w, Hd, Fd, true);
    }
    m->w_hd = take_w(m, b, m->hd_w, Hd, Hd, true);
    m->ndesc = (int)m->h_desc.size();
    m->d_desc = b.take<KzvCastDesc>(m->ndesc);
    const int64_t weights_end = b.off;
    m->h_qdesc.clear(); m->qrows = 0;
    if (m->fp8) {
        m->w8_qkv.resize(m->Le); m->w8_fc1.resize(m->Le); m->w8_fc2.resize(m->Le);
        for (int i = 0; i < m->Le; ++i) {
            m->w8_qkv[i] = take_w8(m, b, m->ep[i].qkvw, 3 * He, He);
            m->w8_fc1[i] = take_w8(m, b, m->ep[i].fc1w, Fe, He);
            m->w8_fc2[i] = take_w8(m, b, m->ep[i].fc2w, He, Fe);
        }
        if (m->fp8 >= 2) {
            m->f8_wnorm = b.take<float>(m->Le);
            m->w8t_fc1.resize(m->Le); m->w8t_fc2.resize(m->Le);
            for (int i = 0; i < m->Le; ++i) {
                m->w8t_fc2[i] = take_w8t(m, b, m->w_efc2[i].wt, m->w_efc2[i].ldt, Fe, He, m->f8_wnorm ? m->f8_wnorm + i : nullptr);   // rows = W2 columns
                m->w8t_fc1[i] = take_w8t(m, b, m->w_efc1[i].wt, m->w_efc1[i].ldt, He, Fe, nullptr);
            }
            m->dy8 = b.take<unsigned char>(Me * He); m->dbig8 = b.take<unsigned char>(Me * Fe);
            m->dy8_scale = b.take<float>(Me); m->dy8_rq = b.take<float>(Me); m->dy8_rqinv = b.take<float>(Me);
        }
        m->nqdesc = (int)m->h_qdesc.size();
        m->d_qdesc = b.take<KzvQuantDesc>(m->nqdesc);
        m->x8 = b.take<unsigned char>(Me * He); m->act8 = b.take<unsigned char>(Me * Fe);
        m->x8_scale = b.take<float>(Me);
        m->f8_q = b.take<float>(m->Le); m->f8_amax = b.take<float>(m->Le);
        m->f8_stride = Me; m->f8_rows = b.take<float>(Me * m->Le);
    }

    // scalars
    m->count = b.take<float>(64); m->loss_acc = m->count + 1; m->err = (int*)(m->count + 2);
    m->d_t = (int*)b.take<float>(64);
    m->posids = b.take<int>(Md);
    // encoder activations
    m->patches = b.take<bf16_t>(Mp * m->PD);
    m->pe32 = b.take<float>(Mp * He);
    m->ea.resize(m->Le);
    const int64_t lse_e = (int64_t)B * m->c.enc_heads * m->Se;
    for (int i = 0; i < m->Le; ++i) {
        EncAct& a = m->ea[i];
        a.x_in = b.take<float>(Me * He); a.x_mid = b.take<float>(Me * He);
        a.st1 = b.take<float>(Me * 2); a.st2 = b.take<float>(Me * 2); a.lse = b.take<float>(lse_e);
        a.ln1 = b.take<bf16_t>(Me * He); a.qkv = b.take<bf16_t>(Me * 3 * He); a.ctx = b.take<bf16_t>(Me * He);
        a.ln2 = b.take<bf16_t>(Me * He); a.pre = b.take<bf16_t>(Me * Fe); a.act = b.take<bf16_t>(Me * Fe);
    }
    m->x_last = b.take<float>(Me * He); m->stf = b.take<float>(Me * 2);
    m->enc_out = b.take<bf16_t>(Mp * He);
    m->proj_out = m->has_proj ? b.take<bf16_t>(Mp * Hd) : m->enc_out;
    // decoder activations
    const int64_t CK = (int64_t)m->Ld * 2 * Hd;
    m->crosskv = b.take<bf16_t>(Mp * CK);
    m->emb_sum = b.take<float>(Md * Hd); m->emb_st = b.take<float>(Md * 2);
    m->xd0 = b.take<float>(Md * Hd); m->xd0h = b.take<bf16_t>(Md * Hd);
    m->da.resize(m->Ld);
    const int64_t lse_d = (int64_t)B * m->c.dec_heads * T;
    for (int i = 0; i < m->Ld; ++i) {
        DecAct& a = m->da[i];
        a.s1 = b.take<float>(Md * Hd); a.x1 = b.take<float>(Md * Hd); a.s2 = b.take<float>(Md * Hd); a.x2 = b.take<float>(Md * Hd);
        a.s3 = b.take<float>(Md * Hd); a.x3 = b.take<float>(Md * Hd);
        a.st1 = b.take<float>(Md * 2); a.st2 = b.take<float>(Md * 2); a.st3 = b.take<float>(Md * 2);
        a.lse_sa = b.take<float>(lse_d); a.lse_ca = b.take<float>(lse_d);
        a.qkv = b.take<bf16_t>(Md * 3 * Hd); a.ctx = b.take<bf16_t>(Md * Hd); a.x1h = b.take<bf16_t>(Md * Hd);
        a.cq = b.take<bf16_t>(Md * Hd); a.cctx = b.take<bf16_t>(Md * Hd); a.x2h = b.take<bf16_t>(Md * Hd);
        a.pre = b.take<bf16_t>(Md * Fd); a.act = b.take<bf16_t>(Md * Fd); a.x3h = b.take<bf16_t>(Md * Hd);
    }
    m->hd_pre = b.take<bf16_t>(Md * Hd); m->hd_gelu = b.take<float>(Md * Hd); m->hd_st = b.take<float>(Md * 2);
    m->hd_ln = b.take<bf16_t>(Md * Hd);
    m->logits = b.take<float>(Md * m->Vp); m->dlogits = b.take<bf16_t>(Md * m->Vp);
    // backward scratch
    m->dx_e = b.take<float>(Me * He); m->dy_e = b.take<bf16_t>(Me * He); m->dy_e2 = b.take<bf16_t>(Me * He); m->dbig_e = b.take<bf16_t>(Me * Fe);
    m->dh_e = b.take<bf16_t>(Me * He); m->dqkv_e = b.take<bf16_t>(Me * 3 * He); m->dctx_e = b.take<bf16_t>(Me * He);
    m->dpatch = b.take<bf16_t>(Mp * He); m->denc_out = b.take<bf16_t>(Mp * He);
    m->denc = m->has_proj ? b.take<bf16_t>(Mp * Hd) : m->denc_out;
    m->dckv = b.take<bf16_t>(Mp * CK);
    m->dx_d = b.take<float>(Md * Hd); m->dsum_d = b.take<float>(Md * Hd);
    m->dy_d2 = b.take<bf16_t>(Md * Hd); m->dy_d3 = b.take<bf16_t>(Md * Hd);
    m->dy_d = b.take<bf16_t>(Md * Hd); m->dbig_d = b.take<bf16_t>(Md * Fd); m->dqkv_d = b.take<bf16_t>(Md * 3 * Hd);
    m->dctx_d = b.take<bf16_t>(Md * Hd); m->dq_d = b.take<bf16_t>(Md * Hd); m->dhln = b.take<bf16_t>(Md * Hd);
    if (kzv_dec_chain_supported(Hd, Fd))        // the segment path's per-layer gradient operands (2.8 KB per decoder row and layer)
        for (int i = 0; i < m->Ld; ++i) {
            DecAct& a = m->da[i];
            a.g_dy = b.take<bf16_t>(Md * Hd); a.g_dbig = b.take<bf16_t>(Md * Fd); a.g_dy2 = b.take<bf16_t>(Md * Hd);
            a.g_dq = b.take<bf16_t>(Md * Hd); a.g_dy3 = b.take<bf16_t>(Md * Hd); a.g_dqkv = b.take<bf16_t>(Md * 3 * Hd);
        }
    (void)weights_end;
    return align_up(b.off, 256);
}

// dropout site ids (distinct hash keys per call site and layer)
enum { SITE_ENC_EMB = KZV_SITE_ENC_EMB, SITE_ENC_L = KZV_SITE_ENC_LAYER(0, 0), SITE_DEC_EMB = KZV_SITE_DEC_EMB, SITE_DEC_L = KZV_SITE_DEC_LAYER(0, 0) };
inline uint32_t key(const kzv_model* m, uint32_t site) { return kzv_drop_key(m->seed, site); }
inline float dp(const kzv_model* m, float p) { return m->train ? p : 0.f; }

// decoder_chain.hip: the linear chains of a decoder layer as two launches (KZV_DEC_CHAIN, kzv_set_dec_chain): 0 = off, 1 = the forward
// chains (+ the 256 x 256 input gradients on the row-panel kernel), 2 (default) = also the backward's row-local segments, one launch each
int g_dec_chain = -1;
int dec_chain_mode() {
    if (g_dec_chain < 0) { const char* e = getenv("KZV_DEC_CHAIN"); g_dec_chain = e ? atoi(e) : 2; if (g_dec_chain < 0 || g_dec_chain > 2) g_dec_chain = 2; }
    return g_dec_chain;
}
bool dec_pack_wanted(const kzv_model* m);
int ensure_dec_pack(kzv_model* m, hipStream_t s);
int g_head_ce = -1;
bool head_ce_mode() {       // the one-launch LM head + CE (KZV_HEAD_CE / kzv_set_head_ce; default on)
    if (g_head_ce < 0) { const char* e = getenv("KZV_HEAD_CE"); g_head_ce = e ? (atoi(e) != 0) : 1; }
    return g_head_ce != 0;
}

#define KZV_TRY(expr) do { int rc__ = (expr); if (rc__ != KZV_OK) return rc__; } while (0)

int gemm(const bf16_t* A, int64_t lda, const W16& w, bool transposed, int M, int N, int K, int n_valid, const float* bias,
         void* C, int64_t ldc, int epi, hipStream_t s, const float* resid = nullptr, void* aux = nullptr, int64_t ldaux = 0,
         float drop_p = 0.f, uint32_t drop_key = 0) {
    kzv_gemm_nt_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda;
    a.B = transposed ? w.wt : w.w; a.ldb = transposed ? w.ldt : K;
    a.C = C; a.ldc = ldc; a.bias = bias; a.resid = resid; a.ldr = ldc; a.aux = aux; a.ldaux = ldaux;
    a.M = M; a.N = N; a.K = K; a.n_valid = n_valid; a.drop_p = drop_p; a.drop_key = drop_key;
    return kzv_gemm_nt(&a, epi, s);
}

// fp8 forward GEMM of the encoder: A e4m3 with one scale per row, W e4m3 with one scale per output row
int gemm8(const unsigned char* A, int64_t lda, const float* a_scale, const W8& w, int M, int N, int K, const float* bias, void* C, int64_t ldc,
          int epi, hipStream_t s, const float* resid = nullptr, void* aux = nullptr, int64_t ldaux = 0, float drop_p = 0.f, uint32_t drop_key = 0,
          unsigned char* c8 = nullptr, const float* c8_qscale = nullptr, float* c8_amax = nullptr, const float* c8_rowq = nullptr) {
    kzv_gemm_nt_fp8_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.a_scale = a_scale; a.B = w.w; a.ldb = K; a.b_scale = w.scale;
    a.C = C; a.ldc = ldc; a.bias = bias; a.resid = resid; a.ldr = ldc; a.aux = aux; a.ldaux = ldaux;
    a.c8 = c8; a.ldc8 = N; a.c8_qscale = c8_qscale; a.c8_amax = c8_amax; a.c8_rowq = c8_rowq;
    a.M = M; a.N = N; a.K = K; a.n_valid = N; a.drop_p = drop_p; a.drop_key = drop_key;
    return kzv_gemm_nt_fp8(&a, epi, s);
}

int wgrad(const bf16_t* dY, int64_t ldp, const bf16_t* X, int64_t ldq, float* dW, int Mtok, int N, int K, int n_store, hipStream_t s,
          float* dbias = nullptr) {
    kzv_gemm_tn_args a;
    memset(&a, 0, sizeof(a));
    a.P = dY; a.ldp = ldp; a.Q = X; a.ldq = ldq; a.OUT = dW; a.ldo = K; a.Mtok = Mtok; a.N = N; a.K = K; a.n_store = n_store; a.dbias = dbias;
    return kzv_gemm_tn(&a, s);
}

// Small weight gradients of one backward stage are collected and launched as ONE grid (kzv_gemm_tn_group): each alone
// fills a fraction of the chip (4..12 tiles of 128x128).  Only without the side stream (which has its own overlap).
int wgrad_batch(kzv_model* m, int cls, hipStream_t s, const bf16_t* dY, int64_t ldp, const bf16_t* X, int64_t ldq, float* dW, int Mtok,
                int N, int K, int n_store, float* dbias);
int wgrad_flush(kzv_model* m, hipStream_t s) {
    if (m->wbatch.empty()) return KZV_OK;
    int rc = KZV_OK;
    for (size_t i = 0; i < m->wbatch.size() && rc == KZV_OK; i += 36)        // <= 36 problems per grid (gemm.hip TN_GROUP_MAX = 40)
        rc = kzv_gemm_tn_group(m->wbatch.data() + i, (int)std::min<size_t>(36, m->wbatch.size() - i), s);
    m->wbatch.clear();
    return rc;
}

// buffer classes whose last side-stream reader must finish before the main stream overwrites them
enum { CLS_DY = 0, CLS_DBIG = 1, CLS_DQKV = 2, CLS_MISC = 3 };

int wgrad_async(kzv_model* m, int cls, hipStream_t s, const bf16_t* dY, int64_t ldp, const bf16_t* X, int64_t ldq, float* dW, int Mtok,
                int N, int K, int n_store, float* dbias) {
    if (!m->use_side || (m->side_mode == 2 && !m->side_ok)) return wgrad(dY, ldp, X, ldq, dW, Mtok, N, K, n_store, s, dbias);
    if (hipEventRecord(m->ev_fork, s) != hipSuccess || hipStreamWaitEvent(m->side, m->ev_fork, 0) != hipSuccess)
        return kzv_fail(KZV_E_HIP, "wgrad_async: fork");
    const int rc = wgrad(dY, ldp, X, ldq, dW, Mtok, N, K, n_store, m->side, dbias);
    if (rc != KZV_OK) return rc;
    if (hipEventRecord(m->ev_done[cls], m->side) != hipSuccess) return kzv_fail(KZV_E_HIP, "wgrad_async: record");
    m->pending[cls] = true;
    return KZV_OK;
}
int wgrad_batch(kzv_model* m, int cls, hipStream_t s, const bf16_t* dY, int64_t ldp, const bf16_t* X, int64_t ldq, float* dW, int Mtok,
                int N, int K, int n_store, float* dbias) {
    if (m->use_side) return wgrad_async(m, cls, s, dY, ldp, X, ldq, dW, Mtok, N, K, n_store, dbias);
    kzv_gemm_tn_args a;
    memset(&a, 0, sizeof(a));
    a.P = dY; a.ldp = ldp; a.Q = X; a.ldq = ldq; a.OUT = dW; a.ldo = K; a.Mtok = Mtok; a.N = N; a.K = K; a.n_store = n_store; a.dbias = dbias;
    m->wbatch.push_back(a);
    return KZV_OK;
}
int wait_cls(kzv_model* m, int cls, hipStream_t s) {
    if (m->pending[cls]) {
        if (hipStreamWaitEvent(s, m->ev_done[cls], 0) != hipSuccess) return kzv_fail(KZV_E_HIP, "wait_cls");
        m->pending[cls] = false;
    }
    return KZV_OK;
}
int join_side(kzv_model* m, hipStream_t s) {
    for (int c = 0; c < 4; ++c) KZV_TRY(wait_cls(m, c, s));
    return KZV_OK;
}

// A Linear's input gradient (gemm_nt against the transposed weight, one-store epilogue) and weight gradient (gemm_tn) from the same dY:
// ONE launch when gemm_tn256.hip's pair kernel takes the shapes (kzv_gemm_pair_launch), else the two launches in the order the
// single-stream schedule has always issued them (weight gradient first).
int dgrad_wgrad(kzv_model* m, int cls, hipStream_t s, const bf16_t* dY, int64_t ldy, const W16& w, int Mtok, int Nout, int Kin, void* dX, int64_t ldx,
                int epi, void* aux, int64_t ldaux, const bf16_t* X, int64_t ldxq, float* dW, float* dbias) {
    // dX[Mtok, Kin] = dY[Mtok, Nout] . W[Nout, Kin]  (B operand = W^T copy [Kin, Nout]);  dW[Nout, Kin] += dY^T . X
    if (!m->use_side && !m->fp8) {
        kzv_gemm_nt_args na;
        memset(&na, 0, sizeof(na));
        na.A = dY; na.lda = ldy; na.B = w.wt; na.ldb = w.ldt; na.C = dX; na.ldc = ldx; na.ldr = ldx; na.aux = aux; na.ldaux = ldaux;
        na.M = Mtok; na.N = Kin; na.K = Nout; na.n_valid = Kin;
        kzv_gemm_tn_args ta;
        memset(&ta, 0, sizeof(ta));
        ta.P = dY; ta.ldp = ldy; ta.Q = X; ta.ldq = ldxq; ta.OUT = dW; ta.ldo = Kin; ta.Mtok = Mtok; ta.N = Nout; ta.K = Kin; ta.n_store = Nout; ta.dbias = dbias;
        const int rc = kzv_gemm_pair_launch(&na, epi, &ta, s);
        if (rc < 0) return kzv_fail(KZV_E_HIP, "dgrad_wgrad: pair launch");
        if (rc == 1) return KZV_OK;
    }
    KZV_TRY(wgrad_async(m, cls, s, dY, ldy, X, ldxq, dW, Mtok, Nout, Kin, Nout, dbias));
    return gemm(dY, ldy, w, true, Mtok, Kin, Nout, Kin, nullptr, dX, ldx, epi, s, nullptr, aux, ldaux);
}

int attn(const kzv_model* m, bool bwd, int mode, const bf16_t* Q, int64_t ldq, const bf16_t* K, const bf16_t* V, int64_t ldkv,
         bf16_t* O, int64_t ldo, float* LSE, const bf16_t* dO, bf16_t* dQ, bf16_t* dK, bf16_t* dV, int heads, int Sq, int Sk,
         float drop_p, uint32_t drop_key, hipStream_t s, int batch = 0, int head_dim = 64) {
    kzv_attn_args a;
    memset(&a, 0, sizeof(a));
    a.Q = Q; a.K = K; a.V = V; a.O = O; a.LSE = LSE; a.dO = dO; a.dQ = dQ; a.dK = dK; a.dV = dV;
    a.ldq = ldq; a.ldk = ldkv; a.ldv = ldkv; a.ldo = ldo;
    a.ids = m->labels; a.ld_ids = m->L; a.pad_id = m->c.pad_id;
    a.B = batch > 0 ? batch : m->B; a.heads = heads; a.Sq = Sq; a.Sk = Sk; a.head_dim = head_dim; a.mode = mode; a.drop_p = drop_p; a.drop_key = drop_key;
    return bwd ? kzv_attn_bwd(&a, s) : kzv_attn_fwd(&a, s);
}

// ================================================================================================ forward
int forward(kzv_model* m, const float* px, const int64_t* labels, float* d_loss, float* d_logits, hipStream_t s,
            bool run_encoder = true, int logits_pos = -1, int enc_batch = 0, bool run_decoder = true) {
    const kzv_config& c = m->c;
    // T = ACTIVE decoder length (kzv_set_active_length): positions >= T hold only padding in every sample, are
    // masked as keys and carry no loss, so the decoder runs on the packed [B, T] prefix (rows b*T + t).
    const int T = m->Ta, He = m->He, Fe = m->Fe, Hd = m->Hd, Fd = m->Fd;
    int B = enc_batch > 0 ? enc_batch : m->B;          // encoder batch (kzv_encode_images: fewer images than decoder rows)
    const int Me = B * m->Sa, Mp = B * m->npa;
    float* P = m->P;
    const float eps = c.ln_eps;
    m->labels = labels;
    if (hipMemsetAsync(m->count, 0, 64 * sizeof(float), s) != hipSuccess) return kzv_fail(KZV_E_HIP, "forward: memset");
    const int CK = m->Ld * 2 * Hd;
    if (run_encoder) {
    // ---- patch embedding: Conv2d(k=s=16) == im2row + GEMM (trocr_model.py:77,89-90) -----------------
    KZV_TRY(kzv_im2row(px, m->patches, B, c.channels, c.image_h, m->img_w, c.patch_h, c.patch_w, s));
    KZV_TRY(gemm(m->patches, m->PD, m->w_patch, false, Mp, He, m->PD, He, P + m->patch_b, m->pe32, He, KZV_EPI_F32, s));
    float* x0 = m->Le ? m->ea[0].x_in : m->x_last;
    KZV_TRY(kzv_embed_assemble(m->pe32, P + m->cls, P + m->pos, x0, B, m->npa, He, dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_EMB), s,
                               m->img_w / c.patch_w, c.image_w / c.patch_w));
    // ---- ViT layers (pre-LN; HF modeling_vit.py:257-286) -----------------------------------------------
    const bool f8 = m->fp8 != 0;
    if (f8) KZV_TRY(kzv_fp8_roll(m->f8_q, m->f8_amax, m->f8_rows, m->Le, (int)m->f8_stride, s));
    for (int i = 0; i < m->Le; ++i) {
        EncAct& a = m->ea[i];
        const EncLayerP& e = m->ep[i];
        float* x_out = i + 1 < m->Le ? m->ea[i + 1].x_in : m->x_last;
        KZV_TRY(kzv_ln_fwd_ex(a.x_in, P + e.ln1w, P + e.ln1b, a.ln1, nullptr, a.st1, Me, He, eps, 1, 0, 0.f, 0, s, f8 ? m->x8 : nullptr, f8 ? m->x8_scale : nullptr));
        if (f8) KZV_TRY(gemm8(m->x8, He, m->x8_scale, m->w8_qkv[i], Me, 3 * He, He, P + e.qkvb, a.qkv, 3 * He, KZV_EPI_BF16, s));
        else
        KZV_TRY(gemm(a.ln1, He, m->w_eqkv[i], false, Me, 3 * He, He, 3 * He, P + e.qkvb, a.qkv, 3 * He, KZV_EPI_BF16, s));
        KZV_TRY(attn(m, false, 0, a.qkv, 3 * He, a.qkv + He, a.qkv + 2 * He, 3 * He, a.ctx, He, a.lse, nullptr, nullptr, nullptr, nullptr,
                     c.enc_heads, m->Sa, m->Sa, dp(m, c.enc_attn_dropout), key(m, SITE_ENC_L + 4 * i), s, B, He / c.enc_heads));
        KZV_TRY(gemm(a.ctx, He, m->w_eo[i], false, Me, He, He, He, P + e.ob, a.x_mid, He, KZV_EPI_RESID, s, a.x_in, nullptr, 0,
                     dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * i + 1)));
        KZV_TRY(kzv_ln_fwd_ex(a.x_mid, P + e.ln2w, P + e.ln2b, a.ln2, nullptr, a.st2, Me, He, eps, 1, 0, 0.f, 0, s, f8 ? m->x8 : nullptr, f8 ? m->x8_scale : nullptr));
        if (f8) {
            KZV_TRY(gemm8(m->x8, He, m->x8_scale, m->w8_fc1[i], Me, Fe, He, P + e.fc1b, a.act, Fe, KZV_EPI_GELU, s, nullptr, a.pre, Fe, 0.f, 0,
                          m->act8, m->f8_q + i, m->f8_amax + i));
            KZV_TRY(gemm8(m->act8, Fe, m->f8_rows + (int64_t)i * m->f8_stride, m->w8_fc2[i], Me, He, Fe, P + e.fc2b, x_out, He, KZV_EPI_RESID, s,
                          a.x_mid, nullptr, 0, dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * i + 2)));
        } else {
        KZV_TRY(gemm(a.ln2, He, m->w_efc1[i], false, Me, Fe, He, Fe, P + e.fc1b, a.act, Fe, KZV_EPI_GELU, s, nullptr, a.pre, Fe));
        KZV_TRY(gemm(a.act, Fe, m->w_efc2[i], false, Me, He, Fe, He, P + e.fc2b, x_out, He, KZV_EPI_RESID, s, a.x_mid, nullptr, 0,
                     dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * i + 2)));
        }
    }
    // final LN, drop CLS (trocr_model.py:197-200), projection (:269)
    KZV_TRY(kzv_ln_fwd_ex(m->x_last, P + m->lnf_w, P + m->lnf_b, m->enc_out, nullptr, m->stf, Me, He, eps, m->Sa, 1, 0.f, 0, s));
    if (m->has_proj)
        KZV_TRY(gemm(m->enc_out, He, m->w_proj, false, Mp, Hd, He, Hd, P + m->proj_b, m->proj_out, Hd, KZV_EPI_BF16, s));
    // cross-attention K/V of every decoder layer in one GEMM
    KZV_TRY(gemm(m->proj_out, Hd, m->w_ckv, false, Mp, CK, Hd, CK, P + m->ckv_b, m->crosskv, CK, KZV_EPI_BF16, s));
    m->have_enc = true; m->Be = B; m->ckv_dec_ok = false;
    }   // run_encoder
    if (!run_decoder) return KZV_OK;
    B = m->B;
    if (m->Be != B) return kzv_fail(KZV_E_STATE, "forward: the encoder states hold %d images, the decoder batch is %d (kzv_encode_images is for kzv_decode_step only)", m->Be, B);
    const int Md = B * T;
    // ---- decoder embeddings (HF modeling_roberta.py:75-122,142-155) --------------------------------------
    KZV_TRY(kzv_dec_prepare(labels, B, m->L, T, c.pad_id, c.max_pos, m->posids, m->count, m->err, s));
    KZV_TRY(kzv_embed_gather(labels, m->L, m->posids, P + m->word, P + m->dtype, P + m->dpos, m->emb_sum, B, T, Hd, s));
    KZV_TRY(kzv_ln_fwd_ex(m->emb_sum, P + m->eln_w, P + m->eln_b, m->xd0h, m->xd0, m->emb_st, Md, Hd, eps, 1, 0,
                          dp(m, c.dec_hidden_dropout), key(m, SITE_DEC_EMB), s));
    // ---- decoder layers (post-LN; HF modeling_roberta.py:421-464) -------------------------------------------
    const float* x = m->xd0; const bf16_t* xh = m->xd0h;
    // the linear chains between the attentions as two launches per layer (decoder_chain.hip) where the geometry is the reference's
    const bool packable = dec_pack_wanted(m) && kzv_dec_chain_supported(Hd, Fd);
    const bool chain = dec_chain_mode() && packable;
    const bool fused_head = head_ce_mode() && packable && !d_logits;      // LM head + CE in one launch (below)
    if (chain || fused_head) KZV_TRY(ensure_dec_pack(m, s));
    const int64_t HH = (int64_t)Hd * Hd, FH = (int64_t)Fd * Hd, per = 6 * HH + 2 * FH;
    for (int i = 0; i < m->Ld; ++i) {
        DecAct& a = m->da[i];
        const DecLayerP& d = m->dp[i];
        const uint32_t site = SITE_DEC_L + 8 * i;
        if (!chain || i == 0) KZV_TRY(gemm(xh, Hd, m->w_dqkv[i], false, Md, 3 * Hd, Hd, 3 * Hd, P + d.qkvb, a.qkv, 3 * Hd, KZV_EPI_BF16, s));
        KZV_TRY(attn(m, false, 1, a.qkv, 3 * Hd, a.qkv + Hd, a.qkv + 2 * Hd, 3 * Hd, a.ctx, Hd, a.lse_sa, nullptr, nullptr, nullptr, nullptr,
                     c.dec_heads, T, T, dp(m, c.dec_attn_dropout), key(m, site), s));
        if (chain) {
            const bf16_t* wp = m->dec_pack + per * i;
            // the fp32 LayerNorm outputs x1 / x2 / x3 feed nothing but the next residual add: the chains recompute them from the sums and
            // row statistics the backward needs anyway instead of writing and re-reading them (layer 0 adds the embedding output xd0)
            KzvDecChainA ca{a.ctx, i == 0 ? x : nullptr, wp + 3 * HH, P + d.ob, dp(m, c.dec_hidden_dropout), key(m, site + 1), P + d.ln1w, P + d.ln1b, wp + 4 * HH, P + d.cqb,
                            a.s1, a.st1, nullptr, a.x1h, a.cq, Md, eps};
            if (i > 0) { const DecAct& pa = m->da[i - 1]; const DecLayerP& pd = m->dp[i - 1]; ca.xres_s = pa.s3; ca.xres_st = pa.st3; ca.xres_g = P + pd.ln3w; ca.xres_b = P + pd.ln3b; }
            KZV_TRY(kzv_dec_chain_a(ca, s));
        } else {
            KZV_TRY(gemm(a.ctx, Hd, m->w_do[i], false, Md, Hd, Hd, Hd, P + d.ob, a.s1, Hd, KZV_EPI_RESID, s, x, nullptr, 0,
                         dp(m, c.dec_hidden_dropout), key(m, site + 1)));
            KZV_TRY(kzv_ln_fwd_ex(a.s1, P + d.ln1w, P + d.ln1b, a.x1h, a.x1, a.st1, Md, Hd, eps, 1, 0, 0.f, 0, s));
            KZV_TRY(gemm(a.x1h, Hd, m->w_dcq[i], false, Md, Hd, Hd, Hd, P + d.cqb, a.cq, Hd, KZV_EPI_BF16, s));
        }
        KZV_TRY(attn(m, false, 0, a.cq, Hd, m->crosskv + (int64_t)i * 2 * Hd, m->crosskv + (int64_t)i * 2 * Hd + Hd, CK, a.cctx, Hd, a.lse_ca,
                     nullptr, nullptr, nullptr, nullptr, c.dec_heads, T, m->npa, dp(m, c.dec_attn_dropout), key(m, site + 2), s));
        if (chain) {
            const bf16_t* wp = m->dec_pack + per * i;
            const bool more = i + 1 < m->Ld;
            KzvDecChainB cb{a.cctx, nullptr, wp + 5 * HH, P + d.cob, dp(m, c.dec_hidden_dropout), key(m, site + 3), key(m, site + 4), P + d.ln2w, P + d.ln2b,
                            wp + 6 * HH, P + d.fc1b, wp + 6 * HH + FH, P + d.fc2b, P + d.ln3w, P + d.ln3b,
                            more ? m->dec_pack + per * (i + 1) : nullptr, more ? P + m->dp[i + 1].qkvb : nullptr,
                            a.s2, a.st2, nullptr, a.x2h, a.pre, a.act, a.s3, a.st3, nullptr, a.x3h, more ? m->da[i + 1].qkv : nullptr, Md, eps};
            cb.s1 = a.s1; cb.st1 = a.st1; cb.g1 = P + d.ln1w; cb.b1 = P + d.ln1b;
            KZV_TRY(kzv_dec_chain_b(cb, s));
        } else {
            KZV_TRY(gemm(a.cctx, Hd, m->w_dco[i], false, Md, Hd, Hd, Hd, P + d.cob, a.s2, Hd, KZV_EPI_RESID, s, a.x1, nullptr, 0,
                         dp(m, c.dec_hidden_dropout), key(m, site + 3)));
            KZV_TRY(kzv_ln_fwd_ex(a.s2, P + d.ln2w, P + d.ln2b, a.x2h, a.x2, a.st2, Md, Hd, eps, 1, 0, 0.f, 0, s));
            KZV_TRY(gemm(a.x2h, Hd, m->w_dfc1[i], false, Md, Fd, Hd, Fd, P + d.fc1b, a.act, Fd, KZV_EPI_GELU, s, nullptr, a.pre, Fd));
            KZV_TRY(gemm(a.act, Fd, m->w_dfc2[i], false, Md, Hd, Fd, Hd, P + d.fc2b, a.s3, Hd, KZV_EPI_RESID, s, a.x2, nullptr, 0,
                         dp(m, c.dec_hidden_dropout), key(m, site + 4)));
            KZV_TRY(kzv_ln_fwd_ex(a.s3, P + d.ln3w, P + d.ln3b, a.x3h, a.x3, a.st3, Md, Hd, eps, 1, 0, 0.f, 0, s));
        }
        x = a.x3; xh = a.x3h;
    }
    // ---- LM head (HF modeling_roberta.py:877-893; decoder.weight tied to word embeddings :684-687) + CE --------
    KZV_TRY(gemm(xh, Hd, m->w_hd, false, Md, Hd, Hd, Hd, P + m->hd_b, m->hd_gelu, Hd, KZV_EPI_GELU_F32, s, nullptr, m->hd_pre, Hd));
    KZV_TRY(kzv_ln_fwd_ex(m->hd_gelu, P + m->hln_w, P + m->hln_b, m->hd_ln, nullptr, m->hd_st, Md, Hd, eps, 1, 0, 0.f, 0, s));
    // no logits asked for (the training / validation step): head GEMM + log-softmax + NLL + dlogits in ONE launch, the [B*T, Vp] fp32
    // logits never written (decoder_chain.hip head_ce_kernel).  Otherwise the GEMM materialises them and ce_kernel follows.
    if (fused_head) {
        KzvHeadCE hc{m->hd_ln, m->dec_pack + m->head_pack_off, P + m->hbias, labels, m->count, m->loss_acc, m->train ? m->dlogits : nullptr, Md, m->L, T, m->V, (int)m->Vp, c.pad_id};
        static int fuse_dh = -1;     // the head's input gradient inside the same launch (KZV_HEAD_DGRAD=0: the separate GEMM)
        if (fuse_dh < 0) { const char* e = getenv("KZV_HEAD_DGRAD"); fuse_dh = e ? atoi(e) : 1; }
        m->dhln_fused = m->train && fuse_dh && m->w_word.ldt % 8 == 0;
        if (m->dhln_fused) { hc.wpt = m->dec_pack + m->head_tpack_off; hc.dh = m->dhln; }
        KZV_TRY(kzv_head_ce(hc, s));
    } else {
        m->dhln_fused = false;
        KZV_TRY(gemm(m->hd_ln, Hd, m->w_word, false, Md, m->Vp, Hd, m->V, P + m->hbias, m->logits, m->Vp, KZV_EPI_F32, s));
        KZV_TRY(kzv_ce_fwd_bwd(m->logits, m->Vp, labels, m->L, B, T, m->V, c.pad_id, m->count, m->loss_acc, m->train ? m->dlogits : nullptr, s));
    }
    if (d_loss && hipMemcpyAsync(d_loss, m->loss_acc, sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return kzv_fail(KZV_E_HIP, "forward: loss copy");
    if (d_logits && logits_pos < 0) {
        if (T != m->T) return kzv_fail(KZV_E_STATE, "forward_loss: full logits need the full decoder length (kzv_set_active_length(m, L-1))");
        KZV_TRY(kzv_copy_logits(m->logits, m->Vp, d_logits, Md, m->V, s));
    }
    if (d_logits && logits_pos >= 0)   // one position of every sample: rows b*T + pos
        KZV_TRY(kzv_copy_logits(m->logits + (int64_t)logits_pos * m->Vp, (int64_t)T * m->Vp, d_logits, B, m->V, s));
    return KZV_OK;
}

// ============================================================================================== backward
// "dropout(linear(x)) + residual" backward helper: dy = mask(dx) (bf16) + bias grad, weight grad, input grad
int lin_bwd_drop(const float* dx, bf16_t* dy, float* dbias, int M, int N, float drop_p, uint32_t drop_key, hipStream_t s) {
    return kzv_cast_drop_colsum(dx, dy, dbias, M, N, drop_p, drop_key, s);
}

int backward_decoder(kzv_model* m, hipStream_t s) {
    const kzv_config& c = m->c;
    const int B = m->B, T = m->Ta, Hd = m->Hd, Fd = m->Fd, He = m->He;
    const int Mp = B * m->npa, Md = B * T, Me = B * m->Sa;
    float* P = m->P; float* G = m->G;
    const int CK = m->Ld * 2 * Hd;
    m->wbatch.clear();
    // the decoder's input-gradient GEMMs on the row-panel kernel (decoder_chain.hip kzv_dec_lin) where the forward's fragment-ordered
    // packs exist (the reference decoder's geometry, chains on): transposed packs, same layout as the forward's
    static int dgrad_rows = -1;          // dev A/B: KZV_DEC_DGRAD=0 keeps the 128 x 128 kernel for these
    if (dgrad_rows < 0) { const char* e = getenv("KZV_DEC_DGRAD"); dgrad_rows = e ? atoi(e) : 1; }
    const bool rows_dgrad = dgrad_rows && dec_chain_mode() && dec_pack_wanted(m) && kzv_dec_chain_supported(Hd, Fd) && m->dec_pack_ok && !m->use_side;
    const int64_t HHd = (int64_t)Hd * Hd, FHd = (int64_t)Fd * Hd, perd = 6 * HHd + 2 * FHd;
    // measured (profiles/r04): the 256 x 256 products take 7.8 / 12.6 us there against ~16 us on the 128 x 128 kernel; the 768-wide ones
    // (fc2's DGELU output, the K = 768 reductions of fc1 / qkv) are SLOWER on it (35 / 21 us against 22 / 16 - 22): they keep gemm_nt
    static int wide_rows = -1;
    if (wide_rows < 0) { const char* e = getenv("KZV_DEC_DGRAD_WIDE"); wide_rows = e ? atoi(e) : 0; }
    auto tp = [&](int layer, int64_t off) { return (const bf16_t*)(m->dec_pack + m->tpack_off + perd * layer + off); };
    // ---- CE -> LM head ------------------------------------------------------------------------------
    KZV_TRY(wgrad_batch(m, CLS_MISC, s, m->dlogits, m->Vp, m->hd_ln, Hd, G + m->word, Md, m->Vp, Hd, m->V, G + m->hbias));
    if (!m->dhln_fused) KZV_TRY(gemm(m->dlogits, m->Vp, m->w_word, true, Md, Hd, m->Vp, Hd, nullptr, m->dhln, Hd, KZV_EPI_BF16, s));
    KZV_TRY(kzv_ln_bwd_ex(m->dhln, 0, m->hd_gelu, m->hd_st, P + m->hln_w, m->dsum_d, 0, G + m->hln_w, G + m->hln_b, Md, Hd, 1, 0, 0.f, 0, s));
    KZV_TRY(kzv_cast_drop_colsum(m->dsum_d, m->dy_d, G + m->hd_b, Md, Hd, 0.f, 0, s, m->hd_pre));
    const bf16_t* x_last_h = m->Ld ? m->da[m->Ld - 1].x3h : m->xd0h;
    KZV_TRY(wgrad_batch(m, CLS_DY, s, m->dy_d, Hd, x_last_h, Hd, G + m->hd_w, Md, Hd, Hd, Hd, nullptr));
    // the three row-local segments of a layer's backward, one launch each (decoder_chain.hip dec_bwd_seg_kernel): the head dense's input
    // gradient becomes the first GEMM of the top layer's first segment
    const bool segs = rows_dgrad && dec_chain_mode() >= 2 && m->Ld > 0;
    if (segs) {}
    else if (rows_dgrad) KZV_TRY(kzv_dec_lin(m->dy_d, tp(m->Ld, 0), m->dx_d, nullptr, nullptr, Md, Hd, Hd, 1, s));
    else KZV_TRY(gemm(m->dy_d, Hd, m->w_hd, true, Md, Hd, Hd, Hd, nullptr, m->dx_d, Hd, KZV_EPI_F32, s));
    KZV_TRY(wgrad_flush(m, s));          // LM head (tied word embedding) + head dense: before dy_d is rewritten
    // ---- decoder layers, last to first -----------------------------------------------------------------
    for (int i = m->Ld - 1; i >= 0; --i) {
        DecAct& a = m->da[i];
        const DecLayerP& d = m->dp[i];
        const uint32_t site = SITE_DEC_L + 8 * i;
        const bf16_t* xh = i ? m->da[i - 1].x3h : m->xd0h;
        if (segs) {
            const float hp = dp(m, c.dec_hidden_dropout);
            const bool top = i == m->Ld - 1;
            // every "dY" of this layer goes to the layer's own buffers: the 6 x Ld weight gradients are ONE grouped launch behind the loop
            // (36 problems, 288 tiles of 128 x 128 over all tokens instead of six part-filled grids of ten token splits and their atomics)
            // [head dense | the layer above's qkv] -> LN3 -> fc2 (gelu')
            KZV_TRY(kzv_dec_bwd_seg(KzvDecBwdSeg{top ? m->dy_d : m->da[i + 1].g_dqkv, top ? Hd : 3 * Hd, tp(top ? m->Ld : i + 1, 0), top ? nullptr : m->dsum_d,
                                                 a.s3, a.st3, P + d.ln3w, G + d.ln3w, G + d.ln3b, m->dsum_d, a.g_dy, hp, key(m, site + 4),
                                                 tp(i, 6 * HHd + FHd), a.pre, a.g_dbig, Md}, s));
            KZV_TRY(wgrad_batch(m, CLS_DY, s, a.g_dy, Hd, a.act, Fd, G + d.fc2w, Md, Hd, Fd, Hd, G + d.fc2b));
            KZV_TRY(wgrad_batch(m, CLS_DBIG, s, a.g_dbig, Fd, a.x2h, Hd, G + d.fc1w, Md, Fd, Hd, Fd, G + d.fc1b));
            // fc1 -> LN2 -> cross-attention output projection
            KZV_TRY(kzv_dec_bwd_seg(KzvDecBwdSeg{a.g_dbig, Fd, tp(i, 6 * HHd), m->dsum_d, a.s2, a.st2, P + d.ln2w, G + d.ln2w, G + d.ln2b, m->dsum_d, a.g_dy2,
                                                 hp, key(m, site + 3), tp(i, 5 * HHd), nullptr, m->dctx_d, Md}, s));
            KZV_TRY(wgrad_batch(m, CLS_DY, s, a.g_dy2, Hd, a.cctx, Hd, G + d.cow, Md, Hd, Hd, Hd, G + d.cob));
            KZV_TRY(attn(m, true, 0, a.cq, Hd, m->crosskv + (int64_t)i * 2 * Hd, m->crosskv + (int64_t)i * 2 * Hd + Hd, CK, a.cctx, Hd, a.lse_ca,
                         m->dctx_d, a.g_dq, m->dckv + (int64_t)i * 2 * Hd, m->dckv + (int64_t)i * 2 * Hd + Hd, c.dec_heads, T, m->npa,
                         dp(m, c.dec_attn_dropout), key(m, site + 2), s));
            KZV_TRY(wgrad_batch(m, CLS_MISC, s, a.g_dq, Hd, a.x1h, Hd, G + d.cqw, Md, Hd, Hd, Hd, G + d.cqb));
            // cross-attention query -> LN1 -> self-attention output projection
            KZV_TRY(kzv_dec_bwd_seg(KzvDecBwdSeg{a.g_dq, Hd, tp(i, 4 * HHd), m->dsum_d, a.s1, a.st1, P + d.ln1w, G + d.ln1w, G + d.ln1b, m->dsum_d, a.g_dy3,
                                                 hp, key(m, site + 1), tp(i, 3 * HHd), nullptr, m->dctx_d, Md}, s));
            KZV_TRY(wgrad_batch(m, CLS_DY, s, a.g_dy3, Hd, a.ctx, Hd, G + d.ow, Md, Hd, Hd, Hd, G + d.ob));
            KZV_TRY(attn(m, true, 1, a.qkv, 3 * Hd, a.qkv + Hd, a.qkv + 2 * Hd, 3 * Hd, a.ctx, Hd, a.lse_sa, m->dctx_d, a.g_dqkv, a.g_dqkv + Hd,
                         a.g_dqkv + 2 * Hd, c.dec_heads, T, T, dp(m, c.dec_attn_dropout), key(m, site), s));
            KZV_TRY(wgrad_batch(m, CLS_DQKV, s, a.g_dqkv, 3 * Hd, xh, Hd, G + d.qkvw, Md, 3 * Hd, Hd, 3 * Hd, G + d.qkvb));
            if (i == 0) {    // the bottom layer's qkv feeds the embedding LayerNorm: its own launch (a lower layer's first segment takes it otherwise)
                KZV_TRY(gemm(a.g_dqkv, 3 * Hd, m->w_dqkv[i], true, Md, Hd, 3 * Hd, Hd, nullptr, m->dx_d, Hd, KZV_EPI_RESID, s, m->dsum_d));
                KZV_TRY(wgrad_flush(m, s));      // all 6 x Ld weight gradients of the decoder layers
            }
            continue;
        }
        // FFN block: x3 = LN(s3), s3 = x2 + drop(fc2(gelu(fc1(x2))))
        KZV_TRY(wait_cls(m, CLS_DY, s));
        KZV_TRY(kzv_ln_bwd_ex(m->dx_d, 1, a.s3, a.st3, P + d.ln3w, m->dsum_d, 0, G + d.ln3w, G + d.ln3b, Md, Hd, 1, 0, 0.f, 0, s,
                              m->dy_d, dp(m, c.dec_hidden_dropout), key(m, site + 4)));
        KZV_TRY(wgrad_batch(m, CLS_DY, s, m->dy_d, Hd, a.act, Fd, G + d.fc2w, Md, Hd, Fd, Hd, G + d.fc2b));
        KZV_TRY(wait_cls(m, CLS_DBIG, s));
        if (rows_dgrad && wide_rows) KZV_TRY(kzv_dec_lin(m->dy_d, tp(i, 6 * HHd + FHd), m->dbig_d, nullptr, a.pre, Md, Fd, Hd, 2, s));
        else KZV_TRY(gemm(m->dy_d, Hd, m->w_dfc2[i], true, Md, Fd, Hd, Fd, nullptr, m->dbig_d, Fd, KZV_EPI_DGELU, s, nullptr, a.pre, Fd));
        KZV_TRY(wgrad_batch(m, CLS_DBIG, s, m->dbig_d, Fd, a.x2h, Hd, G + d.fc1w, Md, Fd, Hd, Fd, G + d.fc1b));
        if (rows_dgrad && wide_rows) KZV_TRY(kzv_dec_lin(m->dbig_d, tp(i, 6 * HHd), m->dx_d, m->dsum_d, nullptr, Md, Hd, Fd, 1, s));
        else KZV_TRY(gemm(m->dbig_d, Fd, m->w_dfc1[i], true, Md, Hd, Fd, Hd, nullptr, m->dx_d, Hd, KZV_EPI_RESID, s, m->dsum_d));
        // cross-attention block: x2 = LN(s2), s2 = x1 + drop(o(CA(q(x1), kv(enc))))
        KZV_TRY(wait_cls(m, CLS_DY, s));
        bf16_t* dy2 = m->use_side ? m->dy_d : m->dy_d2;      // grouped launch: the three dy of a layer stay alive until its end
        bf16_t* dy3 = m->use_side ? m->dy_d : m->dy_d3;
        KZV_TRY(kzv_ln_bwd_ex(m->dx_d, 1, a.s2, a.st2, P + d.ln2w, m->dsum_d, 0, G + d.ln2w, G + d.ln2b, Md, Hd, 1, 0, 0.f, 0, s,
                              dy2, dp(m, c.dec_hidden_dropout), key(m, site + 3)));
        KZV_TRY(wgrad_batch(m, CLS_DY, s, dy2, Hd, a.cctx, Hd, G + d.cow, Md, Hd, Hd, Hd, G + d.cob));
        KZV_TRY(wait_cls(m, CLS_MISC, s));   // dq_d (and, first layer, dlogits' reader) before the cross-attention backward rewrites dq_d
        if (rows_dgrad) KZV_TRY(kzv_dec_lin(dy2, tp(i, 5 * HHd), m->dctx_d, nullptr, nullptr, Md, Hd, Hd, 0, s));
        else KZV_TRY(gemm(dy2, Hd, m->w_dco[i], true, Md, Hd, Hd, Hd, nullptr, m->dctx_d, Hd, KZV_EPI_BF16, s));
        KZV_TRY(attn(m, true, 0, a.cq, Hd, m->crosskv + (int64_t)i * 2 * Hd, m->crosskv + (int64_t)i * 2 * Hd + Hd, CK, a.cctx, Hd, a.lse_ca,
                     m->dctx_d, m->dq_d, m->dckv + (int64_t)i * 2 * Hd, m->dckv + (int64_t)i * 2 * Hd + Hd, c.dec_heads, T, m->npa,
                     dp(m, c.dec_attn_dropout), key(m, site + 2), s));
        KZV_TRY(wgrad_batch(m, CLS_MISC, s, m->dq_d, Hd, a.x1h, Hd, G + d.cqw, Md, Hd, Hd, Hd, G + d.cqb));
        if (rows_dgrad) KZV_TRY(kzv_dec_lin(m->dq_d, tp(i, 4 * HHd), m->dx_d, m->dsum_d, nullptr, Md, Hd, Hd, 1, s));
        else KZV_TRY(gemm(m->dq_d, Hd, m->w_dcq[i], true, Md, Hd, Hd, Hd, nullptr, m->dx_d, Hd, KZV_EPI_RESID, s, m->dsum_d));
        // self-attention block: x1 = LN(s1), s1 = x + drop(o(SA(qkv(x))))
        KZV_TRY(wait_cls(m, CLS_DY, s));
        KZV_TRY(kzv_ln_bwd_ex(m->dx_d, 1, a.s1, a.st1, P + d.ln1w, m->dsum_d, 0, G + d.ln1w, G + d.ln1b, Md, Hd, 1, 0, 0.f, 0, s,
                              dy3, dp(m, c.dec_hidden_dropout), key(m, site + 1)));
        KZV_TRY(wgrad_batch(m, CLS_DY, s, dy3, Hd, a.ctx, Hd, G + d.ow, Md, Hd, Hd, Hd, G + d.ob));
        KZV_TRY(wait_cls(m, CLS_DQKV, s));
        if (rows_dgrad) KZV_TRY(kzv_dec_lin(dy3, tp(i, 3 * HHd), m->dctx_d, nullptr, nullptr, Md, Hd, Hd, 0, s));
        else KZV_TRY(gemm(dy3, Hd, m->w_do[i], true, Md, Hd, Hd, Hd, nullptr, m->dctx_d, Hd, KZV_EPI_BF16, s));
        KZV_TRY(attn(m, true, 1, a.qkv, 3 * Hd, a.qkv + Hd, a.qkv + 2 * Hd, 3 * Hd, a.ctx, Hd, a.lse_sa, m->dctx_d, m->dqkv_d, m->dqkv_d + Hd,
                     m->dqkv_d + 2 * Hd, c.dec_heads, T, T, dp(m, c.dec_attn_dropout), key(m, site), s));
        KZV_TRY(wgrad_batch(m, CLS_DQKV, s, m->dqkv_d, 3 * Hd, xh, Hd, G + d.qkvw, Md, 3 * Hd, Hd, 3 * Hd, G + d.qkvb));
        if (rows_dgrad && wide_rows) KZV_TRY(kzv_dec_lin(m->dqkv_d, tp(i, 0), m->dx_d, m->dsum_d, nullptr, Md, Hd, 3 * Hd, 1, s));
        else KZV_TRY(gemm(m->dqkv_d, 3 * Hd, m->w_dqkv[i], true, Md, Hd, 3 * Hd, Hd, nullptr, m->dx_d, Hd, KZV_EPI_RESID, s, m->dsum_d));
        KZV_TRY(wgrad_flush(m, s));      // the six weight gradients of this layer in one grid
    }
    // ---- decoder embeddings: x0 = drop(LN(word + type + pos)) ---------------------------------------------
    KZV_TRY(kzv_ln_bwd_ex(m->dx_d, 1, m->emb_sum, m->emb_st, P + m->eln_w, m->dsum_d, 0, G + m->eln_w, G + m->eln_b, Md, Hd, 1, 0,
                          dp(m, c.dec_hidden_dropout), key(m, SITE_DEC_EMB), s));
    KZV_TRY(kzv_embed_scatter_bwd(m->dsum_d, m->labels, m->L, m->posids, G + m->word, G + m->dtype, G + m->dpos, B, T, Hd, c.pad_id, s));
    // ---- cross K/V projection of all layers, encoder_decoder_proj, final encoder LN ---------------------------
    KZV_TRY(wgrad_batch(m, CLS_MISC, s, m->dckv, CK, m->proj_out, Hd, G + m->ckv_w, Mp, CK, Hd, CK, G + m->ckv_b));
    KZV_TRY(gemm(m->dckv, CK, m->w_ckv, true, Mp, Hd, CK, Hd, nullptr, m->denc, Hd, KZV_EPI_BF16, s));
    if (m->has_proj) {
        KZV_TRY(wgrad_batch(m, CLS_DQKV, s, m->denc, Hd, m->enc_out, He, G + m->proj_w, Mp, Hd, He, Hd, G + m->proj_b));
        KZV_TRY(gemm(m->denc, Hd, m->w_proj, true, Mp, He, Hd, He, nullptr, m->denc_out, He, KZV_EPI_BF16, s));
    }
    KZV_TRY(wgrad_flush(m, s));          // cross-attention K/V of all layers + encoder_decoder_proj
    // also emits the masked bf16 copy the top ViT layer's fc2 backward starts from
    KZV_TRY(wait_cls(m, CLS_DY, s));
    const KzvLnBwdF8 f8top{m->dy8, m->dy8_scale, m->dy8_rq, m->dy8_rqinv, m->f8_wnorm ? m->f8_wnorm + (m->Le - 1) : nullptr};
    KZV_TRY(kzv_ln_bwd_ex(m->denc_out, 0, m->x_last, m->stf, P + m->lnf_w, m->dx_e, 0, G + m->lnf_w, G + m->lnf_b, Me, He, m->Sa, 1, 0.f, 0, s,
                          m->Le ? m->dy_e : nullptr, dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * (m->Le - 1) + 2),
                          (m->fp8 >= 2 && m->Le && !m->use_side) ? &f8top : nullptr));
    return KZV_OK;
}

int backward_enc_layer(kzv_model* m, int i, hipStream_t s) {
    const kzv_config& c = m->c;
    const int He = m->He, Fe = m->Fe, Me = m->B * m->Sa;
    float* P = m->P; float* G = m->G;
    EncAct& a = m->ea[i];
    const EncLayerP& e = m->ep[i];
    // the layer's four weight-gradient GEMMs fold their partial tiles in one launch at the end of the layer (one stream only)
    struct TnFolds { KzvTnFoldScope* sc; ~TnFolds() { delete sc; } } tn_folds{m->use_side ? nullptr : new KzvTnFoldScope(s)};
    if (m->side_mode == 2) {
        // Overlap mode 2: the four weight-gradient GEMMs (MFMA-bound, 710 us per layer) run on the side stream ONLY while
        // the caller's stream runs an HBM- or issue-bound kernel (LayerNorm backward x2, attention backward: 344 us per
        // layer); every input-gradient GEMM first joins the side stream, so the gemm_nt kernels never share the machine
        // (their per-launch times stay what they are alone) and nothing MFMA-bound competes with anything MFMA-bound.
        struct SideOk { kzv_model* m; ~SideOk() { m->side_ok = false; } } side_guard{m};
        m->side_ok = true;
        KZV_TRY(join_side(m, s));
        KZV_TRY(gemm(m->dy_e, He, m->w_efc2[i], true, Me, Fe, He, Fe, nullptr, m->dbig_e, Fe, KZV_EPI_DGELU, s, nullptr, a.pre, Fe));
        KZV_TRY(gemm(m->dbig_e, Fe, m->w_efc1[i], true, Me, He, Fe, He, nullptr, m->dh_e, He, KZV_EPI_BF16, s));
        KZV_TRY(wgrad_async(m, CLS_DY, s, m->dy_e, He, a.act, Fe, G + e.fc2w, Me, He, Fe, He, G + e.fc2b));
        KZV_TRY(kzv_ln_bwd_ex(m->dh_e, 0, a.x_mid, a.st2, P + e.ln2w, m->dx_e, 1, G + e.ln2w, G + e.ln2b, Me, He, 1, 0, 0.f, 0, s,
                              m->dy_e2, dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * i + 1)));
        KZV_TRY(join_side(m, s));
        KZV_TRY(gemm(m->dy_e2, He, m->w_eo[i], true, Me, He, He, He, nullptr, m->dctx_e, He, KZV_EPI_BF16, s));
        KZV_TRY(wgrad_async(m, CLS_DY, s, m->dy_e2, He, a.ctx, He, G + e.ow, Me, He, He, He, G + e.ob));
        KZV_TRY(wgrad_async(m, CLS_DBIG, s, m->dbig_e, Fe, a.ln2, He, G + e.fc1w, Me, Fe, He, Fe, G + e.fc1b));
        KZV_TRY(attn(m, true, 0, a.qkv, 3 * He, a.qkv + He, a.qkv + 2 * He, 3 * He, a.ctx, He, a.lse, m->dctx_e, m->dqkv_e, m->dqkv_e + He,
                     m->dqkv_e + 2 * He, c.enc_heads, m->Sa, m->Sa, dp(m, c.enc_attn_dropout), key(m, SITE_ENC_L + 4 * i), s, 0, He / c.enc_heads));
        KZV_TRY(join_side(m, s));
        KZV_TRY(gemm(m->dqkv_e, 3 * He, m->w_eqkv[i], true, Me, He, 3 * He, He, nullptr, m->dh_e, He, KZV_EPI_BF16, s));
        KZV_TRY(wgrad_async(m, CLS_DQKV, s, m->dqkv_e, 3 * He, a.ln1, He, G + e.qkvw, Me, 3 * He, He, 3 * He, G + e.qkvb));
        KZV_TRY(kzv_ln_bwd_ex(m->dh_e, 0, a.x_in, a.st1, P + e.ln1w, m->dx_e, 1, G + e.ln1w, G + e.ln1b, Me, He, 1, 0, 0.f, 0, s,
                              i > 0 ? m->dy_e : nullptr, dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * (i - 1) + 2)));
        return KZV_OK;
    }
    // x_out = x_mid + drop(fc2(gelu(fc1(LN2(x_mid)))))
    // on entry dy_e = dropout-masked bf16 copy of dx_e for this layer's fc2 site (written by the LN backward above it)
    const bool f8g = m->fp8 >= 2 && !m->use_side;       // e4m3 input-gradient GEMMs of the MLP (the weight gradients keep reading bf16)
    if (f8g) {
        KZV_TRY(wgrad_async(m, CLS_DY, s, m->dy_e, He, a.act, Fe, G + e.fc2w, Me, He, Fe, He, G + e.fc2b));
        KZV_TRY(wait_cls(m, CLS_DBIG, s));
        KZV_TRY(gemm8(m->dy8, He, m->dy8_scale, m->w8t_fc2[i], Me, Fe, He, nullptr, m->dbig_e, Fe, KZV_EPI_DGELU, s, nullptr, a.pre, Fe, 0.f, 0,
                      m->dbig8, nullptr, nullptr, m->dy8_rq));
        KZV_TRY(wgrad_async(m, CLS_DBIG, s, m->dbig_e, Fe, a.ln2, He, G + e.fc1w, Me, Fe, He, Fe, G + e.fc1b));
        KZV_TRY(gemm8(m->dbig8, Fe, m->dy8_rqinv, m->w8t_fc1[i], Me, He, Fe, nullptr, m->dh_e, He, KZV_EPI_BF16, s));
    } else {
        // each Linear's input gradient and weight gradient read the same dY: one launch per pair where the 256x256 kernels take both
        KZV_TRY(wait_cls(m, CLS_DBIG, s));
        KZV_TRY(dgrad_wgrad(m, CLS_DY, s, m->dy_e, He, m->w_efc2[i], Me, He, Fe, m->dbig_e, Fe, KZV_EPI_DGELU, a.pre, Fe, a.act, Fe, G + e.fc2w, G + e.fc2b));
        KZV_TRY(dgrad_wgrad(m, CLS_DBIG, s, m->dbig_e, Fe, m->w_efc1[i], Me, Fe, He, m->dh_e, He, KZV_EPI_BF16, nullptr, 0, a.ln2, He, G + e.fc1w, G + e.fc1b));
    }
    KZV_TRY(wait_cls(m, CLS_DY, s));      // dy_e is rewritten below
    KZV_TRY(kzv_ln_bwd_ex(m->dh_e, 0, a.x_mid, a.st2, P + e.ln2w, m->dx_e, 1, G + e.ln2w, G + e.ln2b, Me, He, 1, 0, 0.f, 0, s,
                          m->dy_e, dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * i + 1)));
    // x_mid = x_in + drop(o(attn(qkv(LN1(x_in)))))
    KZV_TRY(dgrad_wgrad(m, CLS_DY, s, m->dy_e, He, m->w_eo[i], Me, He, He, m->dctx_e, He, KZV_EPI_BF16, nullptr, 0, a.ctx, He, G + e.ow, G + e.ob));
    KZV_TRY(wait_cls(m, CLS_DQKV, s));    // dqkv_e is rewritten below
    KZV_TRY(attn(m, true, 0, a.qkv, 3 * He, a.qkv + He, a.qkv + 2 * He, 3 * He, a.ctx, He, a.lse, m->dctx_e, m->dqkv_e, m->dqkv_e + He,
                 m->dqkv_e + 2 * He, c.enc_heads, m->Sa, m->Sa, dp(m, c.enc_attn_dropout), key(m, SITE_ENC_L + 4 * i), s, 0, He / c.enc_heads));
    KZV_TRY(dgrad_wgrad(m, CLS_DQKV, s, m->dqkv_e, 3 * He, m->w_eqkv[i], Me, 3 * He, He, m->dh_e, He, KZV_EPI_BF16, nullptr, 0, a.ln1, He, G + e.qkvw, G + e.qkvb));
    // ... and the masked copy for the fc2 site of the layer below (layer 0 hands fp32 dx_e to the embedding backward)
    KZV_TRY(wait_cls(m, CLS_DY, s));
    const KzvLnBwdF8 f8n{m->dy8, m->dy8_scale, m->dy8_rq, m->dy8_rqinv, (m->f8_wnorm && i > 0) ? m->f8_wnorm + (i - 1) : nullptr};
    KZV_TRY(kzv_ln_bwd_ex(m->dh_e, 0, a.x_in, a.st1, P + e.ln1w, m->dx_e, 1, G + e.ln1w, G + e.ln1b, Me, He, 1, 0, 0.f, 0, s,
                          i > 0 ? m->dy_e : nullptr, dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_L + 4 * (i - 1) + 2), (f8g && i > 0) ? &f8n : nullptr));
    return KZV_OK;
}

int backward_embed(kzv_model* m, hipStream_t s) {
    const kzv_config& c = m->c;
    const int He = m->He, Mp = m->B * m->npa;
    float* G = m->G;
    KZV_TRY(kzv_embed_assemble_bwd(m->dx_e, m->dpatch, G + m->cls, G + m->pos, G + m->patch_b, m->B, m->npa, He,
                                   dp(m, c.enc_hidden_dropout), key(m, SITE_ENC_EMB), s, m->img_w / c.patch_w, c.image_w / c.patch_w));
    KZV_TRY(wgrad_async(m, CLS_MISC, s, m->dpatch, He, m->patches, m->PD, G + m->patch_w, Mp, He, m->PD, He, nullptr));
    return KZV_OK;
}

}  // namespace

// ================================================================================================== C ABI
extern "C" int kzv_model_create(const kzv_config* cfg, kzv_model** out) {
    if (!cfg || !out) return kzv_fail(KZV_E_ARG, "model_create: null");
    const kzv_config& c = *cfg;
    if (c.patch_h <= 0 || c.patch_w <= 0 || c.image_h % c.patch_h || c.image_w % c.patch_w)
        return kzv_fail(KZV_E_ARG, "model_create: image %dx%d not divisible by patch %dx%d", c.image_h, c.image_w, c.patch_h, c.patch_w);
    if (c.enc_heads <= 0 || c.dec_heads <= 0 || c.dec_hidden != 64 * c.dec_heads)
        return kzv_fail(KZV_E_ARG, "model_create: the decoder's head_dim must be 64 (hidden = 64 * heads)");
    if (c.enc_hidden % c.enc_heads || (c.enc_hidden / c.enc_heads) % 8 || c.enc_hidden / c.enc_heads > 128)
        return kzv_fail(KZV_E_ARG, "model_create: the encoder's head_dim must be a multiple of 8 up to 128 (64 takes the MFMA attention kernels)");
    if (c.enc_ffn % 64 || c.dec_ffn % 64 || (c.channels * c.patch_h * c.patch_w) % 64 || c.patch_w % 8)
        return kzv_fail(KZV_E_ARG, "model_create: ffn sizes and C*ph*pw must be multiples of 64, patch_w of 8");
    const int np = (c.image_h / c.patch_h) * (c.image_w / c.patch_w);
    if (np + 1 > 288) return kzv_fail(KZV_E_ARG, "model_create: %d patches + CLS exceed the 288-token attention kernels", np);
    if (c.vocab < 8 || c.max_pos < 4 || c.type_vocab < 1 || c.pad_id < 0 || c.pad_id >= c.vocab)
        return kzv_fail(KZV_E_ARG, "model_create: bad vocabulary geometry");
    if (c.enc_layers < 1 || c.dec_layers < 1) return kzv_fail(KZV_E_ARG, "model_create: encoder and decoder need at least one layer each");
    kzv_model* m = new kzv_model();
    m->c = c;
    m->np = np; m->Se = np + 1; m->PD = c.channels * c.patch_h * c.patch_w;
    m->npa = np; m->Sa = np + 1; m->img_w = c.image_w;
    m->He = c.enc_hidden; m->Fe = c.enc_ffn; m->Hd = c.dec_hidden; m->Fd = c.dec_ffn;
    m->V = c.vocab; m->Vp = (int)align_up(c.vocab, 64); m->Le = c.enc_layers; m->Ld = c.dec_layers;
    m->has_proj = m->He != m->Hd;
    build_param_table(m);
    *out = m;
    return KZV_OK;
}

extern "C" int kzv_model_destroy(kzv_model* m) {
    if (m) {
        if (m->side) (void)hipStreamDestroy(m->side);
        if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
        for (int i = 0; i < 4; ++i) if (m->ev_done[i]) (void)hipEventDestroy(m->ev_done[i]);
        for (int i = 0; i < 2; ++i) if (m->kvc[i]) (void)hipFree(m->kvc[i]);
        for (int i = 0; i < 2; ++i) if (m->rowtab[i]) (void)hipFree(m->rowtab[i]);
        if (m->ckv_dec) (void)hipFree(m->ckv_dec);
        if (m->dec_pack) (void)hipFree(m->dec_pack);
        for (int i = 0; i < 3; ++i) if (m->dgraph[i]) (void)hipGraphExecDestroy(m->dgraph[i]);
    }
    delete m;
    return KZV_OK;
}
extern "C" int kzv_param_count(const kzv_model* m) { return m ? (int)m->table.size() : 0; }
extern "C" int64_t kzv_param_total(const kzv_model* m) { return m ? m->total : 0; }

extern "C" int kzv_param_info(const kzv_model* m, int i, const char** name, int64_t* offset, int64_t* rows, int64_t* cols) {
    if (!m || i < 0 || i >= (int)m->table.size()) return kzv_fail(KZV_E_ARG, "param_info: index out of range");
    const PEntry& e = m->table[i];
    if (name) *name = e.name.c_str();
    if (offset) *offset = e.off;
    if (rows) *rows = e.rows;
    if (cols) *cols = e.cols;
    return KZV_OK;
}

static int check_batch(const kzv_model* m, int batch, int label_len) {
    if (batch <= 0) return kzv_fail(KZV_E_ARG, "batch must be positive");
    if (label_len < 2) return kzv_fail(KZV_E_ARG, "labels need at least 2 columns");
    if (label_len - 1 > 192) return kzv_fail(KZV_E_ARG, "decoder length %d exceeds the 192-token attention kernels", label_len - 1);
    if ((int64_t)batch * m->Se * (int64_t)m->Fe >= (1ll << 31)) return kzv_fail(KZV_E_ARG, "batch too large for 32-bit element indices");
    return KZV_OK;
}

extern "C" int64_t kzv_workspace_bytes(const kzv_model* m, int batch, int label_len) {
    if (!m || check_batch(m, batch, label_len)) return -1;
    kzv_model tmp = *m;   // plan() only writes pointer fields
    tmp.P = nullptr;
    return plan(&tmp, nullptr, batch, label_len);
}

extern "C" int kzv_model_bind(kzv_model* m, float* d_params, float* d_grads, void* d_workspace, int64_t workspace_bytes,
                              int batch, int label_len) {
    if (!m || !d_params || !d_workspace) return kzv_fail(KZV_E_ARG, "model_bind: null");
    KZV_TRY(check_batch(m, batch, label_len));
    if (((uintptr_t)d_params | (uintptr_t)d_grads | (uintptr_t)d_workspace) & 255) return kzv_fail(KZV_E_ARG, "model_bind: buffers must be 256-byte aligned");
    m->P = d_params; m->G = d_grads;
    const int64_t need = plan(m, (char*)d_workspace, batch, label_len);
    if (workspace_bytes < need) return kzv_fail(KZV_E_ARG, "model_bind: workspace %lld < required %lld", (long long)workspace_bytes, (long long)need);
    m->ws = (char*)d_workspace; m->ws_bytes = workspace_bytes;
    m->B = batch; m->L = label_len; m->T = label_len - 1; m->Ta = m->T;
    if (!kzv_zero_page()) return kzv_fail(KZV_E_HIP, "model_bind: zero page");
    // zero the bf16 weight region once (padding of transposed copies must read as 0), upload descriptors
    const int64_t wbytes = (char*)m->d_desc - (char*)d_workspace;
    if (hipMemset(d_workspace, 0, wbytes) != hipSuccess) return kzv_fail(KZV_E_HIP, "model_bind: memset");
    if (hipMemcpy(m->d_desc, m->h_desc.data(), sizeof(KzvCastDesc) * m->ndesc, hipMemcpyHostToDevice) != hipSuccess)
        return kzv_fail(KZV_E_HIP, "model_bind: descriptor upload");
    if (m->fp8) {
        if (hipMemcpy(m->d_qdesc, m->h_qdesc.data(), sizeof(KzvQuantDesc) * m->nqdesc, hipMemcpyHostToDevice) != hipSuccess)
            return kzv_fail(KZV_E_HIP, "model_bind: fp8 descriptor upload");
        const std::vector<float> ones((size_t)m->Le, 1.f);
        if (hipMemcpy(m->f8_q, ones.data(), sizeof(float) * m->Le, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(m->f8_amax, 0, sizeof(float) * m->Le) != hipSuccess)
            return kzv_fail(KZV_E_HIP, "model_bind: fp8 scale state");
    }
    if (!m->side) {
        const char* e = getenv("KZV_SIDE_STREAM");
        // KZV_SIDE_STREAM: 0 (default) = one stream; 1 = weight gradients free-running on a side stream: +3 % img/s
        // (7,180 -> 7,410), but the co-running kernels stretch each other (gemm_nt family 920 -> 715 TFLOP/s per launch),
        // so the per-kernel roofline accounting is only meaningful with it off; 2 = encoder weight gradients only under
        // the LayerNorm / attention backward kernels, every input-gradient GEMM joining the side stream first: the 120
        // cross-stream waits per step cost more than the overlap returns (6,400 img/s) -- kept for the record.
        m->side_mode = (e && e[0]) ? atoi(e) : 0;
        m->use_side = m->side_mode != 0;
        if (m->use_side) {
            // (stream priorities make no measurable difference here: the range on this part is {0, -1})
            if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess) return kzv_fail(KZV_E_HIP, "model_bind: side stream");
            if (hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess) return kzv_fail(KZV_E_HIP, "model_bind: event");
            for (int i = 0; i < 4; ++i)
                if (hipEventCreateWithFlags(&m->ev_done[i], hipEventDisableTiming) != hipSuccess) return kzv_fail(KZV_E_HIP, "model_bind: event");
        }
    }
    // a captured decode step holds pointers INTO the workspace and the parameter buffer: none survives a rebind
    for (int i = 0; i < 3; ++i) if (m->dgraph[i]) { (void)hipGraphExecDestroy(m->dgraph[i]); m->dgraph[i] = nullptr; }
    m->ckv_dec_ok = false; m->dec_pack_ok = false;
    m->bound = true; m->have_fwd = false; m->have_enc = false;
    return KZV_OK;
}

extern "C" int kzv_model_sync_weights(kzv_model* m, void* stream) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "sync_weights: model not bound");
    KZV_TRY(kzv_cast_weights(m->d_desc, m->ndesc, m->cast_tiles, (hipStream_t)stream));
    m->dec_pack_ok = false;
    if (m->fp8) {
        if (m->fp8 >= 2 && hipMemsetAsync(m->f8_wnorm, 0, sizeof(float) * m->Le, (hipStream_t)stream) != hipSuccess)
            return kzv_fail(KZV_E_HIP, "sync_weights: memset");
        KZV_TRY(kzv_quant_rows(m->d_qdesc, m->nqdesc, m->qrows, (hipStream_t)stream));
    }
    return KZV_OK;
}

// fp8 weight path on / off; before kzv_model_bind (the workspace layout depends on it).
extern "C" int kzv_set_fp8(kzv_model* m, int mode) {
    if (!m) return kzv_fail(KZV_E_ARG, "set_fp8: null model");
    if (m->bound) return kzv_fail(KZV_E_STATE, "set_fp8: call before kzv_model_bind");
    if (mode < 0 || mode > 2) return kzv_fail(KZV_E_ARG, "set_fp8: mode 0 (bf16), 1 (e4m3 forward GEMMs of the encoder) or 2 (+ the MLP's input-gradient GEMMs)");
    if (mode && (m->He % 256 || m->Fe % 256))
        return kzv_fail(KZV_E_ARG, "set_fp8: encoder hidden %d and ffn %d must be multiples of 256 (128-byte K-tiles in pairs)", m->He, m->Fe);
    m->fp8 = mode;
    return KZV_OK;
}
extern "C" int kzv_get_fp8(const kzv_model* m) { return m ? m->fp8 : 0; }

// parity hook: the per-tensor multipliers the LAST forward quantised each layer's GELU output with -> d_out[enc_layers]
extern "C" int kzv_fp8_act_scales(const kzv_model* m, float* d_out, void* stream) {
    if (!m || !m->bound || !m->fp8 || !d_out) return kzv_fail(KZV_E_STATE, "fp8_act_scales: needs a bound model with the fp8 path on");
    if (hipMemcpyAsync(d_out, m->f8_q, sizeof(float) * m->Le, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return kzv_fail(KZV_E_HIP, "fp8_act_scales: copy");
    return KZV_OK;
}

extern "C" int kzv_forward_loss(kzv_model* m, const float* d_pixel_values, const int64_t* d_labels, float* d_loss,
                                float* d_logits, int train, uint64_t seed, void* stream) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "forward_loss: model not bound");
    if (!d_pixel_values || !d_labels) return kzv_fail(KZV_E_ARG, "forward_loss: null input");
    m->train = train != 0; m->seed = seed;
    const int rc = forward(m, d_pixel_values, d_labels, d_loss, d_logits, (hipStream_t)stream);
    m->have_fwd = rc == KZV_OK && m->train;
    return rc;
}

extern "C" int kzv_set_image_width(kzv_model* m, int width) {
    if (!m) return kzv_fail(KZV_E_STATE, "set_image_width: null model");
    const kzv_config& c = m->c;
    if (width < c.patch_w || width > c.image_w || width % c.patch_w)
        return kzv_fail(KZV_E_ARG, "set_image_width: %d is not a multiple of the patch width %d within %d..%d", width, c.patch_w, c.patch_w, c.image_w);
    if (width != m->img_w) { m->have_fwd = false; m->have_enc = false; }    // saved activations belong to the old geometry
    m->img_w = width;
    m->npa = (c.image_h / c.patch_h) * (width / c.patch_w);
    m->Sa = m->npa + 1;
    return KZV_OK;
}

extern "C" int kzv_check_positions(kzv_model* m, void* stream) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "check_positions: model not bound");
    int flag = 0;
    if (hipMemcpyAsync(&flag, m->err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return kzv_fail(KZV_E_HIP, "check_positions: copy");
    if (flag) return kzv_fail(KZV_E_ARG, "labels too long: a position id reached max_position_embeddings = %d (index out of range in the reference)", m->c.max_pos);
    return KZV_OK;
}

extern "C" int kzv_encode_images(kzv_model* m, const float* d_pixel_values, int n_images, void* stream) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "encode_images: model not bound");
    if (!d_pixel_values || n_images < 1 || n_images > m->B || m->B % n_images)
        return kzv_fail(KZV_E_ARG, "encode_images: 1 <= n_images <= bound batch %d, which must be a multiple of it", m->B);
    m->train = false; m->seed = 0; m->have_fwd = false;
    return forward(m, d_pixel_values, nullptr, nullptr, nullptr, (hipStream_t)stream, true, -1, n_images, false);
}

extern "C" int kzv_set_active_length(kzv_model* m, int t_active) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "set_active_length: model not bound");
    if (t_active < 1 || t_active > m->T) return kzv_fail(KZV_E_ARG, "set_active_length: must be in 1..%d", m->T);
    if (m->have_fwd && t_active != m->Ta) m->have_fwd = false;   // saved activations belong to the old length
    m->Ta = t_active;
    return KZV_OK;
}

extern "C" int kzv_decode_logits(kzv_model* m, const int64_t* d_labels, int pos, float* d_logits, void* stream) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "decode_logits: model not bound");
    if (!m->have_enc) return kzv_fail(KZV_E_STATE, "decode_logits: call kzv_forward_loss on the images first");
    if (!d_labels || !d_logits || pos < 0 || pos >= m->Ta) return kzv_fail(KZV_E_ARG, "decode_logits: position outside the active decoder length");
    m->train = false;
    m->have_fwd = false;     // decoder activations are overwritten: no backward after this
    return forward(m, nullptr, d_labels, nullptr, d_logits, (hipStream_t)stream, false, pos);
}

// ---- KV-cached generation step (N1) -------------------------------------------------------------------------------
static int ensure_kv_cache(kzv_model* m) {
    if (m->kvc[0] && m->kvB == m->B && m->kvT == m->T) return KZV_OK;
    for (int i = 0; i < 2; ++i) { if (m->kvc[i]) (void)hipFree(m->kvc[i]); m->kvc[i] = nullptr; }
    for (int i = 0; i < 2; ++i) { if (m->rowtab[i]) (void)hipFree(m->rowtab[i]); m->rowtab[i] = nullptr; }
    // ONE cache [2*Ld][B][T][Hd]: beam steps re-parent rows through the row tables instead of copying into a second cache
    const size_t bytes = (size_t)2 * m->Ld * m->B * m->T * m->Hd * sizeof(bf16_t);
    if (hipMalloc((void**)&m->kvc[0], bytes) != hipSuccess) return kzv_fail(KZV_E_HIP, "decode_step: KV cache allocation (%zu bytes)", bytes);
    for (int i = 0; i < 2; ++i)
        if (hipMalloc((void**)&m->rowtab[i], (size_t)m->B * m->T * sizeof(int)) != hipSuccess) return kzv_fail(KZV_E_HIP, "decode_step: row table allocation");
    m->kvB = m->B; m->kvT = m->T; m->kv_cur = 0; m->rt_cur = -1;
    for (int i = 0; i < 3; ++i) if (m->dgraph[i]) { (void)hipGraphExecDestroy(m->dgraph[i]); m->dgraph[i] = nullptr; }
    return KZV_OK;
}

// the decode-layout copy of the cross-attention K/V of the images encoded last (once per generation)
static int ensure_cross_layout(kzv_model* m, hipStream_t s) {
    if (m->ckv_dec_ok) return KZV_OK;
    const size_t bytes = (size_t)m->Ld * 2 * m->Be * m->npa * m->Hd * sizeof(bf16_t);
    if (bytes > m->ckv_dec_bytes) {
        if (m->ckv_dec) (void)hipFree(m->ckv_dec);
        m->ckv_dec = nullptr; m->ckv_dec_bytes = 0;
        if (hipMalloc((void**)&m->ckv_dec, bytes) != hipSuccess) return kzv_fail(KZV_E_HIP, "decode: cross K/V copy allocation (%zu bytes)", bytes);
        m->ckv_dec_bytes = bytes;
        for (int i = 0; i < 3; ++i) if (m->dgraph[i]) { (void)hipGraphExecDestroy(m->dgraph[i]); m->dgraph[i] = nullptr; }
    }
    KZV_TRY(kzv_cross_relayout(m->crosskv, m->ckv_dec, m->Be, m->npa, m->c.dec_heads, 2 * m->Ld, s));
    m->ckv_dec_ok = true;
    return KZV_OK;
}

// one decoder step for the newest token of every sequence; tptr != nullptr: the step index is read from device memory
// (graph replay), `t` is then only the host's copy for argument checks
// a decoder sub-layer's output GEMM and the LayerNorm after it.  (Fusing the two for N = 256 -- one 16-wave workgroup per 16
// rows, wave w finishing row w -- was built and measured in round 2: bit-identical, and 100 us per step SLOWER: 16..64 large
// workgroups lose more to launch and single-CU load paths than the 19 saved ~5-us LayerNorm launches return.)
static int gemm_ln(kzv_model* m, const bf16_t* A, int64_t lda, const W16& w, int M, int K, const float* bias, const float* resid, bool gelu,
                   float* tmp32, bf16_t* aux_tmp, const float* gamma, const float* beta, bf16_t* y16, float* y32, float* stats, hipStream_t s) {
    const int Hd = m->Hd;
    if (gelu) KZV_TRY(gemm(A, lda, w, false, M, Hd, K, Hd, bias, tmp32, Hd, KZV_EPI_GELU_F32, s, nullptr, aux_tmp, Hd));
    else KZV_TRY(gemm(A, lda, w, false, M, Hd, K, Hd, bias, tmp32, Hd, KZV_EPI_RESID, s, resid, nullptr, 0, 0.f, 0));
    return kzv_ln_fwd_ex(tmp32, gamma, beta, y16, y32, stats, M, Hd, m->c.ln_eps, 1, 0, 0.f, 0, s);
}

// a generation-step GEMM whose A operand is LN(ln_a) and / or whose RESID residual is LN(ln_r) (hidden size 256: gemm_rows.hip)
static int gemm_f(const bf16_t* A, int64_t lda, const W16& w, int M, int N, int K, int n_valid, const float* bias, void* C, int64_t ldc, int epi,
                  hipStream_t s, void* aux, int64_t ldaux, const float* ln_a, const float* ga, const float* ba, const float* ln_r, const float* gr,
                  const float* br, float eps) {
    kzv_gemm_rows_ln_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.B = w.w; a.ldb = K; a.C = C; a.ldc = ldc; a.bias = bias; a.aux = aux; a.ldaux = ldaux;
    a.M = M; a.N = N; a.K = K; a.n_valid = n_valid;
    a.ln_a = ln_a; a.ln_a_gamma = ga; a.ln_a_beta = ba; a.ln_r = ln_r; a.ln_r_gamma = gr; a.ln_r_beta = br; a.eps = eps;
    return kzv_gemm_rows_ln(&a, epi, s);
}

// Hidden size 256: no LayerNorm launch at all -- each sub-layer output (fp32) stays un-normalised in memory and its two consumers
// (one GEMM's A operand, one later residual add) normalise it themselves (gemm_rows_ln_kernel).  20 launches fewer per token.
static int decode_step_body_fused(kzv_model* m, const int64_t* d_tokens, const int* d_posids, int t, const int* tptr, const unsigned char* d_valid,
                                  int64_t ld_valid, float* d_logits, hipStream_t s) {
    const kzv_config& c = m->c;
    const int B = m->B, Hd = m->Hd, Fd = m->Fd, T = m->T;
    float* P = m->P;
    const float eps = c.ln_eps;
    bf16_t* cache = m->kvc[m->kv_cur];
    const int64_t plane = (int64_t)B * T * Hd;
    KZV_TRY(kzv_embed_gather(d_tokens, 1, d_posids, P + m->word, P + m->dtype, P + m->dpos, m->emb_sum, B, 1, Hd, s));
    const float* src = m->emb_sum; const float* sg = P + m->eln_w; const float* sb = P + m->eln_b;      // x = LN(src; sg, sb), never stored
    for (int i = 0; i < m->Ld; ++i) {
        DecAct& a = m->da[i];
        const DecLayerP& d = m->dp[i];
        KZV_TRY(gemm_f(nullptr, 0, m->w_dqkv[i], B, 3 * Hd, Hd, 3 * Hd, P + d.qkvb, a.qkv, 3 * Hd, KZV_EPI_BF16, s, nullptr, 0, src, sg, sb, nullptr, nullptr, nullptr, eps));
        KZV_TRY(kzv_attn_decode(a.qkv, 3 * Hd, a.qkv + Hd, a.qkv + 2 * Hd, 3 * Hd, cache + (int64_t)(2 * i) * plane, cache + (int64_t)(2 * i + 1) * plane,
                                (int64_t)T * Hd, 64, d_valid, ld_valid, a.ctx, Hd, B, c.dec_heads, tptr ? T : t + 1, t, s, tptr, 1,
                                m->rt_cur >= 0 ? m->rowtab[m->rt_cur] : nullptr, T, (int64_t)T * 64));
        KZV_TRY(gemm_f(a.ctx, Hd, m->w_do[i], B, Hd, Hd, Hd, P + d.ob, a.s1, Hd, KZV_EPI_RESID, s, nullptr, 0, nullptr, nullptr, nullptr, src, sg, sb, eps));
        KZV_TRY(gemm_f(nullptr, 0, m->w_dcq[i], B, Hd, Hd, Hd, P + d.cqb, a.cq, Hd, KZV_EPI_BF16, s, nullptr, 0, a.s1, P + d.ln1w, P + d.ln1b, nullptr, nullptr, nullptr, eps));
        {
            const int64_t img = (int64_t)m->npa * Hd, plane2 = (int64_t)m->Be * img;
            KZV_TRY(kzv_attn_decode(a.cq, Hd, nullptr, nullptr, 0, m->ckv_dec + (int64_t)(2 * i) * plane2, m->ckv_dec + (int64_t)(2 * i + 1) * plane2,
                                    img, 64, nullptr, 0, a.cctx, Hd, B, c.dec_heads, m->npa, -1, s, nullptr, B / m->Be, nullptr, 0, (int64_t)m->npa * 64));
        }
        KZV_TRY(gemm_f(a.cctx, Hd, m->w_dco[i], B, Hd, Hd, Hd, P + d.cob, a.s2, Hd, KZV_EPI_RESID, s, nullptr, 0, nullptr, nullptr, nullptr, a.s1, P + d.ln1w, P + d.ln1b, eps));
        KZV_TRY(gemm_f(nullptr, 0, m->w_dfc1[i], B, Fd, Hd, Fd, P + d.fc1b, a.act, Fd, KZV_EPI_GELU, s, a.pre, Fd, a.s2, P + d.ln2w, P + d.ln2b, nullptr, nullptr, nullptr, eps));
        KZV_TRY(gemm_f(a.act, Fd, m->w_dfc2[i], B, Hd, Fd, Hd, P + d.fc2b, a.s3, Hd, KZV_EPI_RESID, s, nullptr, 0, nullptr, nullptr, nullptr, a.s2, P + d.ln2w, P + d.ln2b, eps));
        src = a.s3; sg = P + d.ln3w; sb = P + d.ln3b;
    }
    KZV_TRY(gemm_f(nullptr, 0, m->w_hd, B, Hd, Hd, Hd, P + m->hd_b, m->hd_gelu, Hd, KZV_EPI_GELU_F32, s, m->hd_pre, Hd, src, sg, sb, nullptr, nullptr, nullptr, eps));
    if (m->V % 4 == 0)
        return gemm_f(nullptr, 0, m->w_word, B, m->V, Hd, m->V, P + m->hbias, d_logits, m->V, KZV_EPI_F32, s, nullptr, 0, m->hd_gelu, P + m->hln_w, P + m->hln_b,
                      nullptr, nullptr, nullptr, eps);
    KZV_TRY(gemm_f(nullptr, 0, m->w_word, B, m->Vp, Hd, m->V, P + m->hbias, m->logits, m->Vp, KZV_EPI_F32, s, nullptr, 0, m->hd_gelu, P + m->hln_w, P + m->hln_b,
                   nullptr, nullptr, nullptr, eps));
    return kzv_copy_logits(m->logits, m->Vp, d_logits, B, m->V, s);
}

static int g_decode_one_launch = -1;           // -1: KZV_DECODE_ONE_LAUNCH (default 1)
static int decode_one_launch_mode() {
    if (g_decode_one_launch < 0) { const char* e = getenv("KZV_DECODE_ONE_LAUNCH"); g_decode_one_launch = e ? (atoi(e) != 0) : 1; }
    return g_decode_one_launch;
}
extern "C" int kzv_set_dec_chain(int on) {
    if (on < -1 || on > 2) return kzv_fail(KZV_E_ARG, "set_dec_chain: -1 (environment default), 0, 1 or 2");
    g_dec_chain = on;
    return KZV_OK;
}
extern "C" int kzv_set_head_ce(int on) {
    if (on < -1 || on > 1) return kzv_fail(KZV_E_ARG, "set_head_ce: -1 (environment default), 0 or 1");
    g_head_ce = on;
    return KZV_OK;
}
extern "C" int kzv_set_decode_one_launch(int on) {
    if (on < -1 || on > 1) return kzv_fail(KZV_E_ARG, "set_decode_one_launch: -1 (environment default), 0 or 1");
    g_decode_one_launch = on;
    return KZV_OK;
}
static bool decode_one_launch(const kzv_model* m) {
    return decode_one_launch_mode() && m->Be >= 1 && m->B % m->Be == 0 && kzv_decode_fused_supported(m->Hd, m->c.dec_heads, m->Fd, m->Ld, m->B / m->Be, m->T, m->npa);
}

// fragment-ordered copies of the decoder's weights (9.6 MB) for the one-launch generation step and the training forward's linear
// chains: ONE table-driven launch after every weight change (outside any capture)
namespace {
bool dec_pack_wanted(const kzv_model* m) {
    return m->Hd == 256 && m->c.dec_heads == 4 && m->Fd == 768 && m->Ld >= 1 && m->Ld <= KZV_DECODE_FUSED_MAX_LAYERS;
}
int ensure_dec_pack(kzv_model* m, hipStream_t s) {
    if (m->dec_pack_ok || !dec_pack_wanted(m)) return KZV_OK;
    const int64_t Hd = m->Hd, Fd = m->Fd, per = 3 * Hd * Hd + 3 * Hd * Hd + 2 * Fd * Hd;
    const int64_t vq = (m->V + 255) / 256 * 256;                 // the tied LM-head weight in 256-row chunks, zero rows beyond the vocabulary (head_ce_kernel)
    m->head_pack_off = per * m->Ld + Hd * Hd;
    if (!m->dec_pack) {
        if (hipMalloc((void**)&m->dec_pack, sizeof(bf16_t) * (size_t)(2 * (per * m->Ld + Hd * Hd) + 2 * vq * Hd)) != hipSuccess) return kzv_fail(KZV_E_HIP, "decode: weight pack allocation");
        for (int i = 0; i < 3; ++i) if (m->dgraph[i]) { (void)hipGraphExecDestroy(m->dgraph[i]); m->dgraph[i] = nullptr; }
    }
    std::vector<KzvPackJob> jobs;
    for (int i = 0; i < m->Ld; ++i) {
        bf16_t* o = m->dec_pack + per * i;
        jobs.push_back({m->w_dqkv[i].w, o, 3 * (int)Hd, (int)Hd}); o += 3 * Hd * Hd;
        jobs.push_back({m->w_do[i].w, o, (int)Hd, (int)Hd}); o += Hd * Hd;
        jobs.push_back({m->w_dcq[i].w, o, (int)Hd, (int)Hd}); o += Hd * Hd;
        jobs.push_back({m->w_dco[i].w, o, (int)Hd, (int)Hd}); o += Hd * Hd;
        jobs.push_back({m->w_dfc1[i].w, o, (int)Fd, (int)Hd}); o += Fd * Hd;
        jobs.push_back({m->w_dfc2[i].w, o, (int)Hd, (int)Fd});
    }
    jobs.push_back({m->w_hd.w, m->dec_pack + per * m->Ld, (int)Hd, (int)Hd});
    jobs.push_back({m->w_word.w, m->dec_pack + m->head_pack_off, (int)vq, (int)Hd, m->V});
    // the TRANSPOSED copies (B operands of the input-gradient GEMMs, kzv_dec_lin): [in, out] row-major, dense because out % 64 == 0
    m->tpack_off = m->head_pack_off + vq * Hd;
    for (int i = 0; i < m->Ld; ++i) {
        bf16_t* o = m->dec_pack + m->tpack_off + per * i;
        jobs.push_back({m->w_dqkv[i].wt, o, (int)Hd, 3 * (int)Hd}); o += 3 * Hd * Hd;
        jobs.push_back({m->w_do[i].wt, o, (int)Hd, (int)Hd}); o += Hd * Hd;
        jobs.push_back({m->w_dcq[i].wt, o, (int)Hd, (int)Hd}); o += Hd * Hd;
        jobs.push_back({m->w_dco[i].wt, o, (int)Hd, (int)Hd}); o += Hd * Hd;
        jobs.push_back({m->w_dfc1[i].wt, o, (int)Hd, (int)Fd}); o += Fd * Hd;
        jobs.push_back({m->w_dfc2[i].wt, o, (int)Fd, (int)Hd});
    }
    jobs.push_back({m->w_hd.wt, m->dec_pack + m->tpack_off + per * m->Ld, (int)Hd, (int)Hd});
    // the tied LM-head weight transposed ([hidden, vocabulary]: the B operand of the head's input gradient inside head_ce_kernel), columns
    // beyond the padded vocabulary as zeros
    m->head_tpack_off = m->tpack_off + per * m->Ld + Hd * Hd;
    jobs.push_back({m->w_word.wt, m->dec_pack + m->head_tpack_off, (int)Hd, (int)vq, (int)Hd, (int)m->w_word.ldt, (int)m->Vp});
    KZV_TRY(kzv_pack_frag_multi(jobs.data(), (int)jobs.size(), s));
    m->dec_pack_ok = true;
    return KZV_OK;
}
}  // namespace

// The whole step up to the LM head's dense layer in ONE launch (decode_fused.hip: a workgroup per image owns its beams through all
// layers), then the vocabulary GEMM with the head's LayerNorm folded into its A operand as before.
static int decode_step_body_one_launch(kzv_model* m, const int64_t* d_tokens, const int* d_posids, int t, const int* tptr, const unsigned char* d_valid,
                                       int64_t ld_valid, float* d_logits, hipStream_t s) {
    const int B = m->B, Hd = m->Hd, T = m->T;
    float* P = m->P;
    KzvDecodeFused a;
    memset(&a, 0, sizeof(a));
    if (!m->dec_pack_ok) return kzv_fail(KZV_E_STATE, "decode_step: the fragment-ordered decoder weights are stale");
    const int64_t HH = (int64_t)Hd * Hd, FH = (int64_t)m->Fd * Hd, per = 6 * HH + 2 * FH;
    for (int i = 0; i < m->Ld; ++i) {
        const DecLayerP& d = m->dp[i];
        const bf16_t* o = m->dec_pack + per * i;
        a.layers[i] = KzvDecodeFusedLayer{o, o + 3 * HH, o + 4 * HH, o + 5 * HH, o + 6 * HH, o + 6 * HH + FH,
                                          P + d.qkvb, P + d.ob, P + d.cqb, P + d.cob, P + d.fc1b, P + d.fc2b,
                                          P + d.ln1w, P + d.ln1b, P + d.ln2w, P + d.ln2b, P + d.ln3w, P + d.ln3b};
    }
    a.nlayers = m->Ld; a.tokens = d_tokens; a.posids = d_posids;
    a.word = P + m->word; a.type0 = P + m->dtype; a.postab = P + m->dpos; a.elnw = P + m->eln_w; a.elnb = P + m->eln_b;
    a.whd = m->dec_pack + per * m->Ld; a.bhd = P + m->hd_b; a.hd_out = m->hd_gelu;
    a.cache = m->kvc[m->kv_cur]; a.plane = (int64_t)B * T * Hd;
    a.ckv = m->ckv_dec; a.plane2 = (int64_t)m->Be * m->npa * Hd;
    a.valid = d_valid; a.ldvalid = ld_valid; a.tptr = tptr; a.t = t; a.T = T; a.npa = m->npa; a.B = B; a.group = B / m->Be;
    a.rows = m->rt_cur >= 0 ? m->rowtab[m->rt_cur] : nullptr; a.eps = m->c.ln_eps;
    KZV_TRY(kzv_decode_fused_launch(a, s));
    const float eps = m->c.ln_eps;
    if (m->V % 4 == 0)
        return gemm_f(nullptr, 0, m->w_word, B, m->V, Hd, m->V, P + m->hbias, d_logits, m->V, KZV_EPI_F32, s, nullptr, 0, m->hd_gelu, P + m->hln_w, P + m->hln_b,
                      nullptr, nullptr, nullptr, eps);
    KZV_TRY(gemm_f(nullptr, 0, m->w_word, B, m->Vp, Hd, m->V, P + m->hbias, m->logits, m->Vp, KZV_EPI_F32, s, nullptr, 0, m->hd_gelu, P + m->hln_w, P + m->hln_b,
                   nullptr, nullptr, nullptr, eps));
    return kzv_copy_logits(m->logits, m->Vp, d_logits, B, m->V, s);
}

static int decode_step_body(kzv_model* m, const int64_t* d_tokens, const int* d_posids, int t, const int* tptr, const unsigned char* d_valid,
                            int64_t ld_valid, float* d_logits, hipStream_t s) {
    const kzv_config& c = m->c;
    const int B = m->B, Hd = m->Hd, Fd = m->Fd, T = m->T;
    float* P = m->P;
    const float eps = c.ln_eps;
    static int fuse_ln = -1;
    if (fuse_ln < 0) { const char* e = getenv("KZV_DECODE_FUSE_LN"); fuse_ln = e ? atoi(e) : 1; }
    if (decode_one_launch(m)) return decode_step_body_one_launch(m, d_tokens, d_posids, t, tptr, d_valid, ld_valid, d_logits, s);
    if (fuse_ln && Hd == 256 && B <= 4096) return decode_step_body_fused(m, d_tokens, d_posids, t, tptr, d_valid, ld_valid, d_logits, s);
    KzvRowsScope rows_scope;                     // M = B rows: every GEMM of the step takes the few-rows kernel (gemm_rows.hip)
    bf16_t* cache = m->kvc[m->kv_cur];
    const int64_t plane = (int64_t)B * T * Hd;  // one layer's K (or V) cache
    // embeddings of the one new token per sequence (HF modeling_roberta.py:75-122; position ids from the caller)
    KZV_TRY(kzv_embed_gather(d_tokens, 1, d_posids, P + m->word, P + m->dtype, P + m->dpos, m->emb_sum, B, 1, Hd, s));
    KZV_TRY(kzv_ln_fwd_ex(m->emb_sum, P + m->eln_w, P + m->eln_b, m->xd0h, m->xd0, m->emb_st, B, Hd, eps, 1, 0, 0.f, 0, s));
    const float* x = m->xd0; const bf16_t* xh = m->xd0h;
    for (int i = 0; i < m->Ld; ++i) {
        DecAct& a = m->da[i];
        const DecLayerP& d = m->dp[i];
        KZV_TRY(gemm(xh, Hd, m->w_dqkv[i], false, B, 3 * Hd, Hd, 3 * Hd, P + d.qkvb, a.qkv, 3 * Hd, KZV_EPI_BF16, s));
        KZV_TRY(kzv_attn_decode(a.qkv, 3 * Hd, a.qkv + Hd, a.qkv + 2 * Hd, 3 * Hd, cache + (int64_t)(2 * i) * plane, cache + (int64_t)(2 * i + 1) * plane,
                                (int64_t)T * Hd, 64, d_valid, ld_valid, a.ctx, Hd, B, c.dec_heads, tptr ? T : t + 1, t, s, tptr, 1,
                                m->rt_cur >= 0 ? m->rowtab[m->rt_cur] : nullptr, T, (int64_t)T * 64));      // cache rows: [head][T][64]
        KZV_TRY(gemm_ln(m, a.ctx, Hd, m->w_do[i], B, Hd, P + d.ob, x, false, a.s1, nullptr, P + d.ln1w, P + d.ln1b, a.x1h, a.x1, a.st1, s));
        KZV_TRY(gemm(a.x1h, Hd, m->w_dcq[i], false, B, Hd, Hd, Hd, P + d.cqb, a.cq, Hd, KZV_EPI_BF16, s));
        {
            const int64_t img = (int64_t)m->npa * Hd, plane2 = (int64_t)m->Be * img;      // [layer][K|V][image][head][key][64]
            KZV_TRY(kzv_attn_decode(a.cq, Hd, nullptr, nullptr, 0, m->ckv_dec + (int64_t)(2 * i) * plane2, m->ckv_dec + (int64_t)(2 * i + 1) * plane2,
                                    img, 64, nullptr, 0, a.cctx, Hd, B, c.dec_heads, m->npa, -1, s, nullptr, B / m->Be, nullptr, 0, (int64_t)m->npa * 64));
        }
        KZV_TRY(gemm_ln(m, a.cctx, Hd, m->w_dco[i], B, Hd, P + d.cob, a.x1, false, a.s2, nullptr, P + d.ln2w, P + d.ln2b, a.x2h, a.x2, a.st2, s));
        KZV_TRY(gemm(a.x2h, Hd, m->w_dfc1[i], false, B, Fd, Hd, Fd, P + d.fc1b, a.act, Fd, KZV_EPI_GELU, s, nullptr, a.pre, Fd));
        KZV_TRY(gemm_ln(m, a.act, Fd, m->w_dfc2[i], B, Fd, P + d.fc2b, a.x2, false, a.s3, nullptr, P + d.ln3w, P + d.ln3b, a.x3h, a.x3, a.st3, s));
        x = a.x3; xh = a.x3h;
    }
    KZV_TRY(gemm_ln(m, xh, Hd, m->w_hd, B, Hd, P + m->hd_b, nullptr, true, m->hd_gelu, m->hd_pre, P + m->hln_w, P + m->hln_b, m->hd_ln, nullptr, m->hd_st, s));
    if (m->V % 4 == 0)        // straight into the caller's [B, V] buffer (the padded scratch + copy costs a launch per token)
        return gemm(m->hd_ln, Hd, m->w_word, false, B, m->V, Hd, m->V, P + m->hbias, d_logits, m->V, KZV_EPI_F32, s);
    KZV_TRY(gemm(m->hd_ln, Hd, m->w_word, false, B, m->Vp, Hd, m->V, P + m->hbias, m->logits, m->Vp, KZV_EPI_F32, s));
    KZV_TRY(kzv_copy_logits(m->logits, m->Vp, d_logits, B, m->V, s));
    return KZV_OK;
}

static int decode_step_check(kzv_model* m, const void* a, const void* b, const void* c, const void* d, const char* who) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "%s: model not bound", who);
    if (!m->have_enc) return kzv_fail(KZV_E_STATE, "%s: call kzv_forward_loss / kzv_encode_images on the images first", who);
    if (!a || !b || !c || !d) return kzv_fail(KZV_E_ARG, "%s: null operand", who);
    if (m->Be < 1 || m->B % m->Be) return kzv_fail(KZV_E_STATE, "%s: %d decoder rows are not a multiple of the %d encoded images", who, m->B, m->Be);
    return KZV_OK;
}

extern "C" int kzv_decode_step(kzv_model* m, const int64_t* d_tokens, const int* d_posids, int t, const unsigned char* d_valid,
                               int64_t ld_valid, float* d_logits, void* stream) {
    KZV_TRY(decode_step_check(m, d_tokens, d_posids, d_valid, d_logits, "decode_step"));
    if (t < 0 || t >= m->T) return kzv_fail(KZV_E_ARG, "decode_step: step outside 0..T-1");
    KZV_TRY(ensure_kv_cache(m));
    if (t == 0) m->rt_cur = -1;                 // a new generation: no beam has been re-parented yet
    KZV_TRY(ensure_cross_layout(m, (hipStream_t)stream));
    KZV_TRY(ensure_dec_pack(m, (hipStream_t)stream));
    m->train = false; m->have_fwd = false;      // decoder activations are overwritten: no backward after this
    return decode_step_body(m, d_tokens, d_posids, t, nullptr, d_valid, ld_valid, d_logits, (hipStream_t)stream);
}

extern "C" int kzv_decode_begin(kzv_model* m, void* stream) {
    if (!m || !m->bound) return kzv_fail(KZV_E_STATE, "decode_begin: model not bound");
    KZV_TRY(ensure_kv_cache(m));
    if (m->have_enc) KZV_TRY(ensure_cross_layout(m, (hipStream_t)stream));
    KZV_TRY(ensure_dec_pack(m, (hipStream_t)stream));
    m->rt_cur = -1;                              // a new generation: every sequence reads its own cache row
    if (hipMemsetAsync(m->d_t, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) return kzv_fail(KZV_E_HIP, "decode_begin: memset");
    return KZV_OK;
}

extern "C" int kzv_decode_step_graph(kzv_model* m, const int64_t* d_tokens, const int* d_posids, const unsigned char* d_valid, int64_t ld_valid,
                                     float* d_logits, void* stream) {
    KZV_TRY(decode_step_check(m, d_tokens, d_posids, d_valid, d_logits, "decode_step_graph"));
    if (!stream) return kzv_fail(KZV_E_ARG, "decode_step_graph: needs a non-default stream (stream capture)");
    KZV_TRY(ensure_kv_cache(m));
    KZV_TRY(ensure_cross_layout(m, (hipStream_t)stream));          // before any capture: a plain launch, once per generation
    KZV_TRY(ensure_dec_pack(m, (hipStream_t)stream));
    m->train = false; m->have_fwd = false;
    hipStream_t s = (hipStream_t)stream;
    const int g = m->rt_cur + 1;
    const void* key[6] = {d_tokens, d_posids, d_valid, d_logits, m->kvc[0], (const void*)((intptr_t)m->ckv_dec ^ (intptr_t)(m->npa * 4096 + m->Be) ^ ((intptr_t)decode_one_launch_mode() << 40))};
    bool same = m->dgraph[g] != nullptr && m->dg_ld[g] == ld_valid;
    for (int i = 0; i < 6 && same; ++i) same = m->dg_key[g][i] == key[i];
    if (!same) {                               // (re)capture: the step with its index read from m->d_t, then t += 1
        if (m->dgraph[g]) { (void)hipGraphExecDestroy(m->dgraph[g]); m->dgraph[g] = nullptr; }
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return kzv_fail(KZV_E_HIP, "decode_step_graph: begin capture");
        int rc = decode_step_body(m, d_tokens, d_posids, 0, m->d_t, d_valid, ld_valid, d_logits, s);
        if (rc == KZV_OK) rc = kzv_step_inc(m->d_t, s);
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (rc != KZV_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) return kzv_fail(KZV_E_HIP, "decode_step_graph: end capture (%s)", hipGetErrorString(e));
        const hipError_t ei = hipGraphInstantiate(&m->dgraph[g], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { m->dgraph[g] = nullptr; return kzv_fail(KZV_E_HIP, "decode_step_graph: instantiate (%s)", hipGetErrorString(ei)); }
        for (int i = 0; i < 6; ++i) m->dg_key[g][i] = key[i];
        m->dg_ld[g] = ld_valid;
    }
    if (hipGraphLaunch(m->dgraph[g], s) != hipSuccess) return kzv_fail(KZV_E_HIP, "decode_step_graph: launch");
    return KZV_OK;
}

extern "C" int kzv_decode_reorder(kzv_model* m, const int64_t* d_rows, int len, void* stream) {
    if (!m || !m->bound || !m->kvc[0]) return kzv_fail(KZV_E_STATE, "decode_reorder: no KV cache (call kzv_decode_step first)");
    if (!d_rows || len < 1 || len > m->T) return kzv_fail(KZV_E_ARG, "decode_reorder: rows / length");
    // no cache row moves: the next step's attention reads key j of sequence b from the row of the ancestor that wrote it
    const int nxt = m->rt_cur < 0 ? 0 : m->rt_cur ^ 1;
    KZV_TRY(kzv_kv_rows(m->rt_cur < 0 ? nullptr : m->rowtab[m->rt_cur], m->rowtab[nxt], d_rows, m->B, m->T, len, (hipStream_t)stream));
    m->rt_cur = nxt;
    return KZV_OK;
}

extern "C" int kzv_zero_grads(kzv_model* m, void* stream) {
    if (!m || !m->bound || !m->G) return kzv_fail(KZV_E_STATE, "zero_grads: no gradient buffer bound");
    if (hipMemsetAsync(m->G, 0, m->total * sizeof(float), (hipStream_t)stream) != hipSuccess) return kzv_fail(KZV_E_HIP, "zero_grads");
    return KZV_OK;
}

extern "C" int kzv_backward_segments(const kzv_model* m) { return m ? m->Le + 2 : 0; }

extern "C" int kzv_backward_segment_range(const kzv_model* m, int seg, int64_t* lo, int64_t* hi) {
    if (!m || seg < 0 || seg >= m->Le + 2) return kzv_fail(KZV_E_ARG, "segment_range: bad segment");
    int64_t a, b;
    if (seg == 0) { a = m->lnf_w; b = m->total; }                                   // decoder + proj + final LN
    else if (seg <= m->Le) {                                                         // encoder layer Le - seg
        const int i = m->Le - seg;
        a = m->ep[i].ln1w; b = i + 1 < m->Le ? m->ep[i + 1].ln1w : m->lnf_w;
    } else { a = 0; b = m->Le ? m->ep[0].ln1w : m->lnf_w; }                            // patch / cls / pos
    if (lo) *lo = a;
    if (hi) *hi = b;
    return KZV_OK;
}

extern "C" int kzv_backward_segment(kzv_model* m, int seg, void* stream) {
    if (!m || !m->bound || !m->G) return kzv_fail(KZV_E_STATE, "backward: no gradient buffer bound");
    if (!m->have_fwd) return kzv_fail(KZV_E_STATE, "backward: call kzv_forward_loss(train=1) first");
    if (seg < 0 || seg >= m->Le + 2) return kzv_fail(KZV_E_ARG, "backward: bad segment");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    {
        KzvLnDeferScope ln_folds(s);             // the segment's LayerNorm gamma / beta folds: one launch when the scope closes (under
        if (seg == 0) rc = backward_decoder(m, s);              // kzv_backward: when ITS scope closes, once per backward pass)
        else if (seg <= m->Le) rc = backward_enc_layer(m, m->Le - seg, s);
        else rc = backward_embed(m, s);
    }
    if (rc != KZV_OK) return rc;
    // contract: in `stream` order, this segment's gradient range is final -> the side stream must be joined
    // (kzv_backward, which has no consumer between segments, joins once at the end instead)
    if (m->join_each_segment || seg == m->Le + 1) return join_side(m, s);
    return KZV_OK;
}

extern "C" int kzv_backward(kzv_model* m, void* stream) {
    if (!m) return kzv_fail(KZV_E_STATE, "backward: null model");
    const int n = kzv_backward_segments(m);
    m->join_each_segment = false;
    int rc = KZV_OK;
    {
        KzvLnDeferScope ln_folds((hipStream_t)stream);
        for (int sgm = 0; sgm < n && rc == KZV_OK; ++sgm) rc = kzv_backward_segment(m, sgm, stream);
    }
    m->join_each_segment = true;
    if (rc != KZV_OK) (void)join_side(m, (hipStream_t)stream);
    return rc;
}
