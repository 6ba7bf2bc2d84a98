// Shared between the two gemm_tn kernels (gemm.hip: 128x128 two-stage; gemm_tn256.hip: 256x256 eight-phase).
#pragma once
#include "kzv_common.h"

struct TnParams {
    const bf16_t* P; const bf16_t* Q; float* OUT; const void* zero16;
    int64_t ldp, ldq, ldo;
    int Mtok, N, K, n_store, splits, chunk;   // chunk = tokens per split (multiple of 64)
    float* dbias;                              // optional: dbias[n] += sum_t P[t][n]  (the nn.Linear bias gradient)
};

// gemm_tn256.hip: returns 1 when it took the launch (p.splits / p.chunk are chosen inside), 0 when the shape is left
// to the 128x128 kernel.
int kzv_tn256_launch(const TnParams& p, hipStream_t s);

// While one of these is alive, kzv_tn256_launch leaves the fold of its partial tiles pending (each launch gets a workspace region of
// its own, up to 5); the outermost scope's destructor folds all of them in ONE launch on `stream` (the launches must be on it too).
struct KzvTnFoldScope { explicit KzvTnFoldScope(hipStream_t stream); ~KzvTnFoldScope(); hipStream_t s; };

// gemm_tn256.hip: a layer's input-gradient GEMM (kzv_gemm_nt arguments, one-store epilogue) and its weight-gradient GEMM (kzv_gemm_tn
// arguments) as ONE launch when both take the 256x256 kernels: every workgroup runs its gemm_nt tiles and then a token range of one
// weight-gradient tile sized so that all workgroups finish together.  1 = launched, 0 = not taken (issue the two GEMMs separately), < 0 = error.
// Same results as the separate launches: the gemm_nt part bit for bit, the weight gradient up to the bf16 rounding of its partial tiles
// (the token splits differ).  KZV_PAIR=0 turns it off.
struct kzv_gemm_nt_args; struct kzv_gemm_tn_args;
int kzv_gemm_pair_launch(const kzv_gemm_nt_args* na, int epilogue, const kzv_gemm_tn_args* ta, hipStream_t s);
