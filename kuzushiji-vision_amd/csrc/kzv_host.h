// Host-side helpers shared by the launchers (error string, zero page, dropout thresholds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int kzv_fail(int code, const char* fmt, ...);          // records message, returns code
int kzv_check_launch(const char* what);
int kzv_cu_reserve();                                  // CUs left to concurrent collectives (kzv_set_cu_reserve / KZV_CU_RESERVE)                // hipGetLastError -> KZV_E_HIP
const void* kzv_zero_page();                           // 4 KiB of device zeros (allocated once per process)
void kzv_drop_params(float p, unsigned* thr16, float* inv_keep);
extern "C" uint32_t kzv_drop_key(uint64_t seed, uint32_t site);

// ---- optional per-launch HIP-event timing of the hot kernels (bench.py's roofline leg) -------------
// kind: 0 gemm_nt, 1 gemm_tn, 2 attn_fwd, 3 attn_bwd.  Off by default: zero cost when disabled.
struct KzvProfScope {
    int slot;
    hipStream_t s;
    KzvProfScope(int kind, double work, hipStream_t stream);
    ~KzvProfScope();
};
