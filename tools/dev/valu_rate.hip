// dev: issue cost (cycles per wave-instruction) of the integer / transcendental VALU ops a dropout hash is built from.
//   hipcc -O3 --offload-arch=gfx950 tools/dev/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, body)                                                                         \
    __global__ void name(unsigned* out, unsigned long long* cyc, int iters) {                      \
        unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
        unsigned m = 0x7feb352du | threadIdx.x;                                                      \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int i = 0; i < iters; ++i) { REP16(body) }                                              \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                     \
    }
#define OP8(ins) asm volatile(ins " %0, %0, %8\n\t" ins " %1, %1, %8\n\t" ins " %2, %2, %8\n\t" ins " %3, %3, %8\n\t" \
                              ins " %4, %4, %8\n\t" ins " %5, %5, %8\n\t" ins " %6, %6, %8\n\t" ins " %7, %7, %8"       \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
#define OP8_1(ins) asm volatile(ins " %0, %0\n\t" ins " %1, %1\n\t" ins " %2, %2\n\t" ins " %3, %3\n\t" \
                                ins " %4, %4\n\t" ins " %5, %5\n\t" ins " %6, %6\n\t" ins " %7, %7"       \
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define OP8_3(ins) asm volatile(ins " %0, %0, %8, %1\n\t" ins " %1, %1, %8, %2\n\t" ins " %2, %2, %8, %3\n\t" ins " %3, %3, %8, %4\n\t" \
                                ins " %4, %4, %8, %5\n\t" ins " %5, %5, %8, %6\n\t" ins " %6, %6, %8, %7\n\t" ins " %7, %7, %8, %0"       \
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
KERNEL(k_mul_lo, OP8("v_mul_lo_u32"))
KERNEL(k_mul_hi, OP8("v_mul_hi_u32"))
KERNEL(k_mul24, OP8("v_mul_u32_u24"))
KERNEL(k_mulhi24, OP8("v_mul_hi_u32_u24"))
KERNEL(k_xor, OP8("v_xor_b32"))
KERNEL(k_add, OP8("v_add_u32"))
KERNEL(k_lshr, OP8("v_lshrrev_b32"))
KERNEL(k_mad24, OP8_3("v_mad_u32_u24"))
KERNEL(k_xad, OP8_3("v_xad_u32"))
KERNEL(k_alignbit, OP8_3("v_alignbit_b32"))
KERNEL(k_perm, OP8_3("v_perm_b32"))
KERNEL(k_lshladd, OP8_3("v_lshl_add_u32"))
KERNEL(k_bfe, OP8_3("v_bfe_u32"))
KERNEL(k_fmaf, OP8_3("v_fma_f32"))
KERNEL(k_mulf, OP8("v_mul_f32"))
KERNEL(k_expf, OP8_1("v_exp_f32"))
KERNEL(k_rcpf, OP8_1("v_rcp_f32"))
#define KERNEL2(name, body)                                                                         \
    __global__ void name(unsigned* out, unsigned long long* cyc, int iters) {                      \
        typedef float f2 __attribute__((ext_vector_type(2)));                                       \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 3.f, a2 = a0 * 5.f, a3 = a0 * 7.f, a4 = a0 + 11.f, a5 = a0 + 13.f, a6 = a0 + 17.f, a7 = a0 + 19.f; \
        f2 m = {1.0001f, 0.9999f};                                                                  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                        \
        for (int i = 0; i < iters; ++i) { REP16(body) }                                              \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                        \
        f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = __builtin_bit_cast(unsigned, r[0] + r[1]);      \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                     \
    }
KERNEL2(k_pkfma, OP8_3("v_pk_fma_f32"))
KERNEL2(k_pkmul, OP8("v_pk_mul_f32"))
KERNEL(k_madu16, OP8_3("v_mad_u16"))
KERNEL(k_pkmul16, OP8("v_pk_mul_lo_u16"))
KERNEL(k_pkmad16, OP8_3("v_pk_mad_u16"))
KERNEL(k_mullo16, OP8("v_mul_lo_u16"))

template <typename K> void run(const char* name, K k, int threads) {
    unsigned* out; unsigned long long* cyc; hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // threads/256 waves per SIMD, 128 wave-instructions per iteration per wave
    printf("%-14s %4d thr: %6.2f cycles per wave-instruction per SIMD\n", name, threads, (double)h / (iters * 128.0) / (threads / 256.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int threads : {256, 512, 1024}) {
        run("v_mul_lo_u32", k_mul_lo, threads); run("v_mul_hi_u32", k_mul_hi, threads); run("v_mul_u32_u24", k_mul24, threads);
        run("v_mul_hi_u24", k_mulhi24, threads); run("v_mad_u32_u24", k_mad24, threads); run("v_xor_b32", k_xor, threads);
        run("v_add_u32", k_add, threads); run("v_lshrrev_b32", k_lshr, threads); run("v_xad_u32", k_xad, threads);
        run("v_alignbit", k_alignbit, threads); run("v_perm_b32", k_perm, threads); run("v_lshl_add", k_lshladd, threads);
        run("v_mad_u16", k_madu16, threads); run("v_pk_mul_lo_u16", k_pkmul16, threads); run("v_pk_mad_u16", k_pkmad16, threads);
        run("v_mul_lo_u16", k_mullo16, threads);
        run("v_fma_f32", k_fmaf, threads); run("v_mul_f32", k_mulf, threads); run("v_exp_f32", k_expf, threads); run("v_rcp_f32", k_rcpf, threads);
        run("v_pk_fma_f32", k_pkfma, threads); run("v_pk_mul_f32", k_pkmul, threads);
    }
    return 0;
}
