// Issue cost of the VALU instructions the GDN kernels are made of, on gfx950: cycles per wave-instruction with 1 and with 4 waves per
// SIMD, 8 independent chains per wave (no dependency stalls).  Answers: which of them are quarter / half / full rate, and what an SDWA
// or packed form costs.   hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define KERNEL(NAME, ASM)                                                                                              \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *out, int iters, uint32_t seed)                    \
    {                                                                                                                  \
        uint32_t r[8];                                                                                                 \
        for (int i = 0; i < 8; i++) r[i] = seed + threadIdx.x * 8 + i + 0x3f800000u;                                   \
        uint32_t k = 0x3f800123u, z = 0;                                                                               \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < iters; it++) {                                                                           \
            _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                            \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(ASM : "+v"(r[i]) : "v"(k), "v"(z));        \
            }                                                                                                          \
        }                                                                                                              \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                    \
        uint32_t s = 0;                                                                                                \
        for (int i = 0; i < 8; i++) s ^= r[i];                                                                         \
        if (s == 0x12345678u) out[1] = s;                                                                              \
        if ((threadIdx.x & 63) == 0) { out[2 + 2 * (threadIdx.x >> 6)] = t0; out[3 + 2 * (threadIdx.x >> 6)] = t1; }          \
    }

KERNEL(k_and, "v_and_b32 %0, %1, %0")
KERNEL(k_mul_f32, "v_mul_f32 %0, %1, %0")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
KERNEL(k_rsq_f32, "v_rsq_f32 %0, %0")
KERNEL(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL(k_exp_f32, "v_exp_f32 %0, %0")
KERNEL(k_rsq_f16, "v_rsq_f16 %0, %0")
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL(k_cvt_f32_i32_sdwa, "v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1")
KERNEL(k_cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 8, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_pk_ashr, "v_pk_ashrrev_i16 %0, 8, %0 op_sel_hi:[0,1]")
KERNEL(k_mul_i24_sdwa, "v_mul_i32_i24_sdwa %0, sext(%0), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1")
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_xor, "v_xor_b32 %0, 0x80808080, %0")
KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %1")
KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")

int main()
{
    unsigned long long *d; CK(hipMalloc(&d, 64 * 8));
    const int iters = 2000;
    struct { const char *name; void (*fn)(unsigned long long *, int, uint32_t); } ks[] = {
        {"v_and_b32", k_and}, {"v_xor_b32 (literal)", k_xor}, {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_lshl_add_u32", k_lshl_add},
        {"v_perm_b32", k_perm}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_cvt_f32_i32_sdwa", k_cvt_f32_i32_sdwa}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8_f32},
        {"v_mul_i32_i24_sdwa", k_mul_i24_sdwa}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16}, {"v_pk_ashrrev_i16", k_pk_ashr}, {"v_pk_fma_f16", k_pk_fma_f16},
        {"v_cvt_pkrtz_f16_f32", k_cvt_pkrtz}, {"v_rsq_f32", k_rsq_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_exp_f32", k_exp_f32},
        {"v_rsq_f16", k_rsq_f16}, {"v_mul_lo_u32", k_mul_lo_u32}};
    printf("%-24s %22s %22s\n", "instruction", "cycles/instr, 1 wave/SIMD", "4 waves/SIMD (per instr)");
    for (auto &k : ks) {
        double c[2];
        int wi = 0;
        for (int threads : {256, 1024}) {   // one workgroup on one CU: 4 waves = 1 per SIMD, 16 waves = 4 per SIMD
            unsigned long long h[64], span = 0;   // the arbiter serves the oldest wave first: time the whole workgroup, not wave 0
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, nullptr, d, iters, 1u);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                unsigned long long lo = ~0ull, hi = 0;
                for (int w = 0; w < threads / 64; w++) { if (h[2 + 2 * w] < lo) lo = h[2 + 2 * w]; if (h[3 + 2 * w] > hi) hi = h[3 + 2 * w]; }
                span = hi - lo;
            }
            c[wi++] = (double)span / (iters * 32.0 * (threads / 256));   // per wave-instruction of one SIMD
        }
        printf("%-24s %22.2f %22.2f\n", k.name, c[0], c[1]);
    }
    return 0;
}
