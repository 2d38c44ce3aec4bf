// What the per-element arithmetic of k_gdn costs IN CONTEXT on gfx950: the chain of k_gdn_body.hpp (accumulator -> output byte) over 8
// independent elements, instruction kind by instruction kind (the order hipcc emits: no instruction depends on its predecessor), in registers
// (no memory, no MFMA) — and the same chain with single kinds taken out, so that each one's marginal cost inside the mix shows
// (valu_rates.hip times them in isolation: 2.5 / 4.3 / 8.2 cycles; k_gdn's PMC row says its VALU is 97 % busy at ~ 30 % more cycles than
// those isolated rates add up to).   hipcc --offload-arch=gfx950 -O2 gdn_chain.hip -o gdn_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { SHIFTADD = 1, CVTN = 2, AND1 = 4, ROOT = 8, MUL = 16, AND2 = 32, CVTX = 64, FMA = 128, PK = 256, ALL = 511, RSQ = 512,
       MF16 = 1024,     // + the gamma product's share: one v_mfma_i32_16x16x64_i8 per element of a lane (32 per item of 16 positions x 128 channels)
       MF32 = 2048 };   // + the same MACs as v_mfma_i32_32x32x32_i8: half as many instructions (what an item of 32 positions would issue)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define EACH for (int i = 0; i < 8; i++)

template <int M>
__global__ __launch_bounds__(1024) void k_chain(unsigned long long *out, int iters, uint32_t seed)
{
    uint32_t n[8], t[8], u[8], o[8];
    for (int i = 0; i < 8; i++) { n[i] = seed + threadIdx.x * 8 + i + 1000; t[i] = 0x3f800000u + i; u[i] = 0; o[i] = 0; }
    uint32_t beta = 77, xb = seed * 0x01010101u + threadIdx.x;
    float kc = 1.0001f;
    v4i fa = {(int)seed, 1, 2, 3}, fb = {4, 5, (int)threadIdx.x, 7};
    v4i a16[8];
    v16i a32[4];
    for (int i = 0; i < 8; i++) a16[i] = v4i{0, 0, 0, 0};
    for (int i = 0; i < 4; i++) a32[i] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int mf = 0;   // one MFMA behind each group of 8 VALU instructions: 8 per pass of 8 elements (MF16) or every second group (MF32)
#define MFMA_SLOT()                                                                                                                                  \
    do {                                                                                                                                              \
        if (M & MF16) { asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(a16[mf & 7]) : "v"(fa), "v"(fb)); mf++; }                        \
        if (M & MF32) { if (!(mf & 1)) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(a32[(mf >> 1) & 3]) : "v"(fa), "v"(fb)); mf++; }  \
    } while (0)
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (M & SHIFTADD) { _Pragma("unroll") EACH asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(n[i]) : "v"(beta)); }
            MFMA_SLOT();
            if (M & CVTN) { _Pragma("unroll") EACH asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(t[i]) : "v"(n[i])); }
            MFMA_SLOT();
            if (M & AND1) { _Pragma("unroll") EACH asm volatile("v_and_b32 %0, 0xffffe000, %0" : "+v"(t[i])); }
            MFMA_SLOT();
            if (M & ROOT) {
                if (M & RSQ) { _Pragma("unroll") EACH asm volatile("v_rsq_f32 %0, %0" : "+v"(t[i])); }
                else { _Pragma("unroll") EACH asm volatile("v_sqrt_f32 %0, %0" : "+v"(t[i])); }
            }
            MFMA_SLOT();
            if (M & MUL) { _Pragma("unroll") EACH asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t[i]) : "v"(kc)); }
            MFMA_SLOT();
            if (M & AND2) { _Pragma("unroll") EACH asm volatile("v_and_b32 %0, 0xffffe000, %0" : "+v"(t[i])); }
            MFMA_SLOT();
            if (M & CVTX) { _Pragma("unroll") EACH asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(u[i]) : "v"(xb)); }
            if (M & FMA) { _Pragma("unroll") EACH asm volatile("v_fmaak_f32 %0, %0, %1, 0x43000000" : "+v"(u[i]) : "v"(t[i])); }
            MFMA_SLOT();
            if (M & PK) { _Pragma("unroll") EACH asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(o[i]) : "v"(u[i])); }
            MFMA_SLOT();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= n[i] ^ t[i] ^ u[i] ^ o[i] ^ (uint32_t)a16[i][0];
    for (int i = 0; i < 4; i++) s ^= (uint32_t)a32[i][0];
    if (s == 0x12345678u) out[1] = s;
    if ((threadIdx.x & 63) == 0) { out[2 + 2 * (threadIdx.x >> 6)] = t0; out[3 + 2 * (threadIdx.x >> 6)] = t1; }
}

int main()
{
    unsigned long long *d; CK(hipMalloc(&d, 64 * 8));
    const int iters = 1000;
    struct { const char *name; void (*fn)(unsigned long long *, int, uint32_t); } ks[] = {
        {"full chain (9 instr., v_sqrt)", k_chain<ALL>}, {"full chain (v_rsq)", k_chain<ALL | RSQ>}, {"- v_lshl_add_u32", k_chain<ALL & ~SHIFTADD>},
        {"- v_cvt_f32_u32", k_chain<ALL & ~CVTN>}, {"- both v_and_b32 (literal)", k_chain<ALL & ~AND1 & ~AND2>}, {"- v_sqrt_f32", k_chain<ALL & ~ROOT>},
        {"- v_mul_f32", k_chain<ALL & ~MUL>}, {"- v_cvt_f32_i32_sdwa", k_chain<ALL & ~CVTX>}, {"- v_fmaak_f32", k_chain<ALL & ~FMA>},
        {"- v_cvt_pk_u8_f32", k_chain<ALL & ~PK>}, {"only the 2.5-cycle kinds (3)", k_chain<AND1 | AND2 | MUL>}, {"only v_sqrt_f32", k_chain<ROOT>},
        {"only the five 4.3-cycle kinds", k_chain<SHIFTADD | CVTN | CVTX | FMA | PK>},
        {"full chain + 1 MFMA 16x16x64 / el.", k_chain<ALL | MF16>}, {"full chain + 1/2 MFMA 32x32x32 / el.", k_chain<ALL | MF32>},
        {"only the MFMAs 16x16x64", k_chain<MF16>}, {"only the MFMAs 32x32x32", k_chain<MF32>}};
    printf("%-34s %28s %28s\n", "chain", "cycles/element, 1 wave/SIMD", "4 waves/SIMD (per element)");
    for (auto &k : ks) {
        double c[2];
        int wi = 0;
        for (int threads : {256, 1024}) {
            unsigned long long h[64], span = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k.fn, dim3(1), dim3(threads), 0, nullptr, d, iters, 1u);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
                unsigned long long lo = ~0ull, hi = 0;
                for (int w = 0; w < threads / 64; w++) { if (h[2 + 2 * w] < lo) lo = h[2 + 2 * w]; if (h[3 + 2 * w] > hi) hi = h[3 + 2 * w]; }
                span = hi - lo;
            }
            c[wi++] = (double)span / (iters * 32.0 * (threads / 256));   // per element (= one run of the chain) of one wave, per SIMD
        }
        printf("%-34s %28.2f %28.2f\n", k.name, c[0], c[1]);
    }
    return 0;
}
