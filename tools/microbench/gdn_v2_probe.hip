// What the instructions the v2 GDN / IGDN specification is written around actually return on gfx950 (round 5):
//   1. v_rsq_f32 / v_sqrt_f32 on floats with a 12-bit mantissa fraction (both exponent parities)  -> probe.bin, and whether the
//      result's mantissa depends only on (fraction, exponent parity) across all exponents a u32 n can take;
//   2. v_cvt_pk_u8_f32: rounding and saturation;
//   3. the SDWA forms the kernel wants (byte-select + sign-extend on v_cvt_f32_i32 and v_mul_i32_i24, word destination).
// hipcc --offload-arch=gfx950 -O2 gdn_v2_probe.hip -o gdn_v2_probe && ./gdn_v2_probe out.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_roots(uint32_t *rsq, uint32_t *sq, int fbits)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;   // i = parity << fbits | fraction
    if (i >= (2u << fbits)) return;
    const uint32_t par = i >> fbits, fr = i & ((1u << fbits) - 1);
    const uint32_t bits = ((127u + par) << 23) | (fr << (23 - fbits));
    const float f = __uint_as_float(bits);
    rsq[i] = __float_as_uint(__builtin_amdgcn_rsqf(f));
    sq[i] = __float_as_uint(__builtin_amdgcn_sqrtf(f));
}

// exponent invariance: for e in [0, 32): op(2^e * 1.fr) mantissa == op(2^(e&1) * 1.fr) mantissa and the exponent moves by e>>1
__global__ void k_invariance(unsigned long long *bad, int fbits)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (2u << fbits)) return;
    const uint32_t par = i >> fbits, fr = i & ((1u << fbits) - 1);
    const uint32_t base = ((127u + par) << 23) | (fr << (23 - fbits));
    const uint32_t r0 = __float_as_uint(__builtin_amdgcn_rsqf(__uint_as_float(base)));
    const uint32_t s0 = __float_as_uint(__builtin_amdgcn_sqrtf(__uint_as_float(base)));
    unsigned long long b = 0;
    for (uint32_t k = 1; k < 16; k++) {
        const uint32_t bits = base + ((2 * k) << 23);
        const uint32_t r = __float_as_uint(__builtin_amdgcn_rsqf(__uint_as_float(bits)));
        const uint32_t s = __float_as_uint(__builtin_amdgcn_sqrtf(__uint_as_float(bits)));
        b += (r != r0 - (k << 23)) + (s != s0 + (k << 23));
    }
    if (b) atomicAdd(bad, b);
}

__global__ void k_cvt(const float *in, uint32_t *out, int n)
{
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xAABBCCDDu);
}

__global__ void k_sdwa(uint32_t v, uint32_t *out)
{
    if (threadIdx.x) return;
    float f0, f1, f2, f3;
    asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(f0) : "v"(v));
    asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f1) : "v"(v));
    asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f2) : "v"(v));
    asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(f3) : "v"(v));
    out[0] = __float_as_uint(f0); out[1] = __float_as_uint(f1); out[2] = __float_as_uint(f2); out[3] = __float_as_uint(f3);
    uint32_t s01, s23;
    asm volatile("v_mul_i32_i24_sdwa %0, sext(%1), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(s01) : "v"(v));
    asm volatile("v_mul_i32_i24_sdwa %0, sext(%1), sext(%1) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1" : "+v"(s01) : "v"(v));
    asm volatile("v_mul_i32_i24_sdwa %0, sext(%1), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(s23) : "v"(v));
    asm volatile("v_mul_i32_i24_sdwa %0, sext(%1), sext(%1) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3" : "+v"(s23) : "v"(v));
    out[4] = s01; out[5] = s23;
    out[6] = __builtin_amdgcn_perm(s23, s01, 0x06040200u);   // low bytes of the four squares
    out[7] = __builtin_amdgcn_perm(s23, s01, 0x07050301u);   // high bytes
}

int main(int argc, char **argv)
{
    const int FB = 12, NCL = 2 << FB;
    uint32_t *d_r, *d_s; unsigned long long *d_bad;
    CK(hipMalloc(&d_r, NCL * 4)); CK(hipMalloc(&d_s, NCL * 4)); CK(hipMalloc(&d_bad, 8)); CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(k_roots, dim3(NCL / 256), dim3(256), 0, nullptr, d_r, d_s, FB);
    hipLaunchKernelGGL(k_invariance, dim3(NCL / 256), dim3(256), 0, nullptr, d_bad, FB);
    std::vector<uint32_t> r(NCL), s(NCL); unsigned long long bad = 0;
    CK(hipMemcpy(r.data(), d_r, NCL * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s.data(), d_s, NCL * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    printf("exponent invariance violations (rsq + sqrt, 12-bit fractions x 2 parities x 15 even shifts): %llu\n", bad);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "wb");
        if (!f) { printf("cannot write %s\n", argv[1]); return 1; }
        fwrite(r.data(), 4, NCL, f); fwrite(s.data(), 4, NCL, f); fclose(f);
        printf("wrote %s: rsq[2][4096], sqrt[2][4096] (u32 bit patterns of the results on 2^par * 1.frac)\n", argv[1]);
    }
    const float in[] = {-1e9f, -1.0f, -0.75f, -0.5f, -0.25f, 0.0f, 0.25f, 0.49999997f, 0.5f, 0.50000006f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f, 2.5f, 3.5f,
                        127.5f, 128.5f, 254.5f, 254.99998f, 255.0f, 255.4f, 255.5f, 255.6f, 256.0f, 1e9f, 1e30f};
    const int NI = sizeof(in) / sizeof(in[0]);
    float *d_in; uint32_t *d_out;
    CK(hipMalloc(&d_in, sizeof(in))); CK(hipMalloc(&d_out, 64 * 4)); CK(hipMemcpy(d_in, in, sizeof(in), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, nullptr, d_in, d_out, NI);
    uint32_t out[64];
    CK(hipMemcpy(out, d_out, NI * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < NI; i++) printf("cvt_pk_u8_f32(%.9g, byte 1, 0xAABBCCDD) = 0x%08X -> %u\n", in[i], out[i], (out[i] >> 8) & 255);
    hipLaunchKernelGGL(k_sdwa, dim3(1), dim3(64), 0, nullptr, 0x80FF7F03u, d_out);
    CK(hipMemcpy(out, d_out, 8 * 4, hipMemcpyDeviceToHost));
    float f[4]; memcpy(f, out, 16);
    printf("sdwa cvt of bytes of 0x80FF7F03: %g %g %g %g (want 3 127 -1 -128)\n", f[0], f[1], f[2], f[3]);
    printf("sdwa squares: s01 = 0x%08X (want 0x3F010009) s23 = 0x%08X (want 0x40000001)  lo = 0x%08X (want 0x00010109) hi = 0x%08X (want 0x40003F00)\n",
           out[4], out[5], out[6], out[7]);
    return 0;
}
