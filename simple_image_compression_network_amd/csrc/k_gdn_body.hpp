// The per-element arithmetic of GDN / IGDN and the work of one wave on 16 positions x C channels, shared by k_gdn.hip (the tensor
// rewritten in place) and k_l0g.hip (layer 0 with the GDN applied before its one store).  Specification:
// oracle/sicn_gdn_oracle.c; the reference has no GDN (activations.hpp:127-224).
#pragma once
#include <type_traits>

#include "k_common.hpp"

namespace sicn {

typedef short v2s __attribute__((ext_vector_type(2)));

// Version 2 of the specification (round 5, oracle/sicn_gdn_oracle.c): written around what this machine does in one instruction.
//   nq = top 11 bits of n:        v_cvt_f32_u32 (IEEE nearest-even to 24 bits) + v_and_b32 on the float's bits
//   r  = trunc11(root(nq) * kc):  v_rsq_f32 / v_sqrt_f32 (quarter rate, 1 ulp) + v_mul_f32 + v_and_b32.  kc = 2^s (1 + b 2^-16); the
//        bias b keeps every one of the 2 x 1024 classes of nq >= 7.5 ulps away from a step of trunc11, so the 1-ulp root cannot land on
//        the wrong side (sicn_gdn_selftest_roots: every n < 2^31 against exact integers on the device)
//   y  = clamp(rne(fma(x, r, 128)), 0, 255) - 128:  v_cvt_f32_i32 (SDWA byte select) + v_fma_f32 + v_cvt_pk_u8_f32 (nearest-even,
//        saturating, writes one byte of the packed result) + one v_xor per 4 elements
// 9 instructions = 12 issue slots per element behind the accumulator (v1: 18 / 24 instructions with a 64-bit product).
constexpr uint32_t GDN_KEEP = 0xFFFFE000u;   // sign, exponent and 10 fraction bits of a binary32: 11 significant bits

template <bool INVERSE>
__device__ __forceinline__ float gdn_root(uint32_t n, float kc)
{
    const float nq = __uint_as_float(__float_as_uint((float)n) & GDN_KEEP);
    const float h = INVERSE ? __builtin_amdgcn_sqrtf(nq) : __builtin_amdgcn_rsqf(nq);
    return __uint_as_float(__float_as_uint(h * kc) & GDN_KEEP);
}

// One item: the 16 x C pre-activation bytes a wave holds in B-operand layout (lane (pos, g): chunk J = channels 64 J + 16 g .. + 15 of
// position pos) -> their GDN / IGDN outputs in the same lanes and bytes.  gl: the permuted gamma image (pack_gdn_gamma), bl: beta' =
// beta + 128 * (row sum of gamma) (the low digit of x^2 travels with an offset of -128, below), kc: sicn_gdn::kc.
template <int NJ, bool INVERSE>
__device__ __forceinline__ void gdn_item(const v4i (&xf)[NJ], const uint8_t *gl, const uint32_t *bl, int g, int pos, float kc, v4i (&y)[NJ])
{
    constexpr int C = 64 * NJ, NT = C / 16;
    // x^2 <= 16384 in base 256: hi = x^2 >> 8 <= 64, lo = x^2 & 255 as the signed byte lo - 128 (one XOR per dword); the MFMA runs twice
    // over the same gamma fragment: acc = gamma * hi; acc = (acc << 8) + beta'; acc += gamma * (lo - 128).
    v4i hf[NJ], lf[NJ];
#pragma unroll
    for (int J = 0; J < NJ; J++) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            // the four squares as packed 16-bit products of the sign-extended bytes (x^2 <= 16384 fits; the low 16 bits of the product of
            // two sign-extended 16-bit values are x^2): bytes 1, 3 by an arithmetic shift of each half, bytes 0, 2 moved up first
            const int v = xf[J][d], vs = v << 8;   // (a bit_cast straight from the vector element read element 0 for every d: hipcc 7.2)
            const v2s e13 = __builtin_bit_cast(v2s, v) >> 8;
            const v2s e02 = __builtin_bit_cast(v2s, vs) >> 8;
            const uint32_t s13 = __builtin_bit_cast(uint32_t, (v2s)(e13 * e13)), s02 = __builtin_bit_cast(uint32_t, (v2s)(e02 * e02));
            lf[J][d] = (int)(__builtin_amdgcn_perm(s13, s02, 0x06020400u) ^ 0x80808080u);   // [lo0 lo1 lo2 lo3]
            hf[J][d] = (int)__builtin_amdgcn_perm(s13, s02, 0x07030501u);                   // [hi0 hi1 hi2 hi3]
        }
    }
    // Tile j's 4 accumulators per lane are 4 consecutive channels: register r of tile j = channel
    // 64 (j>>2) + 16 g + 4 (j&3) + r = byte r of dword (j&3) of this lane's chunk J = j>>2.  They are turned into output
    // bytes right behind the NEXT tile's MFMAs (whose latency the 4 root computations cover) and never kept: the kernel
    // needs < 128 VGPRs, so four waves share a SIMD.
    auto finish = [&](const v4i &a, int j) {
        const int J = j >> 2, d = j & 3;
        const int v = xf[J][d];
        uint32_t packed = 0;
        packed = __builtin_amdgcn_cvt_pk_u8_f32(fmaf((float)(int)(int8_t)v, gdn_root<INVERSE>((uint32_t)a[0], kc), 128.0f), 0u, packed);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(fmaf((float)(int)(int8_t)(v >> 8), gdn_root<INVERSE>((uint32_t)a[1], kc), 128.0f), 1u, packed);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(fmaf((float)(int)(int8_t)(v >> 16), gdn_root<INVERSE>((uint32_t)a[2], kc), 128.0f), 2u, packed);
        packed = __builtin_amdgcn_cvt_pk_u8_f32(fmaf((float)(v >> 24), gdn_root<INVERSE>((uint32_t)a[3], kc), 128.0f), 3u, packed);
        y[J][d] = (int)(packed ^ 0x80808080u);
    };
    v4i a_prev = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NT; j++) {
        v4i gf[NJ];
#pragma unroll
        for (int J = 0; J < NJ; J++) gf[J] = *(const v4i *)(gl + (((J * NT + j) * 4 + g) * 16 + pos) * 16);
        v4i a = {0, 0, 0, 0};
#pragma unroll
        for (int J = 0; J < NJ; J++) a = __builtin_amdgcn_mfma_i32_16x16x64_i8(gf[J], hf[J], a, 0, 0, 0);
        const uint4 b4 = *(const uint4 *)(bl + 64 * (j >> 2) + 16 * g + 4 * (j & 3));
        a[0] = (a[0] << 8) + (int)b4.x;
        a[1] = (a[1] << 8) + (int)b4.y;
        a[2] = (a[2] << 8) + (int)b4.z;
        a[3] = (a[3] << 8) + (int)b4.w;
#pragma unroll
        for (int J = 0; J < NJ; J++) a = __builtin_amdgcn_mfma_i32_16x16x64_i8(gf[J], lf[J], a, 0, 0, 0);
        if (j > 0) finish(a_prev, j - 1);
        a_prev = a;
        __builtin_amdgcn_sched_barrier(0);   // keep the tiles apart: interleaving all of them costs VGPRs
    }
    finish(a_prev, NT - 1);
}

// the scalar form of one output lane (k_gdn_generic, any channel count)
template <bool INVERSE>
__device__ __forceinline__ uint32_t gdn_out(int x, uint32_t n, float kc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(fmaf((float)x, gdn_root<INVERSE>(n, kc), 128.0f), 0u, 0u) ^ 0x80u;
}

}  // namespace sicn
