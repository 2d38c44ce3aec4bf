// The per-element arithmetic of GDN / IGDN and the work of one wave on 16 positions x C channels, shared by k_gdn.hip (the tensor
// rewritten in place) and k_l0g.hip (layer 0 with the GDN applied before its one store).  Specification:
// oracle/sicn_gdn_oracle.c; the reference has no GDN (activations.hpp:127-224).
#pragma once
#include <type_traits>

#include "k_common.hpp"

namespace sicn {

// The kernel is VALU-bound on these two roots (r02: 55 instructions per element, 5 - 9 of them quarter-rate 32-bit
// multiplies), so they are written for instruction count; tests/test_gdn.py checks both against integer bisection for EVERY
// n < 2^31 on the device (sicn_gdn_selftest_roots).
//
// r = max{ r : r^2 n <= 2^32 } = floor(2^16 / sqrt(n)), 1 <= n < 2^31.  t = 2^16 / sqrt(n) in float: cvt (2^-24) and
// v_rsq_f32 (1 ulp) leave a relative error below 2^-22, i.e. < 0.012 absolute for t <= 46341 (n >= 2).  With q = floor(t~ + 0.05)
// the true floor is q or q - 1, and ONE exact test decides: r = q - 1 + [q^2 n <= 2^32].  q < 2^24, so q^2 is a full-rate
// 24-bit multiply; the 64-bit product costs the two slow ones.
__device__ __forceinline__ uint32_t gdn_rsqrt16(uint32_t n)
{
    // 1 <= n < 2^31 puts t~ + 0.05 into [1.46, 65536.06], so q needs no clamp (n = 0 is excluded by beta >= 1)
    const uint32_t q = (uint32_t)(65536.0f * __frsqrt_rn((float)n) + 0.05f);
    // q = 65536 only for n = 1 (r = 2^16 exactly): its 24-bit square wraps to 0, the product is 0 and the test passes — as it must
    const unsigned long long prod = (unsigned long long)(uint32_t)__umul24(q, q) * n;   // one v_mad_u64_u32 (__umul24 returns int)
    return q - (prod > (1ull << 32) ? 1u : 0u);
}

// r = max{ r : r^2 <= n 2^16 } = floor(2^8 sqrt(n)), n < 2^31 (r < 2^23.6).  The float estimate r0 is within 2 of it (relative
// 2^-22.6 of up to 1.2e7), so r = r0 - 2 + sum_{k=-1..2} [(r0 + k)^2 <= N], N = n 2^16.  With d = N - r0^2 the tests read
// 2 k r0 + k^2 <= d, and |d| < 2^27: d is exact in the LOW 32 bits of N and of r0^2 (one 24-bit multiply) — no 64-bit
// arithmetic, no slow multiply at all.
__device__ __forceinline__ uint32_t gdn_sqrt8(uint32_t n)
{
    uint32_t r0 = (uint32_t)(256.0f * __fsqrt_rn((float)n));
    r0 = max(r0, 2u);
    const int d = (int)((n << 16) - __umul24(r0, r0));
    const int r2 = (int)(2u * r0);
    return r0 - 2 + (1 - r2 <= d ? 1u : 0u) + (0 <= d ? 1u : 0u) + (r2 + 1 <= d ? 1u : 0u) + (2 * r2 + 4 <= d ? 1u : 0u);
}

// The same root for n < 2^29 (the MFMA kernels: C <= 192), where the estimate is within 0.89 of the true value (cvt 2^-25 +
// v_sqrt_f32 2^-23 relative, of at most 5.93e6): with q = floor(t~ + 0.95) the floor is q - 2, q - 1 or q — TWO tests instead of
// four: r = q - 2 + [(q - 1)^2 <= N] + [q^2 <= N], i.e. with d = N - q^2: [1 - 2 q <= d] + [0 <= d].
// NZ: n >= 1 is known (the kernels: beta >= 1), so q >= 256 and the guard for n = 0 goes.  The two tests are taken as sign bits:
// [d >= 0] - 1 = d >> 31 and [d + 2 q - 1 >= 0] - 1 = (d + 2 q - 1) >> 31 (arithmetic), r = q + both — 8 integer instructions.
template <bool NZ = false>
__device__ __forceinline__ uint32_t gdn_sqrt8_narrow(uint32_t n)
{
    uint32_t q = (uint32_t)(256.0f * __fsqrt_rn((float)n) + 0.95f);
    if (!NZ) q = max(q, 2u);
    const int d = (int)((n << 16) - __umul24(q, q));
    const int e = (int)(2u * q) + d - 1;
    return q + (uint32_t)(d >> 31) + (uint32_t)(e >> 31);
}

// NARROW (the MFMA kernels: C <= 192, so n < 2^29 and r < 2^23): x r is a full-rate 24-bit multiply-add
template <bool INVERSE, bool NARROW = false>
__device__ __forceinline__ int gdn_out(int x, uint32_t n, int sh)
{
    const uint32_t r = INVERSE ? (NARROW ? gdn_sqrt8_narrow<true>(n) : gdn_sqrt8(n)) : gdn_rsqrt16(n);   // NARROW = the MFMA kernels: n >= beta >= 1
    int t;
    if (NARROW)   // v_mad_i32_i24 through the builtin (left to itself hipcc folds x * r + c into a quarter-rate v_mad_u64_u32; an asm
        t = __mul24(x, (int)r) + (1 << (sh - 1));   // statement here had its output placed on an in-flight MFMA operand: isa_hazards.py)
    else
        t = x * (int)r + (1 << (sh - 1));
    t >>= sh;   // |x r| < 2^30; arithmetic shift
    return max(-128, min(127, t));
}

// Byte K of `dst` = the low byte of an ALU result, the other bytes kept (K = 0: cleared) — SDWA destination select.  The split
// x^2 = 128 hi + lo and the output bytes are packed this way: one instruction per byte instead of shift + mask + or (r03: the
// pass is VALU-bound; 29 -> 24 instructions per element).
// Byte 0 is NOT written by an asm statement: a fresh asm output may be given the register of an MFMA operand that is still in
// flight, and hipcc's hazard recogniser does not look inside asm (tools/isa_hazards.py found exactly that); the first write is
// an ordinary instruction the compiler spaces correctly, bytes 1 - 3 then go into a register that is already live.
#define SICN_SDWA_BYTE(K, OPC, TAIL)                                                                                  \
    static_assert(K >= 1 && K <= 3, "byte 0 is written by the caller with an ordinary instruction");                  \
    if constexpr (K == 1)                                                                                             \
        asm(OPC " dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE " TAIL : "+v"(dst) : "v"(a), "v"(b));                     \
    else if constexpr (K == 2)                                                                                        \
        asm(OPC " dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE " TAIL : "+v"(dst) : "v"(a), "v"(b));                     \
    else                                                                                                              \
        asm(OPC " dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE " TAIL : "+v"(dst) : "v"(a), "v"(b));
template <int K>
__device__ __forceinline__ void sdwa_put_shr(int &dst, uint32_t a /* shift */, uint32_t b /* value */)   // byte K = (b >> a) & 255
{
    SICN_SDWA_BYTE(K, "v_lshrrev_b32_sdwa %0, %1, %2", "src0_sel:DWORD src1_sel:DWORD")
}
template <int K>
__device__ __forceinline__ void sdwa_put_and(int &dst, uint32_t a /* mask */, uint32_t b /* value */)    // byte K = (a & b) & 255
{
    SICN_SDWA_BYTE(K, "v_and_b32_sdwa %0, %1, %2", "src0_sel:DWORD src1_sel:DWORD")
}
template <int K>
__device__ __forceinline__ void sdwa_put_or(int &dst, uint32_t a /* 0 */, uint32_t b /* value */)        // byte K = low byte of b
{
    SICN_SDWA_BYTE(K, "v_or_b32_sdwa %0, %1, %2", "src0_sel:DWORD src1_sel:BYTE_0")
}
#undef SICN_SDWA_BYTE

// One item: the 16 x C pre-activation bytes a wave holds in B-operand layout (lane (pos, g): chunk J = channels 64 J + 16 g .. + 15 of
// position pos) -> their GDN / IGDN outputs in the same lanes and bytes.  gl: the permuted gamma image (pack_gdn_gamma), bl: beta.
template <int NJ, bool INVERSE>
__device__ __forceinline__ void gdn_item(const v4i (&xf)[NJ], const uint8_t *gl, const uint32_t *bl, int g, int pos, int sh, v4i (&y)[NJ])
{
    constexpr int C = 64 * NJ, NT = C / 16;
    v4i hf[NJ], lf[NJ];
    uint32_t c7 = 7u, c127 = 127u, c0 = 0u;
    asm volatile("" : "+v"(c7), "+v"(c127), "+v"(c0));   // SDWA takes no literals: the three constants live in registers
#pragma unroll
    for (int J = 0; J < NJ; J++) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int h, l;
            auto one = [&](auto kc) {
                constexpr int b = decltype(kc)::value;
                int v = (int)(int8_t)((uint32_t)xf[J][d] >> (8 * b));
                v = max(v, -127);
                const uint32_t sq = (uint32_t)(v * v);   // <= 16129: hi = sq >> 7 <= 126 and lo = sq & 127 are bytes
                if constexpr (b == 0) {
                    h = (int)(sq >> 7);
                    l = (int)(sq & 127u);
                } else {
                    sdwa_put_shr<b>(h, c7, sq);
                    sdwa_put_and<b>(l, c127, sq);
                }
            };
            one(std::integral_constant<int, 0>{});
            one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{});
            one(std::integral_constant<int, 3>{});
            hf[J][d] = h;
            lf[J][d] = l;
        }
    }
    // Tile j's 4 accumulators per lane are 4 consecutive channels: register r of tile j = channel
    // 64 (j>>2) + 16 g + 4 (j&3) + r = byte r of dword (j&3) of this lane's chunk J = j>>2.  They are turned into output
    // bytes right behind the NEXT tile's MFMAs (whose latency the 4 root computations cover) and never kept: the kernel
    // needs < 128 VGPRs, so four waves share a SIMD — what a VALU-bound loop with MFMA -> VALU wait states wants.
    auto finish = [&](const v4i &a, int j) {
        const int J = j >> 2, d = j & 3;
        int packed;
        auto one = [&](auto kc) {
            constexpr int r = decltype(kc)::value;
            int x = (int)(int8_t)((uint32_t)xf[J][d] >> (8 * r));
            x = max(x, -127);
            const int t = gdn_out<INVERSE, true>(x, (uint32_t)a[r], sh);
            if constexpr (r == 0)
                packed = t & 255;
            else
                sdwa_put_or<r>(packed, c0, (uint32_t)t);
        };
        one(std::integral_constant<int, 0>{});
        one(std::integral_constant<int, 1>{});
        one(std::integral_constant<int, 2>{});
        one(std::integral_constant<int, 3>{});
        y[J][d] = packed;
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
        a[0] = (a[0] << 7) + (int)b4.x;
        a[1] = (a[1] << 7) + (int)b4.y;
        a[2] = (a[2] << 7) + (int)b4.z;
        a[3] = (a[3] << 7) + (int)b4.w;
#pragma unroll
        for (int J = 0; J < NJ; J++) a = __builtin_amdgcn_mfma_i32_16x16x64_i8(gf[J], lf[J], a, 0, 0, 0);
        if (j > 0) finish(a_prev, j - 1);
        a_prev = a;
        __builtin_amdgcn_sched_barrier(0);   // keep the tiles apart: interleaving all of them costs 218 VGPRs
    }
    finish(a_prev, NT - 1);
}

}  // namespace sicn
