// Fixed-point GDN / IGDN activation (include/sicn_gdn.h; specification: oracle/sicn_gdn_oracle.c).  New
// functionality — the reference has no GDN (activations.hpp:127-224) — parity status "unpinned".
//
//   x_i = max(int8(v_i), -127);  n_i = beta_i + sum_j gamma[i][j] x_j^2;  r_i = floor(2^16 / sqrt(n_i)) or floor(2^8 sqrt(n_i));
//   y_i = clamp((x_i r_i + 2^(SH-1)) >> SH, -128, 127)
//
// k_gdn (C = 128 / 192): the cross-channel sum is a [C x C] x [C x positions] product on v_mfma_i32_16x16x64_i8.
// x^2 <= 16129 does not fit a byte, so it is split x^2 = 128 hi + lo (hi <= 126, lo <= 127) and the product is run
// twice over the SAME gamma fragment: acc = gamma*hi; acc = (acc << 7) + beta; acc += gamma*lo.
// Layout trick (as in k_mfma16.hip): gamma's rows are stored permuted — LDS row 16 j + rho holds channel
// 64 (j>>2) + 16 (rho>>2) + 4 (j&3) + (rho&3) — so the accumulators a lane ends up with (column = its position, rows
// 4 g + r) are exactly the 16 consecutive channels 64 J + 16 g .. + 15 that the SAME lane loaded as its 16-byte B-operand
// chunk: x_i, n_i and the result y_i live in one lane, no cross-lane traffic, and y goes back in place with one
// 16-byte store per chunk.  A wave owns 64 positions (4 column tiles); the tensor may be in any of the three internal
// layouts (k_common.hpp), it is only ever addressed as (position, 16-byte channel chunk).
// k_gdn_generic: any channel count, one thread per output lane, NHWC only (small test shapes).
#include <vector>

#include "k_gdn_body.hpp"

#include "sicn_gdn_internal.h"

namespace sicn {


// exact references for the self-test: integer bisection, as in oracle/sicn_gdn_oracle.c
__device__ uint32_t gdn_rsqrt16_slow(uint32_t n)
{
    unsigned long long lo = 0, hi = 65536;
    while (lo < hi) {
        const unsigned long long mid = (lo + hi + 1) >> 1;
        if (mid * mid * (unsigned long long)n <= (1ull << 32)) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}
__device__ uint32_t gdn_sqrt8_slow(uint32_t n)
{
    const unsigned long long v = (unsigned long long)n << 16;
    unsigned long long lo = 0, hi = 1ull << 24;
    while (lo < hi) {
        const unsigned long long mid = (lo + hi + 1) >> 1;
        if (mid * mid <= v) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}
__global__ __launch_bounds__(256) void k_gdn_selftest(uint32_t n_begin, unsigned long long count, int inverse,
                                                       unsigned long long *__restrict__ bad)
{
    unsigned long long mine = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        const uint32_t n = n_begin + (uint32_t)i;
        if (!inverse && n == 0) continue;   // GDN: n >= 1 (beta >= 1)
        // inverse == 2: the narrow root exactly as the kernels call it (n >= 1: the NZ form; n = 0 only exists for the guarded form)
        const uint32_t fast = inverse == 2 ? (n ? gdn_sqrt8_narrow<true>(n) : gdn_sqrt8_narrow<false>(n)) : inverse ? gdn_sqrt8(n) : gdn_rsqrt16(n);
        const uint32_t slow = inverse ? gdn_sqrt8_slow(n) : gdn_rsqrt16_slow(n);
        mine += fast != slow;
    }
    if (mine) atomicAdd(bad, mine);
}


struct GdnMap {   // byte offset of 16-byte chunk k of position p: (p / P) * plane + (p % P) * pix + (k >> 1) * grp + (k & 1) * 16
    uint32_t P, plane, pix, grp;
};

template <int NJ, bool INVERSE>
__global__ __launch_bounds__(256, 4) void k_gdn(uint8_t *__restrict__ data, const int8_t *__restrict__ gamma_img,
                                                const uint32_t *__restrict__ beta, long long image_bytes, uint32_t n_pos,
                                                GdnMap map, int sh, int blocks_per_image)
{
    constexpr int C = 64 * NJ;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *gl = smem;                              // gamma image: [J][j][kg][rho][16 B]
    uint32_t *bl = (uint32_t *)(smem + C * C);       // beta in natural channel order
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int pos = lane & 15, g = lane >> 4;
    for (int i = tid; i < C * C / 16; i += 256) ((uint4 *)gl)[i] = ((const uint4 *)gamma_img)[i];
    for (int i = tid; i < C; i += 256) bl[i] = beta[i];
    __syncthreads();

    uint8_t *img = data + (size_t)blockIdx.y * (size_t)image_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)image_bytes, 0x00020000);
    // items = (block of 256 positions, quarter c): a wave's 16 positions x C channels per item.  The NEXT item's chunks are
    // requested before the current one is worked on, so the (HBM-latency) load hides behind ~10k cycles of arithmetic.
    auto locate = [&](uint32_t blk, int c, uint32_t &base, bool &okp) {
        const uint32_t p = blk * 256u + (uint32_t)(w * 64 + c * 16 + pos);
        okp = p < n_pos;
        const uint32_t pl = p / map.P, pr = p - pl * map.P;
        base = pl * map.plane + pr * map.pix;
    };
    auto chunk_off = [&](uint32_t base, bool okp, int J) -> uint32_t {   // chunk k = 4 J + g: channels 64 J + 16 g .. + 15
        const uint32_t k = (uint32_t)(4 * J + g);
        return okp ? base + (k >> 1) * map.grp + (k & 1u) * 16u : OOB;
    };
    uint32_t blk = blockIdx.x, base_n = 0;
    int c = 0;
    bool ok_n = false, have = blk < (uint32_t)blocks_per_image;
    v4i xn[NJ];
    if (have) {
        locate(blk, 0, base_n, ok_n);
#pragma unroll
        for (int J = 0; J < NJ; J++) xn[J] = __builtin_amdgcn_raw_buffer_load_b128(rs, chunk_off(base_n, ok_n, J), 0, 0);
    }
#pragma unroll 1
    while (have) {
        {
            const uint32_t base = base_n;
            const bool okp = ok_n;
            v4i xf[NJ], y[NJ];
#pragma unroll
            for (int J = 0; J < NJ; J++) xf[J] = xn[J];
            if (++c == 4) {
                c = 0;
                blk += gridDim.x;
            }
            have = blk < (uint32_t)blocks_per_image;
            if (have) {
                locate(blk, c, base_n, ok_n);
#pragma unroll
                for (int J = 0; J < NJ; J++) xn[J] = __builtin_amdgcn_raw_buffer_load_b128(rs, chunk_off(base_n, ok_n, J), 0, 0);
            }
            gdn_item<NJ, INVERSE>(xf, gl, bl, g, pos, sh, y);
#pragma unroll
            for (int J = 0; J < NJ; J++) __builtin_amdgcn_raw_buffer_store_b128(y[J], rs, chunk_off(base, okp, J), 0, 0);
        }
    }
}

// Any channel count (<= 1024), NHWC, in place: a workgroup stages the lanes of `ppb` whole pixels in LDS, then every
// thread computes output lanes from the staged copy.
__global__ __launch_bounds__(256) void k_gdn_generic(uint8_t *__restrict__ data, const int8_t *__restrict__ gamma,
                                                     const uint32_t *__restrict__ beta, long long n_pos, int C, int inverse, int sh,
                                                     int ppb)
{
    __shared__ int8_t px[1024];
    const long long p0 = (long long)blockIdx.x * ppb;
    const int n_here = (int)min((long long)ppb, n_pos - p0);
    for (int e = threadIdx.x; e < n_here * C; e += 256) px[e] = (int8_t)data[p0 * C + e];
    __syncthreads();
    for (int e = threadIdx.x; e < n_here * C; e += 256) {
        const int q = e / C, i = e - q * C;
        const int8_t *v = px + q * C;
        uint32_t n = beta[i];
        for (int j = 0; j < C; j++) {
            const int t = max((int)v[j], -127);
            n += (uint32_t)(uint8_t)gamma[(size_t)i * C + j] * (uint32_t)(t * t);
        }
        data[p0 * C + e] = (uint8_t)((inverse ? gdn_out<true>(max((int)v[i], -127), n, sh) : gdn_out<false>(max((int)v[i], -127), n, sh)) & 255);
    }
}

// gamma [C][C] (row = output channel) -> the LDS image of k_gdn
void pack_gdn_gamma(const uint8_t *gamma, int C, int8_t *dst)
{
    const int NJ = C / 64, NT = C / 16;
    for (int J = 0; J < NJ; J++)
        for (int j = 0; j < NT; j++)
            for (int kg = 0; kg < 4; kg++)
                for (int rho = 0; rho < 16; rho++) {
                    const int ch = 64 * (j >> 2) + 16 * (rho >> 2) + 4 * (j & 3) + (rho & 3);
                    int8_t *o = dst + ((((size_t)J * NT + j) * 4 + kg) * 16 + rho) * 16;
                    for (int b = 0; b < 16; b++) o[b] = (int8_t)gamma[(size_t)ch * C + 64 * J + 16 * kg + b];
                }
}

hipError_t launch_gdn_generic(const sicn_gdn &g, uint8_t *data, long long n_pos, hipStream_t stream);

// layout: LAYOUT_NHWC / GROUP / PHASE of an image of W x H positions; n_images images `image_bytes` apart.
hipError_t launch_gdn(const sicn_gdn &g, uint8_t *data, int layout, int W, int H, int n_images, hipStream_t stream)
{
    const int C = g.channels;
    const long long hw = (long long)W * H, image_bytes = hw * C;
    if (image_bytes >= (long long)OOB || n_images > 65535) return hipErrorInvalidValue;
    if (hw == 0 || n_images == 0) return hipSuccess;
    if (!g.d_gamma_mfma) {   // channel counts the MFMA kernel does not serve: NHWC only (the generic conv kernel writes NHWC)
        if (layout != LAYOUT_NHWC) return hipErrorInvalidValue;
        return launch_gdn_generic(g, data, hw * n_images, stream);
    }
    GdnMap m;
    if (layout == LAYOUT_NHWC) m = GdnMap{(uint32_t)hw, 0u, (uint32_t)C, 32u};
    else if (layout == LAYOUT_GROUP) m = GdnMap{(uint32_t)hw, 0u, 32u, (uint32_t)hw * 32u};
    else {
        if ((W & 1) || (H & 1)) return hipErrorInvalidValue;
        const uint32_t q = (uint32_t)(hw / 4);
        m = GdnMap{q, (uint32_t)(C / 32) * q * 32u, 32u, q * 32u};
    }
    const int blocks = (int)((hw + 255) / 256);
    const int gx = blocks < 2048 ? blocks : 2048;   // a workgroup re-uses its gamma image over several blocks
    const size_t lds = (size_t)C * C + (size_t)C * 4;
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)n_images), dim3(256), lds, stream, data, g.d_gamma_mfma, g.d_beta,
                           image_bytes, (uint32_t)hw, m, g.shift, blocks);
        return hipGetLastError();
    };
    if (C == 128) return g.inverse ? go(k_gdn<2, true>) : go(k_gdn<2, false>);
    return g.inverse ? go(k_gdn<3, true>) : go(k_gdn<3, false>);
}

hipError_t launch_gdn_generic(const sicn_gdn &g, uint8_t *data, long long n_pos, hipStream_t stream)
{
    if (n_pos == 0) return hipSuccess;
    if (g.channels > 1024) return hipErrorInvalidValue;
    const int ppb = g.channels <= 1024 ? (1024 / g.channels) : 1;
    const long long blocks = (n_pos + ppb - 1) / ppb;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_gdn_generic, dim3((unsigned)blocks), dim3(256), 0, stream, data, g.d_gamma, g.d_beta, n_pos, g.channels,
                       g.inverse, g.shift, ppb);
    return hipGetLastError();
}

hipError_t gdn_selftest_roots(int inverse, uint32_t n_begin, unsigned long long count, unsigned long long *mismatches)
{
    unsigned long long *d_bad = nullptr;
    hipError_t e = hipMalloc(&d_bad, 8);
    if (e != hipSuccess) return e;
    e = hipMemset(d_bad, 0, 8);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_gdn_selftest, dim3(4096), dim3(256), 0, nullptr, n_begin, count, inverse, d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(mismatches, d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    return e;
}

}  // namespace sicn

// Test hook for the root the MFMA kernels use on the IGDN side (gdn_sqrt8_narrow, n < 2^29): as sicn_gdn_selftest_roots.
extern "C" long long sicn_gdn_selftest_roots_narrow(uint32_t n_begin, unsigned long long count)
{
    if ((unsigned long long)n_begin + count > (1ull << 29)) return -22;
    unsigned long long bad = 0;
    if (sicn::gdn_selftest_roots(2, n_begin, count, &bad) != hipSuccess) return -19;
    return (long long)bad;
}
