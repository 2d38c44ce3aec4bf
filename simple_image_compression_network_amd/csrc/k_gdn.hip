// Fixed-point GDN / IGDN activation (include/sicn_gdn.h; specification: oracle/sicn_gdn_oracle.c).  New
// functionality — the reference has no GDN (activations.hpp:127-224) — parity status "unpinned".
//
//   x_i = int8(v_i);  n_i = beta_i + sum_j gamma[i][j] x_j^2;  nq = top 11 bits of n;  r_i = trunc11(kc / sqrt(nq)) or trunc11(kc sqrt(nq));
//   y_i = clamp(rne(fma(x_i, r_i, 128)), 0, 255) - 128          (version 2 of the specification: k_gdn_body.hpp)
//
// k_gdn (C = 128 / 192): the cross-channel sum is a [C x C] x [C x positions] product on v_mfma_i32_16x16x64_i8.
// x^2 <= 16384 does not fit a byte, so it is split x^2 = 256 hi + lo (hi <= 64; lo travels as the signed byte lo - 128, beta' carries
// the 128 * row-sum back) and the product is run twice over the SAME gamma fragment: acc = gamma*hi; acc = (acc << 8) + beta';
// acc += gamma*(lo - 128).
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


// ---- self-test: the root exactly as the kernels compute it against exact integers, for every n --------------------------------
// a 2^ta <= b for a, b < 2^45 (ta any sign) without overflow
__device__ bool le_scaled(unsigned long long a, int ta, unsigned long long b)
{
    if (a == 0) return true;
    if (ta >= 0) return ta <= __clzll((long long)a) - 1 && (a << ta) <= b;
    const int s = -ta;
    return b != 0 && (s > __clzll((long long)b) - 1 || a <= (b << s));
}
// is r (a binary32 with 11 significant bits) == trunc11(B 2^-16 / sqrt(nq)) (GDN) or trunc11(B 2^-16 sqrt(nq)) (IGDN), nq = the top 11
// bits of n rounded nearest-even to 24 bits — all in integers: with r = M 2^k and nq = m 2^e,
//   GDN :  M^2 m 2^(2k+e+32) <= B^2 < (M+1)^2 m 2^(2k+e+32)          IGDN:  M^2 2^(2k+32-e) <= B^2 m < (M+1)^2 2^(2k+32-e)
__device__ bool gdn_root_is_exact(uint32_t n, int inverse, uint32_t rbits)
{
    int len = 32 - __clz((int)n);
    unsigned long long q = n;
    if (len > 24) {   // nearest-even to 24 significant bits
        const int sh = len - 24;
        const uint32_t rem = n & ((1u << sh) - 1), half = 1u << (sh - 1);
        q = n >> sh;
        q += (rem > half || (rem == half && (q & 1))) ? 1 : 0;
        q <<= sh;
        len = 64 - __clzll((long long)q);
    }
    const int e = len - 11;
    const unsigned long long m = e >= 0 ? q >> e : q << -e;
    if (rbits & 0x80001FFFu) return false;   // more than 11 significant bits, or negative
    const unsigned long long M = ((rbits & 0x7FFFFFu) | 0x800000u) >> 13;
    const int k = (int)(rbits >> 23) - 127 - 10;
    const unsigned long long B = 65536 + (inverse ? 33 : 5), B2 = B * B;
    if (inverse) {
        const int t = 2 * k + 32 - e;
        return le_scaled(M * M, t, B2 * m) && !le_scaled((M + 1) * (M + 1), t, B2 * m);
    }
    const int t = 2 * k + e + 32;
    return le_scaled(M * M * m, t, B2) && !le_scaled((M + 1) * (M + 1) * m, t, B2);
}
__global__ __launch_bounds__(256) void k_gdn_selftest(uint32_t n_begin, unsigned long long count, int inverse,
                                                       unsigned long long *__restrict__ bad)
{
    unsigned long long mine = 0;
    const float kc = 1.0f + (inverse ? 33.0f : 5.0f) / 65536.0f;   // the activation's constant at s = 0 (GDN: shift 16, IGDN: shift 8)
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256) {
        const uint32_t n = n_begin + (uint32_t)i;
        if (n == 0) continue;   // n >= 1 (beta >= 1)
        const float r = inverse ? gdn_root<true>(n, kc) : gdn_root<false>(n, kc);
        mine += gdn_root_is_exact(n, inverse, __float_as_uint(r)) ? 0 : 1;
    }
    if (mine) atomicAdd(bad, mine);
}


struct GdnMap {   // byte offset of 16-byte chunk k of position p: (p / P) * plane + (p % P) * pix + (k >> 1) * grp + (k & 1) * 16
    uint32_t P, plane, pix, grp;
};

template <int NJ, bool INVERSE>
__global__ __launch_bounds__(256, 4) void k_gdn(uint8_t *__restrict__ data, const int8_t *__restrict__ gamma_img,
                                                const uint32_t *__restrict__ beta, long long image_bytes, uint32_t n_pos,
                                                GdnMap map, float kc, int blocks_per_image)
{
    constexpr int C = 64 * NJ;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *gl = smem;                              // gamma image: [J][j][kg][rho][16 B]
    uint32_t *bl = (uint32_t *)(smem + C * C);       // beta' (sicn_gdn::d_beta_mfma) in natural channel order
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int pos = lane & 15, g = lane >> 4;
    for (int i = tid; i < C * C / 16; i += 256) ((uint4 *)gl)[i] = ((const uint4 *)gamma_img)[i];
    for (int i = tid; i < C; i += 256) bl[i] = beta[i];
    __syncthreads();

    uint8_t *img = data + (size_t)blockIdx.y * (size_t)image_bytes;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, (int)image_bytes, 0x00020000);
    // items = (block of 256 positions, quarter c): a wave's 16 positions x C channels per item.  The NEXT item's chunks are
    // requested before the current one is worked on, so the (HBM-latency) load hides behind the arithmetic of a whole item.
    // Addresses (round 5: the arithmetic per element halved, so the item's own bookkeeping began to show — 8 quarter-rate multiplies
    // and a division per item): the division by the plane size happens once per block of four items, the quarters step by 16
    // positions with one conditional wrap, and each lane's chunk offsets are constants.
    uint32_t offJ[NJ];   // chunk k = 4 J + g: channels 64 J + 16 g .. + 15
#pragma unroll
    for (int J = 0; J < NJ; J++) offJ[J] = (uint32_t)(2 * J + (g >> 1)) * map.grp + (uint32_t)(g & 1) * 16u;
    const uint32_t step = 16u * map.pix, wrap = map.plane - map.P * map.pix;   // one plane on, P positions back (mod 2^32)
    uint32_t blk = blockIdx.x, p_n = 0, pr_n = 0, base_n = 0;
    int c = 0;
    bool have = blk < (uint32_t)blocks_per_image;
    const bool tiny = map.P < 16u;   // a step of 16 positions could cross several planes: divide every time (2 x 2 PHASE images)
    auto first_of_block = [&]() {
        p_n = blk * 256u + (uint32_t)(w * 64 + 16 * c + pos);
        const uint32_t pl = p_n / map.P;
        pr_n = p_n - pl * map.P;
        base_n = pl * map.plane + pr_n * map.pix;
    };
    auto request = [&](v4i (&x)[NJ]) {
        const bool okp = p_n < n_pos;
#pragma unroll
        for (int J = 0; J < NJ; J++) x[J] = __builtin_amdgcn_raw_buffer_load_b128(rs, okp ? base_n + offJ[J] : OOB, 0, 0);
    };
    v4i xn[NJ];
    if (have) {
        first_of_block();
        request(xn);
    }
#pragma unroll 1
    while (have) {
        const uint32_t base = base_n;
        const bool okp = p_n < n_pos;
        v4i xf[NJ], y[NJ];
#pragma unroll
        for (int J = 0; J < NJ; J++) xf[J] = xn[J];
        if (++c == 4) {
            c = 0;
            blk += gridDim.x;
            have = blk < (uint32_t)blocks_per_image;
            if (have) first_of_block();
        } else if (tiny) {
            first_of_block();
        } else {
            p_n += 16u;
            pr_n += 16u;
            base_n += step;
            if (pr_n >= map.P) {   // at most once, as P >= 16
                pr_n -= map.P;
                base_n += wrap;
            }
        }
        if (have) request(xn);
        gdn_item<NJ, INVERSE>(xf, gl, bl, g, pos, kc, y);
#pragma unroll
        for (int J = 0; J < NJ; J++) __builtin_amdgcn_raw_buffer_store_b128(y[J], rs, okp ? base + offJ[J] : OOB, 0, 0);
    }
}

// Any channel count (<= 1024), NHWC, in place: a workgroup stages the lanes of `ppb` whole pixels in LDS, then every
// thread computes output lanes from the staged copy.
__global__ __launch_bounds__(256) void k_gdn_generic(uint8_t *__restrict__ data, const int8_t *__restrict__ gamma,
                                                     const uint32_t *__restrict__ beta, long long n_pos, int C, int inverse, float kc,
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
        for (int j = 0; j < C; j++) n += (uint32_t)(uint8_t)gamma[(size_t)i * C + j] * (uint32_t)((int)v[j] * (int)v[j]);
        data[p0 * C + e] = (uint8_t)(inverse ? gdn_out<true>((int)v[i], n, kc) : gdn_out<false>((int)v[i], n, kc));
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
        hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)n_images), dim3(256), lds, stream, data, g.d_gamma_mfma, g.d_beta_mfma,
                           image_bytes, (uint32_t)hw, m, g.kc, blocks);
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
                       g.inverse, g.kc, ppb);
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

