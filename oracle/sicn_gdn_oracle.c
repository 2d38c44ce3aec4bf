/*
 * ORACLE — test infrastructure, NOT product code.
 *
 * Specification (CPU statement) of the fixed-point GDN / IGDN activation, SURVEY.md §8(f) row 4.
 * PARITY UNPINNED: the reference has no GDN of any kind — activations.hpp:127-224 offers PassThrough, Threshold and
 * ChannelWise only, and the net's non-linearity is the sign-bit ReLU of conv_nonsquare_top.cpp:273-275.  This
 * activation REPLACES that ReLU in layers that ask for it; everything else of the layer is the reference's.
 *
 * Per pixel, C channels, v = the layer's 8-bit lane after the bias add (conv_nonsquare_top.cpp:272), read as int8:
 *     x_i = max(v_i, -127)                                            (so that x^2 <= 16129 = 126*128 + 1)
 *     n_i = beta_i + sum_j gamma[i][j] * x_j^2                        beta in [1, 65535], gamma in [0, 127]
 *     GDN  (inverse = 0):  r_i = floor(2^16 / sqrt(n_i))  = max{ r : r^2 * n_i <= 2^32 }
 *     IGDN (inverse = 1):  r_i = floor(2^8  * sqrt(n_i))  = max{ r : r^2 <= n_i * 2^16 }
 *     t_i = (x_i * r_i + 2^(SH-1)) >> SH                              arithmetic shift (floor), 1 <= SH <= 24
 *     y_i = clamp(t_i, -128, 127), stored as the byte y_i mod 256
 * With beta, gamma read as Q8 fixed point (beta = 256 <-> 1.0, gamma = 1 <-> 1/256) SH = 12 gives the textbook
 * y = x / sqrt(beta + sum gamma x^2) resp. y = x * sqrt(...).  n_i < 2^16 + 192*127*16129 < 2^29, so every product
 * below fits 64 bits (r^2 * n < 2^61).  The next layer reads the byte as a stream lane; mod 256 the MAC does not
 * care whether a lane is read as signed or unsigned (ring homomorphism, as for pixels >= 128 in layer 0).
 *
 * The square roots here are computed by pure integer bisection — deliberately NOT the float-estimate + integer fix-up the
 * GPU kernel uses — so the two implementations are independent.
 */
#include <stdint.h>

static uint32_t isqrt_floor_u64(uint64_t v) /* max{ r : r*r <= v }, v < 2^62 */
{
    uint64_t lo = 0, hi = (uint64_t)1 << 31;
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (mid * mid <= v) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}

static uint32_t rsqrt16_floor(uint32_t n) /* max{ r : r*r*n <= 2^32 }, n >= 1 */
{
    uint64_t lo = 0, hi = 65536;
    while (lo < hi) {
        const uint64_t mid = (lo + hi + 1) >> 1;
        if (mid * mid * (uint64_t)n <= ((uint64_t)1 << 32)) lo = mid; else hi = mid - 1;
    }
    return (uint32_t)lo;
}

/* x: [npos][C] bytes (pre-activation lanes), y: [npos][C] bytes.  gamma: [C][C] row i = output channel. */
int sicn_or_gdn(const uint8_t *x, uint8_t *y, long long npos, int C, int inverse, int SH, const uint32_t *beta,
                const uint8_t *gamma)
{
    if (C <= 0 || C > 1024 || SH < 1 || SH > 24) return -22;
    for (int i = 0; i < C; i++) {
        if (beta[i] < 1 || beta[i] > 65535) return -22;
        for (int j = 0; j < C; j++)
            if (gamma[(long long)i * C + j] > 127) return -22;
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long long p = 0; p < npos; p++) {
        int32_t xs[1024];
        uint32_t sq[1024];
        for (int j = 0; j < C; j++) {
            int v = (int8_t)x[p * C + j];
            if (v < -127) v = -127;
            xs[j] = v;
            sq[j] = (uint32_t)(v * v);
        }
        for (int i = 0; i < C; i++) {
            uint32_t n = beta[i];
            for (int j = 0; j < C; j++) n += (uint32_t)gamma[(long long)i * C + j] * sq[j];
            const uint32_t r = inverse ? isqrt_floor_u64((uint64_t)n << 16) : rsqrt16_floor(n);
            int64_t t = ((int64_t)xs[i] * (int64_t)r + ((int64_t)1 << (SH - 1))) >> SH;   /* gcc: arithmetic shift */
            if (t < -128) t = -128;
            if (t > 127) t = 127;
            y[p * C + i] = (uint8_t)(t & 0xFF);
        }
    }
    return 0;
}
