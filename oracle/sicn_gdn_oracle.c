/*
 * ORACLE — test infrastructure, NOT product code.
 *
 * Specification (CPU statement) of the fixed-point GDN / IGDN activation, version 2 (round 5), SURVEY.md §8(f) row 4.
 * PARITY UNPINNED: the reference has no GDN of any kind — activations.hpp:127-224 offers PassThrough, Threshold and
 * ChannelWise only, and the net's non-linearity is the sign-bit ReLU of conv_nonsquare_top.cpp:273-275.  This
 * activation REPLACES that ReLU in layers that ask for it; everything else of the layer is the reference's.
 *
 * Per pixel, C channels, v = the layer's 8-bit lane after the bias add (conv_nonsquare_top.cpp:272):
 *     x_i  = int8(v_i)                                                 -128 .. 127, no clamp
 *     n_i  = beta_i + sum_j gamma[i][j] * x_j^2                        beta in [1, 65535], gamma in [0, 127]; exact integer,
 *                                                                      < 2^29 for C <= 192, < 2^31 for C <= 1024
 *     nq_i = n_i rounded to 24 significant bits (nearest, ties to even), then cut to its top 11 significant bits
 *     GDN  (inverse = 0):  r_i = trunc11( 2^(16-SH) * (1 +  5 * 2^-16) / sqrt(nq_i) )
 *     IGDN (inverse = 1):  r_i = trunc11( 2^( 8-SH) * (1 + 33 * 2^-16) * sqrt(nq_i) )      1 <= SH <= 24
 *                          trunc11 = towards zero to 11 significant bits, of the exact real number
 *     u_i  = x_i * r_i + 128, rounded ONCE to IEEE binary32 (nearest-even) — a single fused multiply-add
 *     y_i  = clamp(nearest-even integer of u_i, 0, 255) - 128, stored as the byte y_i mod 256
 * With beta, gamma read as Q8 fixed point (beta = 256 <-> 1.0, gamma = 1 <-> 1/256) SH = 12 gives the textbook
 * y = x / sqrt(beta + sum gamma x^2) resp. y = x * sqrt(...), with the norm carried at 11 significant bits.
 *
 * Why this shape (version 1 asked for floor(2^16 / sqrt(n)) to 16 absolute bits and an arithmetic shift): every step is either an
 * exact integer operation, an IEEE-754 operation that is correctly rounded on any machine (u32 -> binary32 conversion, fma,
 * binary32 -> integer), or a root that is only needed to 11 bits.  The two biases 5 / 33 (in units of 2^-16 relative, i.e. 0.08 /
 * 0.52 of one step of r) are chosen so that for EVERY one of the 2 x 1024 values (exponent parity, 10 fraction bits) nq can
 * take, the exact root lies at least 7.5 binary32 ulps away from a step of trunc11: any root instruction that is accurate to 1 ulp,
 * followed by one binary32 multiply, truncates to the same r.  (tests/test_gdn.py re-derives those margins from exact integers.)
 * The next layer reads the byte as a stream lane; mod 256 the MAC does not care whether a lane is read as signed or unsigned.
 *
 * This file states the roots with 128-bit integer bisection over the 2048 classes and the output step with the C library's
 * fmaf / nearbyintf; oracle/sicn_ref.py: gdn_ref states all of it in Python integers only.  Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;

static u128 isqrt_u128(u128 v) /* max{ r : r*r <= v }, v < 2^126 */
{
    u128 lo = 0, hi = (u128)1 << 63;
    while (lo < hi) {
        const u128 mid = (lo + hi + 1) >> 1;
        if (mid * mid <= v) lo = mid; else hi = mid - 1;
    }
    return lo;
}

/* trunc11 of the root for nq = m * 2^par, m in [1024, 2048) (value 2^par * (1 + frac / 1024) * 2^10), SH = the activation's
 * shift, as a binary32 value.  V^2 = B^2 4^s / (2^32 nq) (GDN, s = 16 - SH) or nq B^2 4^s / 2^32 (IGDN, s = 8 - SH);
 * floor(V 2^T) = isqrt(floor(V^2 4^T)), T = 40; its top 11 bits are trunc11(V). */
static float root_class(int inverse, int SH, unsigned par, unsigned frac)
{
    const int T = 40;
    const u128 nq = (u128)(1024 + frac) << par;           /* scaled by 2^10: taken back out of the exponent below */
    const u128 B = 65536 + (inverse ? 33 : 5);
    const int s = (inverse ? 8 : 16) - SH;
    const int e2 = 2 * s + 2 * T - 32;                     /* >= 16 */
    u128 v2;
    if (inverse) v2 = (nq * B * B) << e2; else v2 = ((B * B) << e2) / nq;   /* nq B^2 < 2^45, e2 <= 62;  B^2 2^e2 < 2^112 */
    u128 v = isqrt_u128(v2);                               /* floor(V 2^T) with nq scaled by 2^10 */
    int L = 0;
    while ((v >> L) > 1) L++;                              /* v in [2^L, 2^(L+1)) */
    const unsigned M = (unsigned)(v >> (L - 10));          /* 11 significant bits */
    /* undo nq's 2^10: V scales by 2^-5 (GDN: 1/sqrt) or 2^+5 (IGDN) */
    return ldexpf((float)M, L - 10 - T + (inverse ? -5 : 5));
}

static float (*class_table(int inverse, int SH))[1024]
{
    static float tab[2][24][2][1024];                      /* [inverse][SH-1][par][frac], filled on demand */
    static unsigned char have[2][24];
    float (*t)[1024] = tab[inverse][SH - 1];
#ifdef _OPENMP
#pragma omp critical(sicn_gdn_tab)
#endif
    if (!have[inverse][SH - 1]) {
        for (unsigned par = 0; par < 2; par++)
            for (unsigned f = 0; f < 1024; f++) t[par][f] = root_class(inverse, SH, par, f);
        have[inverse][SH - 1] = 1;
    }
    return t;
}

/* r for one n: nq from the binary32 conversion of n (IEEE: nearest-even to 24 significant bits) cut to 11 bits, the class's root
 * scaled by the exponent */
static float root_of(uint32_t n, int inverse, float (*t)[1024])
{
    const float nf = (float)n;
    uint32_t b;
    memcpy(&b, &nf, 4);
    const int E = (int)(b >> 23) - 127;                    /* nq = 2^E (1 + frac / 1024) */
    const unsigned par = (unsigned)E & 1u, frac = (b >> 13) & 1023u;
    const int j2 = (E - (int)par) / 2;                     /* nq = 4^j2 * class value: the root scales by 2^-j2 / 2^+j2 */
    return ldexpf(t[par][frac], inverse ? j2 : -j2);
}

static uint8_t output_of(int x, float r)
{
    const float u = fmaf((float)x, r, 128.0f);
    float q = nearbyintf(u);                               /* default rounding mode: nearest-even */
    if (!(q > 0.0f)) q = 0.0f;
    if (q > 255.0f) q = 255.0f;
    return (uint8_t)((unsigned)q ^ 0x80u);
}

/* Test hooks: the two halves of the per-element arithmetic on their own, so that tests/test_gdn.py can hold this statement and the
 * integer-only one of oracle/sicn_ref.py together EXHAUSTIVELY (every class of nq at every exponent and shift; every r x every lane). */
int sicn_or_gdn_roots(const uint32_t *n, float *r, long long count, int inverse, int SH)
{
    if (SH < 1 || SH > 24 || (inverse != 0 && inverse != 1)) return -22;
    float (*t)[1024] = class_table(inverse, SH);
    for (long long i = 0; i < count; i++) {
        if (n[i] == 0) return -22;
        r[i] = root_of(n[i], inverse, t);
    }
    return 0;
}

int sicn_or_gdn_outputs(const int8_t *x, const float *r, uint8_t *y, long long count)
{
    for (long long i = 0; i < count; i++) y[i] = output_of((int)x[i], r[i]);
    return 0;
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
    float (*t)[1024] = class_table(inverse, SH);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (long long p = 0; p < npos; p++) {
        int32_t xs[1024];
        uint32_t sq[1024];
        for (int j = 0; j < C; j++) {
            const int v = (int8_t)x[p * C + j];
            xs[j] = v;
            sq[j] = (uint32_t)(v * v);
        }
        for (int i = 0; i < C; i++) {
            uint32_t n = beta[i];
            for (int j = 0; j < C; j++) n += (uint32_t)gamma[(long long)i * C + j] * sq[j];
            y[p * C + i] = output_of(xs[i], root_of(n, inverse, t));
        }
    }
    return 0;
}
