/*
 * ORACLE — test infrastructure, NOT product code.
 *
 * CPU statement of container mode 4, "rANS-WC": the wavefront rANS of mode 3 (oracle/sicn_codec_oracle.c) with a
 * CONDITIONAL entropy model — the hyperprior / context-model coder of SURVEY.md §8(f) row 4 and BASELINE.json
 * configs[4].  THE REFERENCE HAS NOTHING OF THIS KIND (no coder, no hyperprior, no context model: SURVEY.md §0), so
 * parity is UNPINNED: the specification is this project's own; the tests show decode(encode(y)) == y and GPU == this
 * file byte for byte.
 *
 * Model.  Besides the latent y [lat_h][lat_w][lat_c] (symbols < 128) both sides hold a SCALE MAP s of the same shape
 * (values < 128): the output of the hyper-synthesis stack run on the decoded hyper-latent.  Every symbol is coded with
 * one of 16 static 12-bit frequency tables (measured per class on this latent and carried in the container); the class
 * of element (py, px, ch) is
 *     anchor     ((px + py) even):  k = s >> 3
 *     non-anchor ((px + py) odd):   m = max of y over the in-range 4-neighbours (py-1,px) (py+1,px) (py,px-1) (py,px+1),
 *                                       same channel — all of them anchors;   k = min(15, ((s >> 3) + (m >> 3) + 1) >> 1)
 * i.e. a checkerboard context model: the decoder decodes all anchors first (their classes need s only), then all
 * non-anchors (their classes need s and the decoded anchors).  Two passes, each fully parallel — the GPU-friendly
 * stand-in for a serial autoregressive context model.
 *
 * Symbol order.  Two symbol sets, anchors then non-anchors.  Inside a set: pixels in raster order, channels fastest.
 * With a = ceil(W/2), b = floor(W/2): rows 2r and 2r+1 together hold W pixels of either set;
 *     anchor pixel j:      r = j / W, t = j % W;  t < a ? (y = 2r, x = 2t) : (y = 2r+1, x = 2(t-a)+1)
 *     non-anchor pixel j:  r = j / W, t = j % W;  t < b ? (y = 2r, x = 2t+1) : (y = 2r+1, x = 2(t-b))
 *     anchors: (H/2) W + (H odd ? a : 0) pixels; non-anchors: (H/2) W + (H odd ? b : 0).
 * Each set is cut into streams of 16384 symbols coded exactly like mode 3 (64 interleaved states, 16-bit words, lane l owns
 * symbols 256 q + 4 l + 0..3) except that freq / cum come from the symbol's class table.
 *
 * Container: the 48-byte header of mode 3 with mode = 4 and n_streams = streams(anchors) + streams(non-anchors);
 * then 16 tables of 128 x u16 (class 0 first; a class without symbols is all zero; every other sums to 4096, built from
 * the class histogram by the same normalisation as mode 3); then u32 stream_bytes[n_streams]; then the streams.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SICL_HEADER 48
#define WSS 16384u
#define LANES 64u
#define RANSW_L (1u << 16)
#define PROB_BITS 12
#define NCLS 16

int sicl_or_normalize(const uint32_t h[128], uint32_t n, uint16_t f[128]);   /* sicn_codec_oracle.c */
uint32_t sicl_or_adler32(const uint8_t *d, size_t n);

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static uint32_t get32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t get16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }

static uint32_t set_pixels(int set, uint32_t W, uint32_t H)
{
    const uint32_t a = (W + 1) / 2, b = W / 2;
    return (H / 2) * W + ((H & 1) ? (set == 0 ? a : b) : 0);
}

static void set_pixel(int set, uint32_t j, uint32_t W, uint32_t *y, uint32_t *x)
{
    const uint32_t a = (W + 1) / 2, b = W / 2, r = j / W, t = j % W;
    if (set == 0) {
        if (t < a) { *y = 2 * r; *x = 2 * t; } else { *y = 2 * r + 1; *x = 2 * (t - a) + 1; }
    } else {
        if (t < b) { *y = 2 * r; *x = 2 * t + 1; } else { *y = 2 * r + 1; *x = 2 * (t - b); }
    }
}

static uint32_t elem_class(int set, const uint8_t *lat, const uint8_t *s, uint32_t W, uint32_t H, uint32_t C, uint32_t y,
                           uint32_t x, uint32_t ch)
{
    const uint32_t k0 = s[((size_t)y * W + x) * C + ch] >> 3;
    if (set == 0) return k0 > 15 ? 15 : k0;
    uint32_t m = 0;
#define NB(yy, xx) do { const uint32_t v = lat[((size_t)(yy) * W + (xx)) * C + ch]; if (v > m) m = v; } while (0)
    if (y > 0) NB(y - 1, x);
    if (y + 1 < H) NB(y + 1, x);
    if (x > 0) NB(y, x - 1);
    if (x + 1 < W) NB(y, x + 1);
#undef NB
    const uint32_t k = (k0 + (m >> 3) + 1) >> 1;
    return k > 15 ? 15 : k;
}

uint32_t sicl_or_ctx_streams(uint32_t W, uint32_t H, uint32_t C)
{
    const uint64_t na = (uint64_t)set_pixels(0, W, H) * C, nn = (uint64_t)set_pixels(1, W, H) * C;
    return (uint32_t)((na + WSS - 1) / WSS + (nn + WSS - 1) / WSS);
}

size_t sicl_or_ctx_max_bytes(uint32_t W, uint32_t H, uint32_t C)
{
    const uint64_t n = (uint64_t)W * H * C;
    const uint32_t ns = sicl_or_ctx_streams(W, H, C);
    return SICL_HEADER + NCLS * 256 + 4 * (size_t)ns + 2 * (size_t)n + 256 * (size_t)ns;
}

static uint32_t ransw_index(uint32_t i, uint32_t l) { return (i >> 2) * 256u + l * 4u + (i & 3u); }

static uint32_t encode_stream(const uint8_t *sym, const uint8_t *cls, uint32_t n, uint16_t (*freq)[128], uint16_t (*cum)[129],
                              uint8_t *buf, uint32_t cap)
{
    uint32_t x[LANES], pos = cap;
    for (uint32_t l = 0; l < LANES; l++) x[l] = RANSW_L;
    const uint32_t steps = (n + 255u) / 256u * 4u;
    for (uint32_t i = steps; i-- > 0;) {
        for (uint32_t l = LANES; l-- > 0;) {
            const uint32_t j = ransw_index(i, l);
            if (j >= n) continue;
            const uint32_t f = freq[cls[j]][sym[j]];
            if ((uint64_t)x[l] >= ((uint64_t)f << 20)) {
                buf[--pos] = (uint8_t)(x[l] >> 8);
                buf[--pos] = (uint8_t)(x[l]);
                x[l] >>= 16;
            }
        }
        for (uint32_t l = 0; l < LANES; l++) {
            const uint32_t j = ransw_index(i, l);
            if (j >= n) continue;
            const uint32_t f = freq[cls[j]][sym[j]], c = cum[cls[j]][sym[j]];
            x[l] = ((x[l] / f) << PROB_BITS) + (x[l] % f) + c;
        }
    }
    for (uint32_t l = LANES; l-- > 0;) {
        pos -= 4;
        put32(buf + pos, x[l]);
    }
    return cap - pos;
}

/* latent, scale: [H][W][C].  Returns container bytes or a negative code. */
long long sicl_or_ctx_encode(const uint8_t *latent, const uint8_t *scale, uint32_t W, uint32_t H, uint32_t C, uint32_t img_w,
                             uint32_t img_h, uint8_t *out, size_t cap)
{
    const uint64_t n64 = (uint64_t)W * H * C;
    if (n64 > 0x7f000000u) return -22;
    const uint32_t n = (uint32_t)n64;
    if (cap < sicl_or_ctx_max_bytes(W, H, C)) return -28;
    for (uint32_t i = 0; i < n; i++)
        if (latent[i] > 127 || scale[i] > 127) return -22;
    uint32_t npx[2] = {set_pixels(0, W, H), set_pixels(1, W, H)};
    uint8_t *sym = (uint8_t *)malloc((size_t)n + 1), *cls = (uint8_t *)malloc((size_t)n + 1);
    uint8_t *buf = (uint8_t *)malloc(2 * WSS + 256);
    if (!sym || !cls || !buf) { free(sym); free(cls); free(buf); return -12; }
    /* gather both sets (anchors first) and their classes */
    uint32_t hist[NCLS][128];
    memset(hist, 0, sizeof hist);
    size_t k = 0;
    for (int set = 0; set < 2; set++)
        for (uint32_t j = 0; j < npx[set]; j++) {
            uint32_t y, x;
            set_pixel(set, j, W, &y, &x);
            for (uint32_t ch = 0; ch < C; ch++, k++) {
                sym[k] = latent[((size_t)y * W + x) * C + ch];
                cls[k] = (uint8_t)elem_class(set, latent, scale, W, H, C, y, x, ch);
                hist[cls[k]][sym[k]]++;
            }
        }
    uint16_t freq[NCLS][128], cum[NCLS][129];
    for (int c = 0; c < NCLS; c++) {
        uint32_t tot = 0;
        for (int s = 0; s < 128; s++) tot += hist[c][s];
        memset(freq[c], 0, sizeof freq[c]);
        if (tot && sicl_or_normalize(hist[c], tot, freq[c])) { free(sym); free(cls); free(buf); return -22; }
        cum[c][0] = 0;
        for (int s = 0; s < 128; s++) cum[c][s + 1] = (uint16_t)(cum[c][s] + freq[c][s]);
    }
    const uint32_t nsym[2] = {npx[0] * C, npx[1] * C};
    const uint32_t nst[2] = {(nsym[0] + WSS - 1) / WSS, (nsym[1] + WSS - 1) / WSS};
    const uint32_t ns = nst[0] + nst[1];
    memset(out, 0, SICL_HEADER);
    memcpy(out, "SICL", 4);
    put16(out + 4, 1);
    put16(out + 6, 4);
    put32(out + 8, img_w);
    put32(out + 12, img_h);
    put32(out + 16, W);
    put32(out + 20, H);
    put32(out + 24, C);
    put32(out + 28, n);
    put32(out + 32, ns);
    put32(out + 36, WSS);
    put32(out + 44, sicl_or_adler32(latent, n));
    size_t pos = SICL_HEADER;
    for (int c = 0; c < NCLS; c++)
        for (int s = 0; s < 128; s++, pos += 2) put16(out + pos, freq[c][s]);
    uint8_t *lens = out + pos;
    pos += 4 * (size_t)ns;
    const size_t payload0 = pos;
    uint32_t st = 0;
    size_t base = 0;
    for (int set = 0; set < 2; set++) {
        for (uint32_t q = 0; q < nst[set]; q++, st++) {
            const uint32_t begin = q * WSS, cnt = nsym[set] - begin < WSS ? nsym[set] - begin : WSS;
            const uint32_t len = encode_stream(sym + base + begin, cls + base + begin, cnt, freq, cum, buf, 2 * WSS + 256);
            put32(lens + 4 * st, len);
            memcpy(out + pos, buf + 2 * WSS + 256 - len, len);
            pos += len;
        }
        base += nsym[set];
    }
    put32(out + 40, (uint32_t)(pos - payload0));
    free(sym); free(cls); free(buf);
    return (long long)pos;
}

/* Returns symbols decoded or a negative code (-22 malformed, -28 latent_cap too small, -74 checksum mismatch).
 * The latent shape comes from the header; `scale` must have that shape. */
long long sicl_or_ctx_decode(const uint8_t *in, size_t bytes, const uint8_t *scale, uint8_t *latent, size_t latent_cap,
                             uint32_t info[8])
{
    if (bytes < SICL_HEADER || memcmp(in, "SICL", 4) || get16(in + 4) != 1 || get16(in + 6) != 4) return -22;
    const uint32_t W = get32(in + 16), H = get32(in + 20), C = get32(in + 24), n = get32(in + 28), ns = get32(in + 32);
    const uint32_t payload = get32(in + 40);
    if ((uint64_t)W * H * C != n || n > 0x7f000000u || get32(in + 36) != WSS || ns != sicl_or_ctx_streams(W, H, C)) return -22;
    if (info) {
        info[0] = 4; info[1] = get32(in + 8); info[2] = get32(in + 12); info[3] = W; info[4] = H; info[5] = C; info[6] = n;
        info[7] = payload;
    }
    if (latent_cap < n) return -28;
    size_t pos = SICL_HEADER;
    if (bytes < pos + NCLS * 256 + 4 * (size_t)ns) return -22;
    uint16_t freq[NCLS][128], cum[NCLS][129];
    for (int c = 0; c < NCLS; c++) {
        uint32_t sum = 0;
        for (int s = 0; s < 128; s++, pos += 2) { freq[c][s] = (uint16_t)get16(in + pos); sum += freq[c][s]; }
        if (sum != 0 && sum != 4096) return -22;
        cum[c][0] = 0;
        for (int s = 0; s < 128; s++) cum[c][s + 1] = (uint16_t)(cum[c][s] + freq[c][s]);
    }
    const uint8_t *lens = in + pos;
    pos += 4 * (size_t)ns;
    uint64_t total = 0;
    for (uint32_t st = 0; st < ns; st++) total += get32(lens + 4 * st);
    if (total != payload || bytes < pos + payload) return -22;
    memset(latent, 0, n);
    uint32_t st = 0;
    for (int set = 0; set < 2; set++) {
        const uint32_t npx = set_pixels(set, W, H), nsym = npx * C, nst = (nsym + WSS - 1) / WSS;
        for (uint32_t q = 0; q < nst; q++, st++) {
            const uint32_t len = get32(lens + 4 * st), begin = q * WSS, cnt = nsym - begin < WSS ? nsym - begin : WSS;
            if (len < 4 * LANES || (len & 1)) return -22;
            const uint8_t *p = in + pos, *end = p + len;
            uint32_t x[LANES];
            for (uint32_t l = 0; l < LANES; l++, p += 4) x[l] = get32(p);
            const uint32_t steps = (cnt + 255u) / 256u * 4u;
            for (uint32_t i = 0; i < steps; i++) {
                for (uint32_t l = 0; l < LANES; l++) {
                    const uint32_t j = ransw_index(i, l);
                    if (j >= cnt) continue;
                    const uint32_t e = begin + j, px = e / C, ch = e % C;
                    uint32_t y, xx;
                    set_pixel(set, px, W, &y, &xx);
                    const uint32_t k = elem_class(set, latent, scale, W, H, C, y, xx, ch);
                    const uint32_t v = x[l] & 4095u;
                    if (cum[k][128] != 4096) return -22;              /* a symbol of a class the container has no table for */
                    uint32_t s = 0;
                    while (cum[k][s + 1] <= v) s++;
                    latent[((size_t)y * W + xx) * C + ch] = (uint8_t)s;
                    x[l] = freq[k][s] * (x[l] >> PROB_BITS) + v - cum[k][s];
                }
                for (uint32_t l = 0; l < LANES; l++) {
                    if (ransw_index(i, l) >= cnt || x[l] >= RANSW_L) continue;
                    if (p + 2 > end) return -22;
                    x[l] = (x[l] << 16) | get16(p);
                    p += 2;
                }
            }
            for (uint32_t l = 0; l < LANES; l++)
                if (x[l] != RANSW_L) return -22;
            if (p != end) return -22;
            pos += len;
        }
    }
    if (sicl_or_adler32(latent, n) != get32(in + 44)) return -74;
    return (long long)n;
}
