/*
 * ORACLE — test infrastructure, NOT product code.
 *
 * CPU statement of the latent container / entropy coder of include/sicn_codec.h ("SICL" v1).
 * THIS PART HAS NO COUNTERPART IN THE REFERENCE (the reference never
 * encodes its latent; `conv_3_out` is an in-memory stream, conv_nonsquare_top.cpp:322-325), so its
 * parity is UNPINNED: the specification below is this project's own, and the tests can only show
 * decode(encode(x)) == x and GPU == this file, byte for byte.
 *
 * Specification (all integers little-endian)
 *   header, 48 bytes:
 *     0  char  magic[4] = "SICL"        4  u16 version = 1        6  u16 mode (0 raw8, 1 packed7, 2 rans, 3 rans-w)
 *     8  u32 image_width  12 u32 image_height   (the RGB image the latent came from)
 *    16  u32 lat_w        20 u32 lat_h          24 u32 lat_c      28 u32 n_symbols = lat_w*lat_h*lat_c
 *    32  u32 n_streams    36 u32 stream_symbols (1024; mode 3: the encoder's choice of 1024 .. 16384, a power of two,
 *                            default 16384; the last stream may be shorter)
 *    40  u32 payload_bytes                      44 u32 adler32 of the n_symbols latent bytes
 *   modes 2, 3: u16 freq[128] (sum 4096), then u32 stream_bytes[n_streams]
 *   payload: mode 0: the latent bytes ([lat_h][lat_w][lat_c] order); mode 1: 8 symbols -> 7 bytes
 *            (symbol i in bits [7i, 7i+7) of a 56-bit little-endian group; a short last group is
 *            zero-padded); mode 2: the streams back to back.
 *   symbols must be < 128 (every layer output is, conv_nonsquare_top.cpp:273-275).
 *
 *   rANS (the public-domain byte-wise rANS of F. Giesen, "ryg_rans"): 12-bit frequencies,
 *   state x in [2^23, 2^31), one static table for the whole latent.
 *     encode symbol s (start c, freq f), symbols taken in REVERSE order, bytes written BACKWARDS:
 *        x_max = ((2^23 >> 12) << 8) * f;  while (x >= x_max) { emit(x & 255); x >>= 8; }
 *        x = ((x / f) << 12) + (x % f) + c
 *     flush: emit the 4 state bytes so that the decoder reads them first, least significant first
 *     decode: x = first 4 bytes; repeat: v = x & 4095; s = symbol with c[s] <= v < c[s+1];
 *        x = f[s] * (x >> 12) + v - c[s];  while (x < 2^23) x = (x << 8) | next byte
 *   rANS-W (mode 3), the wavefront form: a stream holds 16384 symbols coded by 64 INTERLEAVED rANS
 *   states (the 64 lanes of a wavefront) that share one stream of 16-bit words.  Symbol j of the
 *   stream belongs to lane (j / 4) % 64, step 4 * (j / 256) + j % 4: a lane owns 4 consecutive bytes of
 *   every 256-byte block (one aligned dword load / store per lane and block on the GPU).  State x in
 *   [2^16, 2^32), renormalisation by one 16-bit word (at most one per symbol):
 *     encode, steps in REVERSE order, every lane that has a symbol in the step:
 *        if (x >= 2^20 * f) { emit(x & 0xFFFF); x >>= 16; }      x = ((x / f) << 12) + (x % f) + c
 *        the words a step emits stand in ascending lane order, in front of the words of the later steps
 *     flush: the 64 final states as u32, lane 0 first, in front of everything
 *     decode: x[lane] = the 64 u32; steps forward: v = x & 4095; s = symbol(v); x = f*(x>>12) + v - c;
 *        then the lanes with x < 2^16, in ascending order, each take the next word: x = (x << 16) | word.
 *        At the end every state is 2^16 again and every word has been read.
 *   On the GPU a step is one wavefront instruction stream; the position of a lane's word is the popcount
 *   of the emitting (reading) lanes below it — a wavefront-level scan of one ballot.
 *
 *   frequency table from the histogram h[] of all n symbols:
 *        f[s] = h[s] ? max(1, floor(h[s] * 4096 / n)) : 0;  then while sum != 4096: add to / take
 *        from the symbol with the largest f (lowest index on ties; never below 1).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SICL_HEADER 48
#define SICL_STREAM_SYMBOLS 1024u
#define SICL_WSTREAM_SYMBOLS 16384u      /* mode 3: 64 lanes x 256 steps */
#define SICL_LANES 64u
#define RANSW_L (1u << 16)
#define RANS_L (1u << 23)
#define PROB_BITS 12

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static uint32_t get32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t get16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }

uint32_t sicl_or_adler32(const uint8_t *d, size_t n)
{
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; i++) {
        a = (a + d[i]) % 65521u;
        b = (b + a) % 65521u;
    }
    return (b << 16) | a;
}

/* histogram -> 12-bit frequencies (see the specification above). Returns 0, or -1 if n == 0. */
int sicl_or_normalize(const uint32_t h[128], uint32_t n, uint16_t f[128])
{
    if (n == 0) return -1;
    int64_t sum = 0;
    for (int s = 0; s < 128; s++) {
        uint64_t v = h[s] ? ((uint64_t)h[s] * 4096u) / n : 0;
        if (h[s] && v == 0) v = 1;
        f[s] = (uint16_t)v;
        sum += (int64_t)v;
    }
    int64_t diff = 4096 - sum;
    while (diff != 0) {
        int best = -1;
        for (int s = 0; s < 128; s++)
            if (f[s] > 0 && (diff > 0 || f[s] > 1) && (best < 0 || f[s] > f[best])) best = s;
        if (best < 0) return -1;
        int64_t step = diff > 0 ? diff : (diff < 1 - (int64_t)f[best] ? 1 - (int64_t)f[best] : diff);
        f[best] = (uint16_t)((int64_t)f[best] + step);
        diff -= step;
    }
    return 0;
}

static uint32_t stream_symbols(int mode) { return mode == 3 ? SICL_WSTREAM_SYMBOLS : SICL_STREAM_SYMBOLS; }

/* Mode 3 with a stream length chosen by the encoder (header dword 9 carries it): a power of two, 1024 .. 16384.  Shorter
 * streams = more independent waves on the GPU = a shorter critical path for small latents, at 260 bytes per extra stream
 * (the 64 final states + the length entry).  Everything else about the format is the same. */
static int wstream_ok(uint32_t ss) { return ss >= 1024u && ss <= SICL_WSTREAM_SYMBOLS && (ss & (ss - 1u)) == 0; }

size_t sicl_or_max_bytes_sl(int mode, uint32_t n, uint32_t ss)
{
    if (mode == 3 ? !wstream_ok(ss) : ss != stream_symbols(mode)) return 0;   /* not a stream length of this mode */
    const uint32_t ns = (n + ss - 1) / ss;
    if (mode == 3) return SICL_HEADER + 256 + 4 * (size_t)ns + 2 * (size_t)n + 256 * (size_t)ns;
    if (mode == 0) return SICL_HEADER + (size_t)n;
    if (mode == 1) return SICL_HEADER + ((size_t)n + 7) / 8 * 7;
    return SICL_HEADER + 256 + 4 * (size_t)ns + 2 * (size_t)n + 8 * (size_t)ns;
}

size_t sicl_or_max_bytes(int mode, uint32_t n) { return sicl_or_max_bytes_sl(mode, n, stream_symbols(mode)); }

/* one stream: returns bytes produced, written at the END of buf[0..cap) */
static uint32_t rans_encode_stream(const uint8_t *sym, uint32_t n, const uint16_t *freq, const uint16_t *cum,
                                   uint8_t *buf, uint32_t cap)
{
    uint32_t x = RANS_L, pos = cap;
    for (uint32_t i = n; i-- > 0;) {
        const uint32_t f = freq[sym[i]], c = cum[sym[i]];
        const uint32_t x_max = ((RANS_L >> PROB_BITS) << 8) * f;
        while (x >= x_max) {
            buf[--pos] = (uint8_t)(x & 0xFF);
            x >>= 8;
        }
        x = ((x / f) << PROB_BITS) + (x % f) + c;
    }
    buf[--pos] = (uint8_t)(x >> 24);
    buf[--pos] = (uint8_t)(x >> 16);
    buf[--pos] = (uint8_t)(x >> 8);
    buf[--pos] = (uint8_t)(x);
    return cap - pos;
}

/* rANS-W: index of the symbol lane l codes in step i */
static uint32_t ransw_index(uint32_t i, uint32_t l) { return (i >> 2) * 256u + l * 4u + (i & 3u); }

/* one rANS-W stream: returns bytes produced, written at the END of buf[0..cap) (cap even) */
static uint32_t ransw_encode_stream(const uint8_t *sym, uint32_t n, const uint16_t *freq, const uint16_t *cum,
                                    uint8_t *buf, uint32_t cap)
{
    uint32_t x[SICL_LANES], pos = cap;
    for (uint32_t l = 0; l < SICL_LANES; l++) x[l] = RANSW_L;
    const uint32_t steps = (n + 255u) / 256u * 4u;
    for (uint32_t i = steps; i-- > 0;) {
        /* the words of one step stand in ascending lane order: writing backwards, take the lanes downwards */
        for (uint32_t l = SICL_LANES; l-- > 0;) {
            const uint32_t j = ransw_index(i, l);
            if (j >= n) continue;
            const uint32_t f = freq[sym[j]];
            if ((uint64_t)x[l] >= ((uint64_t)f << 20)) {
                buf[--pos] = (uint8_t)(x[l] >> 8);
                buf[--pos] = (uint8_t)(x[l]);
                x[l] >>= 16;
            }
        }
        for (uint32_t l = 0; l < SICL_LANES; l++) {
            const uint32_t j = ransw_index(i, l);
            if (j >= n) continue;
            const uint32_t f = freq[sym[j]], c = cum[sym[j]];
            x[l] = ((x[l] / f) << PROB_BITS) + (x[l] % f) + c;
        }
    }
    for (uint32_t l = SICL_LANES; l-- > 0;) {
        pos -= 4;
        put32(buf + pos, x[l]);
    }
    return cap - pos;
}

/* Returns container bytes written, or a negative code (-22 bad argument / symbol >= 128, -28 no space). */
long long sicl_or_encode_sl(int mode, const uint8_t *latent, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                            uint32_t img_w, uint32_t img_h, uint8_t *out, size_t cap, uint32_t ss_enc)
{
    const uint64_t n64 = (uint64_t)lat_w * lat_h * lat_c;
    if (mode < 0 || mode > 3 || n64 > 0x7fffffffu) return -22;
    if (mode == 3 ? !wstream_ok(ss_enc) : ss_enc != stream_symbols(mode)) return -22;
    const uint32_t n = (uint32_t)n64;
    const uint32_t ns = (n + ss_enc - 1) / ss_enc;
    if (cap < sicl_or_max_bytes_sl(mode, n, ss_enc)) return -28;
    uint32_t h[256] = {0};
    for (uint32_t i = 0; i < n; i++) h[latent[i]]++;
    for (int s = 128; s < 256; s++)
        if (h[s]) return -22;
    memset(out, 0, SICL_HEADER);
    memcpy(out, "SICL", 4);
    put16(out + 4, 1);
    put16(out + 6, (uint32_t)mode);
    put32(out + 8, img_w);
    put32(out + 12, img_h);
    put32(out + 16, lat_w);
    put32(out + 20, lat_h);
    put32(out + 24, lat_c);
    put32(out + 28, n);
    put32(out + 32, ns);
    put32(out + 36, ss_enc);
    put32(out + 44, sicl_or_adler32(latent, n));
    size_t pos = SICL_HEADER;
    if (mode == 0) {
        memcpy(out + pos, latent, n);
        pos += n;
    } else if (mode == 1) {
        for (uint32_t g = 0; g < (n + 7) / 8; g++) {
            uint64_t v = 0;
            for (int k = 0; k < 8; k++)
                if (g * 8 + k < n) v |= (uint64_t)latent[g * 8 + k] << (7 * k);
            for (int k = 0; k < 7; k++) out[pos++] = (uint8_t)(v >> (8 * k));
        }
    } else {   /* modes 2 and 3 share the table */
        uint16_t freq[128], cum[129];
        if (n) {
            if (sicl_or_normalize(h, n, freq)) return -22;
        } else
            memset(freq, 0, sizeof freq);
        cum[0] = 0;
        for (int s = 0; s < 128; s++) cum[s + 1] = (uint16_t)(cum[s] + freq[s]);
        for (int s = 0; s < 128; s++) put16(out + pos + 2 * s, freq[s]);
        pos += 256;
        uint8_t *lens = out + pos;
        pos += 4 * (size_t)ns;
        const uint32_t bufcap = 2 * ss_enc + 256;
        uint8_t *buf = (uint8_t *)malloc(bufcap);
        if (!buf) return -12;
        for (uint32_t st = 0; st < ns; st++) {
            const uint32_t begin = st * ss_enc;
            const uint32_t cnt = n - begin < ss_enc ? n - begin : ss_enc;
            const uint32_t len = mode == 3 ? ransw_encode_stream(latent + begin, cnt, freq, cum, buf, bufcap)
                                           : rans_encode_stream(latent + begin, cnt, freq, cum, buf, bufcap);
            put32(lens + 4 * st, len);
            memcpy(out + pos, buf + bufcap - len, len);
            pos += len;
        }
        free(buf);
    }
    const size_t payload0 = SICL_HEADER + (mode >= 2 ? 256 + 4 * (size_t)ns : 0);
    put32(out + 40, (uint32_t)(pos - payload0));
    return (long long)pos;
}

long long sicl_or_encode(int mode, const uint8_t *latent, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                         uint32_t img_w, uint32_t img_h, uint8_t *out, size_t cap)
{
    return sicl_or_encode_sl(mode, latent, lat_w, lat_h, lat_c, img_w, img_h, out, cap, stream_symbols(mode < 0 || mode > 3 ? 0 : mode));
}

/* info[8] = mode, img_w, img_h, lat_w, lat_h, lat_c, n_symbols, payload_bytes. Returns symbols
 * decoded or a negative code (-22 malformed, -28 latent_cap too small, -74 checksum mismatch). */
long long sicl_or_decode(const uint8_t *in, size_t bytes, uint8_t *latent, size_t latent_cap, uint32_t info[8])
{
    if (bytes < SICL_HEADER || memcmp(in, "SICL", 4) || get16(in + 4) != 1) return -22;
    const uint32_t mode = get16(in + 6), n = get32(in + 28), ns = get32(in + 32), ss = get32(in + 36);
    const uint32_t payload = get32(in + 40);
    if (mode > 3 || (mode == 3 ? !wstream_ok(ss) : ss != stream_symbols((int)mode)) || ns != (n + ss - 1) / ss) return -22;
    if ((uint64_t)get32(in + 16) * get32(in + 20) * get32(in + 24) != n) return -22;
    if (info) {
        info[0] = mode; info[1] = get32(in + 8); info[2] = get32(in + 12); info[3] = get32(in + 16);
        info[4] = get32(in + 20); info[5] = get32(in + 24); info[6] = n; info[7] = payload;
    }
    if (latent_cap < n) return -28;
    size_t pos = SICL_HEADER;
    if (mode == 0) {
        if (payload != n || bytes < pos + n) return -22;
        memcpy(latent, in + pos, n);
    } else if (mode == 1) {
        if (payload != (size_t)((n + 7) / 8) * 7 || bytes < pos + payload) return -22;
        for (uint32_t g = 0; g < (n + 7) / 8; g++) {
            uint64_t v = 0;
            for (int k = 0; k < 7; k++) v |= (uint64_t)in[pos + 7 * (size_t)g + k] << (8 * k);
            for (int k = 0; k < 8; k++)
                if (g * 8 + k < n) latent[g * 8 + k] = (uint8_t)((v >> (7 * k)) & 127);
        }
    } else {
        if (bytes < pos + 256 + 4 * (size_t)ns) return -22;
        uint16_t freq[128], cum[129];
        uint32_t sum = 0;
        for (int s = 0; s < 128; s++) { freq[s] = (uint16_t)get16(in + pos + 2 * s); sum += freq[s]; }
        if (n && sum != 4096) return -22;
        cum[0] = 0;
        for (int s = 0; s < 128; s++) cum[s + 1] = (uint16_t)(cum[s] + freq[s]);
        uint8_t slot[4096];
        for (int s = 0; s < 128; s++)
            for (uint32_t v = cum[s]; v < cum[s + 1]; v++) slot[v] = (uint8_t)s;
        pos += 256;
        const uint8_t *lens = in + pos;
        pos += 4 * (size_t)ns;
        uint64_t total = 0;
        for (uint32_t st = 0; st < ns; st++) total += get32(lens + 4 * st);
        if (total != payload || bytes < pos + payload) return -22;
        for (uint32_t st = 0; st < ns; st++) {
            const uint32_t len = get32(lens + 4 * st), begin = st * ss, cnt = n - begin < ss ? n - begin : ss;
            if (mode == 3) {
                if (len < 4 * SICL_LANES || (len & 1)) return -22;
                const uint8_t *p = in + pos, *end = p + len;
                uint32_t x[SICL_LANES];
                for (uint32_t l = 0; l < SICL_LANES; l++, p += 4) x[l] = get32(p);
                const uint32_t steps = (cnt + 255u) / 256u * 4u;
                for (uint32_t i = 0; i < steps; i++) {
                    for (uint32_t l = 0; l < SICL_LANES; l++) {
                        const uint32_t j = ransw_index(i, l);
                        if (j >= cnt) continue;
                        const uint32_t v = x[l] & 4095u, s = slot[v];
                        latent[begin + j] = (uint8_t)s;
                        x[l] = freq[s] * (x[l] >> PROB_BITS) + v - cum[s];
                    }
                    for (uint32_t l = 0; l < SICL_LANES; l++) {
                        if (ransw_index(i, l) >= cnt || x[l] >= RANSW_L) continue;
                        if (p + 2 > end) return -22;
                        x[l] = (x[l] << 16) | get16(p);
                        p += 2;
                    }
                }
                for (uint32_t l = 0; l < SICL_LANES; l++)
                    if (x[l] != RANSW_L) return -22;
                if (p != end) return -22;
                pos += len;
                continue;
            }
            if (len < 4) return -22;
            const uint8_t *p = in + pos, *end = p + len;
            uint32_t x = get32(p);
            p += 4;
            for (uint32_t i = 0; i < cnt; i++) {
                const uint32_t v = x & 4095u, s = slot[v];
                latent[begin + i] = (uint8_t)s;
                x = freq[s] * (x >> PROB_BITS) + v - cum[s];
                while (x < RANS_L) {
                    if (p >= end) return -22;
                    x = (x << 8) | *p++;
                }
            }
            if (p != end || x != RANS_L) return -22;
            pos += len;
        }
    }
    if (sicl_or_adler32(latent, n) != get32(in + 44)) return -74;
    return (long long)n;
}
