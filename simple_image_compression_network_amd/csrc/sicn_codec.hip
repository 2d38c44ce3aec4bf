// Latent container + static rANS coder on the GPU (include/sicn_codec.h; specification:
// oracle/sicn_codec_oracle.c).  New functionality: nothing in the reference corresponds to it
// (SURVEY.md §8f rows 1-2), parity status "unpinned".
//
// Parallelisation: the latent is cut into independent streams of 1024 symbols; ONE LANE encodes or
// decodes one stream (rANS is inherently serial inside a stream), so a 4K latent (6.2 M symbols)
// is 6 075 lanes.  Stream sizes are data dependent: every lane writes its stream backwards into a
// fixed-capacity scratch slot, a wavefront-level prefix scan (k_scan: __shfl_up inside a wave,
// LDS across the 16 waves of one workgroup, running carry across chunks) turns the sizes into
// offsets, and a copy kernel compacts the streams into the container.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/sicn.h"
#include "../../include/sicn_codec.h"

namespace {

constexpr uint32_t SS = SICN_CODEC_STREAM_SYMBOLS;
constexpr uint32_t CAP = 2 * SS + 16;  // scratch bytes per stream (12-bit worst case is 1.5 B/symbol + 4)
constexpr uint32_t RANS_L = 1u << 23;
constexpr int PROB_BITS = 12;
constexpr uint32_t ADLER_MOD = 65521u;

struct Workspace {  // device pointers carved out of the caller's workspace
    uint32_t *hist;                // [256]
    unsigned long long *sums;      // [2]: sum d_i, sum (n-i) d_i, both reduced mod 65521 per lane
    uint16_t *freq;                // [128]
    uint32_t *lens;                // [ns]
    uint32_t *offsets;             // [ns + 1]
    uint8_t *scratch;              // [ns][CAP]
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t carve(Workspace &w, void *base, uint32_t ns, bool with_scratch)
{
    uint8_t *p = (uint8_t *)base;
    size_t off = 0;
    w.hist = (uint32_t *)(p + off); off += 1024;
    w.sums = (unsigned long long *)(p + off); off += 64;
    w.freq = (uint16_t *)(p + off); off += 256;
    w.lens = (uint32_t *)(p + off); off += align_up(4 * (size_t)ns + 4, 64);
    w.offsets = (uint32_t *)(p + off); off += align_up(4 * (size_t)ns + 4, 64);
    w.scratch = p + off;
    if (with_scratch) off += (size_t)ns * CAP;
    return off;
}

// ---- kernels ----------------------------------------------------------------------------------
// histogram (256 bins) + the two sums adler32 is made of; one lane per stream of 1024 symbols
__global__ __launch_bounds__(256) void k_stats(const uint8_t *__restrict__ lat, uint32_t n, uint32_t ns,
                                               uint32_t *__restrict__ hist, unsigned long long *__restrict__ sums)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t st = blockIdx.x * 256 + threadIdx.x;
    unsigned long long s1 = 0, s2 = 0;
    if (st < ns) {
        const uint32_t begin = st * SS, cnt = min(SS, n - begin);
        for (uint32_t i = 0; i < cnt; i++) {
            const uint32_t d = lat[begin + i];
            atomicAdd(&h[d], 1u);
            s1 += d;
            s2 += (unsigned long long)(n - (begin + i)) * d;
        }
        s2 %= ADLER_MOD;
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
    if (st < ns) {
        atomicAdd(&sums[0], s1);
        atomicAdd(&sums[1], s2);
    }
}

__global__ __launch_bounds__(256) void k_rans_encode(const uint8_t *__restrict__ lat, uint32_t n, uint32_t ns,
                                                     const uint16_t *__restrict__ freq_g, uint8_t *__restrict__ scratch,
                                                     uint32_t *__restrict__ lens)
{
    __shared__ uint16_t freq[128], cum[128];
    if (threadIdx.x < 128) freq[threadIdx.x] = freq_g[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int s = 0; s < 128; s++) { cum[s] = (uint16_t)c; c += freq[s]; }
    }
    __syncthreads();
    const uint32_t st = blockIdx.x * 256 + threadIdx.x;
    if (st >= ns) return;
    const uint32_t begin = st * SS, cnt = min(SS, n - begin);
    uint8_t *buf = scratch + (size_t)st * CAP;
    uint32_t x = RANS_L, pos = CAP;
    for (uint32_t i = cnt; i-- > 0;) {
        const uint32_t s = lat[begin + i], f = freq[s], c = cum[s];
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
    buf[--pos] = (uint8_t)x;
    lens[st] = CAP - pos;
}

// Exclusive prefix sum of `in[0..n)` into out[0..n], out[n] = total.  One workgroup of 1024 lanes:
// wavefront-level scan with __shfl_up, wave totals combined through LDS, carry across chunks.
// `in` may be unaligned container bytes (read byte-wise when `in_bytes` != nullptr).
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *__restrict__ in, const uint8_t *__restrict__ in_bytes,
                                               uint32_t n, uint32_t *__restrict__ out, uint8_t *__restrict__ table_out,
                                               uint8_t *__restrict__ total_out)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v = 0;
        if (i < n) {
            if (in_bytes) {
                const uint8_t *p = in_bytes + 4 * (size_t)i;
                v = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            } else
                v = in[i];
        }
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int k = 0; k < wv; k++) wave_off += wave_tot[k];
        const uint32_t carry = carry_s;
        if (i < n) {
            out[i] = carry + wave_off + incl - v;
            if (table_out) {  // the container's per-stream byte counts, little-endian
                uint8_t *p = table_out + 4 * (size_t)i;
                p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t tot = carry_s;
        out[n] = tot;
        if (total_out) {
            total_out[0] = (uint8_t)tot; total_out[1] = (uint8_t)(tot >> 8);
            total_out[2] = (uint8_t)(tot >> 16); total_out[3] = (uint8_t)(tot >> 24);
        }
    }
}

// one workgroup per stream: scratch tail -> payload
__global__ __launch_bounds__(256) void k_compact(const uint8_t *__restrict__ scratch, const uint32_t *__restrict__ lens,
                                                 const uint32_t *__restrict__ offsets, uint8_t *__restrict__ payload)
{
    const uint32_t st = blockIdx.x, len = lens[st];
    const uint8_t *src = scratch + (size_t)st * CAP + (CAP - len);
    uint8_t *dst = payload + offsets[st];
    for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_rans_decode(const uint8_t *__restrict__ payload, const uint8_t *__restrict__ freq_bytes,
                                                     const uint32_t *__restrict__ offsets, uint32_t n, uint32_t ns,
                                                     uint8_t *__restrict__ lat, uint32_t *__restrict__ err)
{
    __shared__ uint16_t freq[128], cum[128];
    __shared__ uint8_t slot[4096];
    if (threadIdx.x < 128) freq[threadIdx.x] = (uint16_t)(freq_bytes[2 * threadIdx.x] | (freq_bytes[2 * threadIdx.x + 1] << 8));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0;
        for (int s = 0; s < 128; s++) { cum[s] = (uint16_t)c; c += freq[s]; }
    }
    __syncthreads();
    if (threadIdx.x < 128)
        for (uint32_t v = cum[threadIdx.x]; v < (uint32_t)cum[threadIdx.x] + freq[threadIdx.x] && v < 4096; v++)
            slot[v] = (uint8_t)threadIdx.x;
    __syncthreads();
    const uint32_t st = blockIdx.x * 256 + threadIdx.x;
    if (st >= ns) return;
    const uint32_t begin = st * SS, cnt = min(SS, n - begin);
    const uint8_t *p = payload + offsets[st], *end = payload + offsets[st + 1];
    if (end - p < 4) { atomicOr(err, 1u); return; }
    uint32_t x = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    p += 4;
    for (uint32_t i = 0; i < cnt; i++) {
        const uint32_t v = x & 4095u, s = slot[v];
        lat[begin + i] = (uint8_t)s;
        x = freq[s] * (x >> PROB_BITS) + v - cum[s];
        while (x < RANS_L) {
            if (p >= end) { atomicOr(err, 1u); return; }
            x = (x << 8) | *p++;
        }
    }
    if (p != end || x != RANS_L) atomicOr(err, 1u);
}

__global__ __launch_bounds__(256) void k_pack7(const uint8_t *__restrict__ lat, uint32_t n, uint8_t *__restrict__ out)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= (n + 7) / 8) return;
    unsigned long long v = 0;
    for (int k = 0; k < 8; k++)
        if (g * 8 + k < n) v |= (unsigned long long)lat[g * 8 + k] << (7 * k);
    for (int k = 0; k < 7; k++) out[7 * (size_t)g + k] = (uint8_t)(v >> (8 * k));
}

__global__ __launch_bounds__(256) void k_unpack7(const uint8_t *__restrict__ in, uint32_t n, uint8_t *__restrict__ lat)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= (n + 7) / 8) return;
    unsigned long long v = 0;
    for (int k = 0; k < 7; k++) v |= (unsigned long long)in[7 * (size_t)g + k] << (8 * k);
    for (int k = 0; k < 8; k++)
        if (g * 8 + k < n) lat[g * 8 + k] = (uint8_t)((v >> (7 * k)) & 127);
}

// ---- host helpers --------------------------------------------------------------------------------
void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
uint32_t get32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t get16(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8); }

// histogram -> 12-bit frequencies; identical to sicl_or_normalize in the oracle's specification
bool normalize(const uint32_t *h, uint32_t n, uint16_t *f)
{
    if (n == 0) return false;
    long long sum = 0;
    for (int s = 0; s < 128; s++) {
        unsigned long long v = h[s] ? ((unsigned long long)h[s] * 4096u) / n : 0;
        if (h[s] && v == 0) v = 1;
        f[s] = (uint16_t)v;
        sum += (long long)v;
    }
    long long diff = 4096 - sum;
    while (diff != 0) {
        int best = -1;
        for (int s = 0; s < 128; s++)
            if (f[s] > 0 && (diff > 0 || f[s] > 1) && (best < 0 || f[s] > f[best])) best = s;
        if (best < 0) return false;
        const long long step = diff > 0 ? diff : (diff < 1 - (long long)f[best] ? 1 - (long long)f[best] : diff);
        f[best] = (uint16_t)((long long)f[best] + step);
        diff -= step;
    }
    return true;
}

uint32_t adler_from_sums(unsigned long long s1, unsigned long long s2, uint32_t n)
{
    const uint32_t a = (uint32_t)((1 + s1) % ADLER_MOD);
    const uint32_t b = (uint32_t)((n % ADLER_MOD + s2) % ADLER_MOD);
    return (b << 16) | a;
}

#define HIP_TRY(x)                         \
    do {                                   \
        if ((x) != hipSuccess) return SICN_ENODEV; \
    } while (0)

}  // namespace

extern "C" size_t sicn_codec_max_bytes(int mode, uint32_t n)
{
    const uint32_t ns = (n + SS - 1) / SS;
    if (mode == SICN_CODEC_RAW8) return SICN_CODEC_HEADER_BYTES + (size_t)n;
    if (mode == SICN_CODEC_PACKED7) return SICN_CODEC_HEADER_BYTES + ((size_t)n + 7) / 8 * 7;
    if (mode == SICN_CODEC_RANS) return SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns + 2 * (size_t)n + 8 * (size_t)ns;
    return 0;
}

extern "C" size_t sicn_codec_workspace_bytes(int mode, uint32_t n)
{
    Workspace w;
    const uint32_t ns = (n + SS - 1) / SS;
    return carve(w, nullptr, ns, mode == SICN_CODEC_RANS) + 64;
}

extern "C" int sicn_codec_parse_header(const uint8_t *h, size_t bytes, sicn_codec_info *info)
{
    if (!h || !info || bytes < SICN_CODEC_HEADER_BYTES) return SICN_EINVAL;
    if (std::memcmp(h, "SICL", 4) || get16(h + 4) != 1) return SICN_EINVAL;
    info->mode = get16(h + 6);
    info->image_width = get32(h + 8);
    info->image_height = get32(h + 12);
    info->lat_w = get32(h + 16);
    info->lat_h = get32(h + 20);
    info->lat_c = get32(h + 24);
    info->n_symbols = get32(h + 28);
    info->n_streams = get32(h + 32);
    info->payload_bytes = get32(h + 40);
    info->adler32 = get32(h + 44);
    if (info->mode > 2 || get32(h + 36) != SS) return SICN_EINVAL;
    if ((unsigned long long)info->lat_w * info->lat_h * info->lat_c != info->n_symbols) return SICN_EINVAL;
    if (info->n_streams != (info->n_symbols + SS - 1) / SS) return SICN_EINVAL;
    return SICN_OK;
}

extern "C" int sicn_codec_encode(int mode, const uint8_t *latent, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                                 uint32_t img_w, uint32_t img_h, uint8_t *out, size_t out_capacity, size_t *out_bytes,
                                 void *workspace, size_t workspace_bytes, void *hip_stream)
{
    if (!out || !out_bytes || mode < 0 || mode > 2) return SICN_EINVAL;
    const unsigned long long n64 = (unsigned long long)lat_w * lat_h * lat_c;
    if (n64 > 0x7fffffffull || (n64 && !latent)) return SICN_EINVAL;
    const uint32_t n = (uint32_t)n64, ns = (n + SS - 1) / SS;
    if (out_capacity < sicn_codec_max_bytes(mode, n)) return SICN_ENOSPC;
    if (!workspace || workspace_bytes < sicn_codec_workspace_bytes(mode, n)) return SICN_ENOSPC;
    hipStream_t stream = (hipStream_t)hip_stream;
    Workspace w;
    carve(w, workspace, ns, mode == SICN_CODEC_RANS);

    // statistics (also the checksum and the symbol-range check) -------------------------------
    HIP_TRY(hipMemsetAsync(w.hist, 0, 1024 + 64, stream));
    if (ns) hipLaunchKernelGGL(k_stats, dim3((ns + 255) / 256), dim3(256), 0, stream, latent, n, ns, w.hist, w.sums);
    uint32_t hist[256];
    unsigned long long sums[2];
    HIP_TRY(hipMemcpyAsync(hist, w.hist, sizeof hist, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(sums, w.sums, sizeof sums, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (int s = 128; s < 256; s++)
        if (hist[s]) return SICN_EINVAL;

    uint8_t head[SICN_CODEC_HEADER_BYTES + 256];
    std::memset(head, 0, sizeof head);
    std::memcpy(head, "SICL", 4);
    put16(head + 4, 1);
    put16(head + 6, (uint32_t)mode);
    put32(head + 8, img_w);
    put32(head + 12, img_h);
    put32(head + 16, lat_w);
    put32(head + 20, lat_h);
    put32(head + 24, lat_c);
    put32(head + 28, n);
    put32(head + 32, ns);
    put32(head + 36, SS);
    put32(head + 44, adler_from_sums(sums[0], sums[1], n));

    size_t total;
    if (mode == SICN_CODEC_RAW8) {
        put32(head + 40, n);
        HIP_TRY(hipMemcpyAsync(out, head, SICN_CODEC_HEADER_BYTES, hipMemcpyHostToDevice, stream));
        if (n) HIP_TRY(hipMemcpyAsync(out + SICN_CODEC_HEADER_BYTES, latent, n, hipMemcpyDeviceToDevice, stream));
        total = SICN_CODEC_HEADER_BYTES + (size_t)n;
    } else if (mode == SICN_CODEC_PACKED7) {
        const uint32_t groups = (n + 7) / 8;
        put32(head + 40, groups * 7);
        HIP_TRY(hipMemcpyAsync(out, head, SICN_CODEC_HEADER_BYTES, hipMemcpyHostToDevice, stream));
        if (groups)
            hipLaunchKernelGGL(k_pack7, dim3((groups + 255) / 256), dim3(256), 0, stream, latent, n,
                               out + SICN_CODEC_HEADER_BYTES);
        total = SICN_CODEC_HEADER_BYTES + (size_t)groups * 7;
    } else {
        uint16_t freq[128];
        std::memset(freq, 0, sizeof freq);
        if (n && !normalize(hist, n, freq)) return SICN_EINVAL;
        for (int s = 0; s < 128; s++) put16(head + SICN_CODEC_HEADER_BYTES + 2 * s, freq[s]);
        HIP_TRY(hipMemcpyAsync(out, head, sizeof head, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(w.freq, freq, sizeof freq, hipMemcpyHostToDevice, stream));
        uint8_t *table = out + SICN_CODEC_HEADER_BYTES + 256;
        uint8_t *payload = table + 4 * (size_t)ns;
        if (ns) hipLaunchKernelGGL(k_rans_encode, dim3((ns + 255) / 256), dim3(256), 0, stream, latent, n, ns, w.freq, w.scratch, w.lens);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, w.lens, (const uint8_t *)nullptr, ns, w.offsets, table, out + 40);
        if (ns) hipLaunchKernelGGL(k_compact, dim3(ns), dim3(256), 0, stream, w.scratch, w.lens, w.offsets, payload);
        uint32_t payload_bytes = 0;
        HIP_TRY(hipMemcpyAsync(&payload_bytes, w.offsets + ns, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        total = SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns + payload_bytes;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (hipGetLastError() != hipSuccess) return SICN_ENODEV;
    *out_bytes = total;
    return SICN_OK;
}

extern "C" int sicn_codec_decode(const uint8_t *container, size_t bytes, uint8_t *latent, size_t latent_capacity,
                                 sicn_codec_info *info_out, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    if (!container || bytes < SICN_CODEC_HEADER_BYTES) return SICN_EINVAL;
    hipStream_t stream = (hipStream_t)hip_stream;
    uint8_t head[SICN_CODEC_HEADER_BYTES];
    HIP_TRY(hipMemcpyAsync(head, container, sizeof head, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    sicn_codec_info info;
    int rc = sicn_codec_parse_header(head, sizeof head, &info);
    if (rc) return rc;
    if (info_out) *info_out = info;
    const uint32_t n = info.n_symbols, ns = info.n_streams;
    if (n > 0x7fffffffu) return SICN_EINVAL;
    if (n && (!latent || latent_capacity < n)) return SICN_ENOSPC;
    if (!workspace || workspace_bytes < sicn_codec_workspace_bytes((int)info.mode, n)) return SICN_ENOSPC;
    Workspace w;
    carve(w, workspace, ns, false);
    HIP_TRY(hipMemsetAsync(w.hist, 0, 1024 + 64, stream));
    uint32_t *err = w.hist + 255;  // bins >= 128 stay zero for a valid latent: reuse the last one as error flag

    const uint8_t *payload = container + SICN_CODEC_HEADER_BYTES;
    if (info.mode == SICN_CODEC_RAW8) {
        if (info.payload_bytes != n || bytes < SICN_CODEC_HEADER_BYTES + (size_t)n) return SICN_EINVAL;
        if (n) HIP_TRY(hipMemcpyAsync(latent, payload, n, hipMemcpyDeviceToDevice, stream));
    } else if (info.mode == SICN_CODEC_PACKED7) {
        const uint32_t groups = (n + 7) / 8;
        if (info.payload_bytes != groups * 7 || bytes < SICN_CODEC_HEADER_BYTES + (size_t)groups * 7) return SICN_EINVAL;
        if (groups) hipLaunchKernelGGL(k_unpack7, dim3((groups + 255) / 256), dim3(256), 0, stream, payload, n, latent);
    } else {
        const size_t fixed = SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns;
        if (bytes < fixed + info.payload_bytes) return SICN_EINVAL;
        const uint8_t *freq_bytes = payload, *table = payload + 256;
        payload = table + 4 * (size_t)ns;
        if (n) {  // the table must sum to 4096 and the stream sizes to payload_bytes
            uint8_t fb[256];
            HIP_TRY(hipMemcpyAsync(fb, freq_bytes, 256, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            uint32_t sum = 0;
            for (int s = 0; s < 128; s++) sum += get16(fb + 2 * s);
            if (sum != 4096) return SICN_EINVAL;
        }
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, (const uint32_t *)nullptr, table, ns, w.offsets,
                           (uint8_t *)nullptr, (uint8_t *)nullptr);
        uint32_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, w.offsets + ns, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (total != info.payload_bytes) return SICN_EINVAL;
        if (ns) hipLaunchKernelGGL(k_rans_decode, dim3((ns + 255) / 256), dim3(256), 0, stream, payload, freq_bytes, w.offsets, n, ns, latent, err);
    }
    // checksum of what was decoded
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, err, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (flag) return SICN_EINVAL;
    HIP_TRY(hipMemsetAsync(w.hist, 0, 1024 + 64, stream));
    if (ns) hipLaunchKernelGGL(k_stats, dim3((ns + 255) / 256), dim3(256), 0, stream, latent, n, ns, w.hist, w.sums);
    unsigned long long sums[2];
    HIP_TRY(hipMemcpyAsync(sums, w.sums, sizeof sums, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (hipGetLastError() != hipSuccess) return SICN_ENODEV;
    return adler_from_sums(sums[0], sums[1], n) == info.adler32 ? SICN_OK : SICN_EBADMSG;
}
