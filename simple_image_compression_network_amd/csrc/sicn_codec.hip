// Latent container + static rANS coder on the GPU (include/sicn_codec.h; specification:
// oracle/sicn_codec_oracle.c).  New functionality: nothing in the reference corresponds to it
// (SURVEY.md §8f rows 1-2), parity status "unpinned".
//
// Two rANS forms.  Mode 3 ("rANS-W") is the wavefront form: a stream of 16384 symbols is coded by ONE
// WAVE, its 64 lanes holding 64 interleaved rANS states that share one stream of 16-bit words; in every
// step each lane codes one symbol and the lanes that renormalise find their word by a wavefront-level
// scan of one ballot (rank = popcount of the emitting lanes below).  Symbols are read and written 64
// consecutive bytes at a time.  Mode 2 is the first, lane-serial form, kept for compatibility:
// the latent is cut into independent streams of 1024 symbols; ONE LANE encodes or
// decodes one stream (rANS is inherently serial inside a stream), so a 4K latent (6.2 M symbols)
// is 6 075 lanes.  Stream sizes are data dependent: every lane writes its stream backwards into a
// fixed-capacity scratch slot, a wavefront-level prefix scan (k_scan: __shfl_up inside a wave,
// LDS across the 16 waves of one workgroup, running carry across chunks) turns the sizes into
// offsets, and a copy kernel compacts the streams into the container.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/sicn.h"
#include "../../include/sicn_codec.h"
#include "sicn_internal.h"   // chip_geom(): how many waves the device holds decides between the decoder's two table forms

namespace {

constexpr uint32_t SS = SICN_CODEC_STREAM_SYMBOLS;
constexpr uint32_t CAP = 2 * SS + 16;  // scratch bytes per stream (12-bit worst case is 1.5 B/symbol + 4)
constexpr uint32_t WSS = SICN_CODEC_WSTREAM_SYMBOLS;   // mode 3: 64 lanes x 256 steps
constexpr uint32_t WCAP = 2 * WSS + 256;               // at most one 16-bit word per symbol + the 64 final states
constexpr uint32_t RANSW_L = 1u << 16;
inline uint32_t stream_symbols(int mode) { return mode == SICN_CODEC_RANSW ? WSS : SS; }
inline uint32_t stream_cap(int mode) { return mode == SICN_CODEC_RANSW ? WCAP : CAP; }
// mode 3 with the encoder's choice of stream length (header dword 9): a power of two, 1024 .. 16384 symbols.  Shorter streams =
// more waves = a shorter serial chain for a small latent, + 260 bytes per stream (64 final states, one length entry).
__host__ __device__ inline bool wstream_ok(uint32_t wss) { return wss >= 1024u && wss <= WSS && (wss & (wss - 1u)) == 0; }
__host__ __device__ inline uint32_t wstream_cap(uint32_t wss) { return 2u * wss + 256u; }   // scratch bytes per stream (= WCAP at 16384)
constexpr uint32_t RANS_L = 1u << 23;
constexpr int PROB_BITS = 12;
constexpr uint32_t ADLER_MOD = 65521u;
// rANS modes: ceil(n / stream symbols) * scratch capacity must stay below 2^32 (32-bit stream offsets):
// 0x7F000000 / 16384 * 33024 = 0x7F000000 / 1024 * 2064 = 4 294 705 152 < 2^32
constexpr uint32_t MAX_RANS_SYMBOLS = 0x7F000000u;
static_assert((unsigned long long)(MAX_RANS_SYMBOLS / WSS) * WCAP < (1ull << 32), "mode 3 offsets would wrap");
static_assert((unsigned long long)(MAX_RANS_SYMBOLS / SS) * CAP < (1ull << 32), "mode 2 offsets would wrap");
// Asynchronous paths, small stream counts: a small image's coder time is launch latency (ten kernels of which seven were 5 us
// bookkeeping stages), so up to this many streams per image the scans are done by the consumers themselves (every wave sums the
// length table up to its own stream) and the statistics are per-workgroup rows instead of atomics on a block that has to be
// cleared first: encode 6 -> 4 launches, decode 4 -> 2.
constexpr uint32_t SELF_SCAN_MAX = 2048;
constexpr uint32_t STAT_ROWS = 64;                 // at most this many statistics workgroups (rows) per image in row mode
constexpr uint32_t STAT_ROW_WORDS = 256 + 4;       // hist[256], then s1, s2 as two u64

struct Workspace {  // device pointers carved out of the caller's workspace
    uint32_t *hist;                // [256]
    unsigned long long *sums;      // [2]: sum d_i, sum (n-i) d_i, both reduced mod 65521 per lane
    uint16_t *freq;                // [128]
    uint32_t *meta;                // [16] async paths: [0] error flags, [1] sanitised payload bytes, [2] header adler32
    uint32_t *lens;                // [ns]
    uint32_t *offsets;             // [ns + 1]
    uint8_t *scratch;              // [ns][CAP]
    uint32_t *rows;                // [STAT_ROWS][STAT_ROW_WORDS] async encode, row mode: per-workgroup histograms and checksum sums
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t carve(Workspace &w, void *base, uint32_t ns, size_t scratch_per_stream)
{
    uint8_t *p = (uint8_t *)base;
    size_t off = 0;
    w.hist = (uint32_t *)(p + off); off += 1024;
    w.sums = (unsigned long long *)(p + off); off += 64;
    w.freq = (uint16_t *)(p + off); off += 256;
    w.meta = (uint32_t *)(p + off); off += 64;
    w.lens = (uint32_t *)(p + off); off += align_up(4 * (size_t)ns + 4, 64);
    w.offsets = (uint32_t *)(p + off); off += align_up(4 * (size_t)ns + 4, 64);
    w.scratch = p + off;
    off += align_up((size_t)ns * scratch_per_stream, 64);
    w.rows = (uint32_t *)(p + off);
    off += (size_t)STAT_ROWS * STAT_ROW_WORDS * 4;
    return off;
}

// k_stats grid: at most 128 workgroups per image — each ends with up to 256 global atomics on the SAME 256 words, and those
// serialise in L2.  Measured r02 (tools/coder_speed.py, encode of a 1080p / of 8 4K latents): 32 workgroups 78 / 146 us,
// 128: 80 / 126, 512: 103 / 215.  (Per-workgroup partial histograms summed by the last workgroup to finish were slower
// still: 91 / 203 us — one workgroup walking 128 rows is a serial chain of its own.)
static inline unsigned stats_blocks(uint32_t n, uint32_t n_images)
{
    (void)n_images;
    return std::min((n / 16 + 255u) / 256u + 1u, 128u);
}

// ---- kernels ----------------------------------------------------------------------------------
// histogram (256 bins) + the two sums adler32 is made of; grid-stride over 16-byte groups (consecutive lanes read
// consecutive groups); zeros — half of a ReLU latent — are counted in a register instead of hammering one LDS bin
// Batches: blockIdx.y = image; pointers into the latent tensor / the per-image workspace / the container
// slots advance by the given byte strides (0 for single-image calls).
template <typename T>
__device__ __forceinline__ T *img_ptr(T *p, size_t stride)
{
    return (T *)((uint8_t *)p + (size_t)blockIdx.y * stride);
}
template <typename T>
__device__ __forceinline__ const T *img_ptr(const T *p, size_t stride)
{
    return (const T *)((const uint8_t *)p + (size_t)blockIdx.y * stride);
}

__global__ __launch_bounds__(256) void k_stats(const uint8_t *__restrict__ lat_, uint32_t n, uint32_t *__restrict__ hist_,
                                               unsigned long long *__restrict__ sums_, size_t s_lat, size_t s_ws,
                                               uint32_t *__restrict__ rows_ = nullptr)
{
    const uint8_t *lat = img_ptr(lat_, s_lat);
    uint32_t *hist = img_ptr(hist_, s_ws);
    unsigned long long *sums = img_ptr(sums_, s_ws);
    // 8 copies of the histogram, copy = lane & 7, at a pitch of 257 words: latents are skewed (a few small values carry most
    // of the mass), so lanes of one wave mostly hit the SAME bin — with one copy those LDS atomics serialise; copies of a bin
    // sit in 8 different banks
    __shared__ uint32_t h[8 * 257];
    for (int i = threadIdx.x; i < 8 * 257; i += 256) h[i] = 0;
    uint32_t *hl = h + (threadIdx.x & 7) * 257;
    __syncthreads();
    unsigned long long s1 = 0, s2 = 0;
    uint32_t zeros = 0;
    const bool vec = (reinterpret_cast<uintptr_t>(lat) & 15) == 0;
    const uint32_t groups = vec ? n / 16 : 0;
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < groups; g += gridDim.x * 256) {
        const uint4 q = reinterpret_cast<const uint4 *>(lat)[g];
        const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t d = (w4[k >> 2] >> (8 * (k & 3))) & 255u;
            if (d) {
                atomicAdd(&hl[d], 1u);
                s1 += d;
                s2 += (unsigned long long)(n - (16 * g + k)) * d;   // < 2^39 per term, far fewer than 2^25 terms per lane
            } else
                zeros++;
        }
    }
    for (uint32_t i = 16 * groups + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t d = lat[i];
        if (d) {
            atomicAdd(&hl[d], 1u);
            s1 += d;
            s2 += (unsigned long long)(n - i) * d;
        } else
            zeros++;
    }
    s2 %= ADLER_MOD;
    // one atomic per wave, not per lane: wavefront-level reduction first
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        s1 += __shfl_down(s1, d);
        s2 += __shfl_down(s2, d);
        zeros += __shfl_down(zeros, d);
    }
    if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(&h[0], zeros);
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) total += h[c * 257 + threadIdx.x];
    if (rows_) {   // row mode: this workgroup's own row, plain stores (nothing to clear, no atomics on shared words)
        __shared__ unsigned long long ws1[4], ws2[4];
        if ((threadIdx.x & 63) == 0) { ws1[threadIdx.x >> 6] = s1; ws2[threadIdx.x >> 6] = s2 % ADLER_MOD; }
        __syncthreads();
        uint32_t *row = img_ptr(rows_, s_ws) + (size_t)blockIdx.x * STAT_ROW_WORDS;
        row[threadIdx.x] = total;
        if (threadIdx.x == 0) {
            unsigned long long *q = (unsigned long long *)(row + 256);
            q[0] = ws1[0] + ws1[1] + ws1[2] + ws1[3];
            q[1] = (ws2[0] + ws2[1] + ws2[2] + ws2[3]) % ADLER_MOD;
        }
        return;
    }
    if ((threadIdx.x & 63) == 0) {
        if (s1) atomicAdd(&sums[0], s1);
        if (s2) atomicAdd(&sums[1], s2 % ADLER_MOD);
    }
    if (total) atomicAdd(&hist[threadIdx.x], total);
}

// ---- rANS-W: one wave (= one 64-lane workgroup) per stream -----------------------------------------
// Lane l codes symbols 256 q + 4 l + k (k = 0..3) in steps 4 q + k: one aligned dword of symbols per lane and
// 256-symbol block.  The 16-bit words live in LDS while the stream is coded (their positions are data
// dependent: rank of the lane among the renormalising lanes = popcount of a ballot) and move between LDS and
// global memory in whole coalesced runs.
struct RanswTab {
    uint32_t fc[128];   // freq | cum << 16
    uint32_t rcp[128];  // m = min(2^32 - 1, floor(2^32 / freq)): umulhi(x, m) is floor(x / freq) or one less, never more (see ransw_div)
};

// Exact x / f for x < 2^32, 1 <= f <= 4096, without an integer divide.  m = floor(2^32 / f) (2^32 - 1 for f = 1)
// satisfies m <= 2^32 / f, so e = floor(x m / 2^32) <= floor(x / f): the estimate is NEVER too large; and
// x / f - x m / 2^32 = x (2^32 / f - m) / 2^32 < x / 2^32 < 1, so e >= floor(x / f) - 1: ONE fix-up step is enough.
// (Round 1 used a float reciprocal whose rounded product could exceed the quotient for 295 of the 4096
// frequencies, e.g. f = 3815, x = 250046544; tests/test_codec.py::test_ransw_div_exhaustive covers all f.)
__host__ __device__ __forceinline__ uint32_t ransw_rcp(uint32_t f)
{
    return f <= 1 ? (f ? 0xFFFFFFFFu : 0u) : (uint32_t)(0x100000000ull / f);
}

__host__ __device__ __forceinline__ uint32_t ransw_div(uint32_t x, uint32_t f, uint32_t m, uint32_t &r)
{
    uint32_t q = (uint32_t)(((unsigned long long)x * m) >> 32);   // v_mul_hi_u32
    r = x - q * f;
    if (r >= f) { q++; r -= f; }
    return q;
}

__device__ __forceinline__ void ransw_build(RanswTab &t, const uint16_t *freq, int lane)
{
    // exclusive prefix sum of 128 frequencies by one wave: two elements per lane + wavefront scan
    const uint32_t f0 = freq[2 * lane], f1 = freq[2 * lane + 1];
    uint32_t incl = f0 + f1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const uint32_t c0 = incl - f0 - f1;
    t.fc[2 * lane] = f0 | (c0 << 16);
    t.fc[2 * lane + 1] = f1 | ((c0 + f0) << 16);
    t.rcp[2 * lane] = ransw_rcp(f0);
    t.rcp[2 * lane + 1] = ransw_rcp(f1);
}

// The 16-bit words of a stream pass through a small LDS RING (8 KB) instead of a buffer sized for the worst case (33 KB):
// the worst case still fits the stream's scratch slot in global memory, but a wave now costs 9 KB (mode 3) / 24 KB (mode 4) of
// LDS, so a CU holds all of its streams at once (a 4K latent gives a CU about 12) instead of 4 at a time — the coders are
// latency-bound serial chains, occupancy is their only source of throughput.
//   encoder: words are produced at DEcreasing global word indices gpos-1, gpos-2, ..; ring slot = index % RING_WORDS; whenever
//            fewer than one step's worth (64) of slots is left, the words [gpos, top) go out to the scratch slot.
//   decoder: words are consumed at INcreasing indices; before every block of 4 steps (<= 256 words) the ring is topped up.
constexpr uint32_t RING_WORDS = 4096;
__device__ __forceinline__ void ring_flush(const uint16_t *ring, uint16_t *dst, uint32_t gpos, uint32_t top, uint32_t lane)
{
    // word i of the stream sits at ring[i % RING_WORDS] and goes to dst[i]: 8-word groups at multiples of 8 are contiguous and
    // 16-byte aligned on both sides (RING_WORDS % 8 == 0, dst is a scratch slot at a 16-byte multiple), so the body moves 16 bytes
    // per lane and instruction; the ragged head and tail go word by word (round 4: the whole run used to go 2 bytes at a time)
    const bool vec = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    const uint32_t a = vec ? min(top, (gpos + 7u) & ~7u) : top, b = vec ? max(a, top & ~7u) : top;
    for (uint32_t i = gpos + lane; i < a; i += 64) dst[i] = ring[i & (RING_WORDS - 1)];
    for (uint32_t i = a + 8 * lane; i < b; i += 8 * 64)
        *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(ring + (i & (RING_WORDS - 1)));
    for (uint32_t i = b + lane; i < top; i += 64) dst[i] = ring[i & (RING_WORDS - 1)];
}
__device__ __forceinline__ uint32_t ring_fill(uint16_t *ring, const uint16_t *src, uint32_t loaded, uint32_t upto, uint32_t lane)
{
    // the payload side is only 2-byte aligned (streams start at even container offsets): dwords when the source happens to be
    // 4-byte aligned relative to the ring's even word indices, else word by word
    if ((reinterpret_cast<uintptr_t>(src) & 3) == 0) {
        const uint32_t a = min(upto, (loaded + 1u) & ~1u), b = max(a, upto & ~1u);
        if (lane == 0 && loaded < a) ring[loaded & (RING_WORDS - 1)] = src[loaded];
        for (uint32_t i = a + 2 * lane; i < b; i += 128)
            *reinterpret_cast<uint32_t *>(ring + (i & (RING_WORDS - 1))) = *reinterpret_cast<const uint32_t *>(src + i);
        if (lane == 0 && b < upto) ring[b & (RING_WORDS - 1)] = src[b];
        return upto;
    }
    for (uint32_t i = loaded + lane; i < upto; i += 64) ring[i & (RING_WORDS - 1)] = src[i];
    return upto;
}

// ---- the encoder's per-symbol table: one 16-byte entry (ONE ds_read_b128) instead of two dword tables, and the renormalisation
// ---- test as a precomputed threshold.  Round 4: the step loop used to make two dependent LDS round trips per symbol (frequency,
// ---- then reciprocal, each behind an s_waitcnt and a branch); now the four entries of a block are fetched while the PREVIOUS
// ---- block's chain runs and the step itself is branch-free.
struct RanswEnt {
    uint32_t fc;    // freq | cum << 16
    uint32_t rcp;   // ransw_rcp(freq)
    uint32_t thr;   // a lane renormalises iff x > thr: freq * 2^20 - 1 (freq = 4096: never, 2^32 - 1; freq = 0: unused symbol)
    uint32_t pad;
};
__device__ __forceinline__ RanswEnt ransw_ent(uint32_t f, uint32_t c)
{
    return RanswEnt{f | (c << 16), ransw_rcp(f), (f == 0 || f >= 4096u) ? 0xFFFFFFFFu : (f << 20) - 1u, 0u};
}
// table of one wave from this lane's two frequencies (symbols 2 lane, 2 lane + 1): exclusive prefix sum by a wavefront scan
__device__ __forceinline__ void ransw_build_ent(RanswEnt *ent, uint32_t f0, uint32_t f1, int lane)
{
    uint32_t incl = f0 + f1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    const uint32_t c0 = incl - f0 - f1;
    ent[2 * lane] = ransw_ent(f0, c0);
    ent[2 * lane + 1] = ransw_ent(f1, c0 + f0);
}

// Histogram -> 12-bit frequencies exactly as `normalize` / sicl_or_normalize do it (same floor, same "largest first, lowest index
// on ties" correction walk), by ONE wave: lane l owns symbols 2l, 2l+1 (hh = their counts).  Returns the error bits (2 = no valid table).
__device__ __forceinline__ uint32_t normalize_wave(const uint32_t (&hh)[2], uint32_t n, uint32_t (&f)[2], int lane)
{
    uint32_t err = 0;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t h = hh[k];
        unsigned long long v = (h && n) ? ((unsigned long long)h * 4096u) / n : 0;
        if (h && v == 0) v = 1;
        f[k] = (uint32_t)v;
    }
    int sum = (int)(f[0] + f[1]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    int diff = n ? 4096 - sum : 0;
    for (int it = 0; it < 200 && diff != 0; it++) {
        // candidate of this lane: f > 0 and (diff > 0 or f > 1); key = (f << 8) | (255 - index): max key = largest f, lowest index
        uint32_t key = 0;
#pragma unroll
        for (int k = 0; k < 2; k++)
            if (f[k] > 0 && (diff > 0 || f[k] > 1)) key = max(key, (f[k] << 8) | (uint32_t)(255 - (2 * lane + k)));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, d));
        if (key == 0) { err |= 2; break; }
        const int best = 255 - (int)(key & 255u), fb = (int)(key >> 8);
        const int step = diff > 0 ? diff : (diff < 1 - fb ? 1 - fb : diff);
        if ((best >> 1) == lane) f[best & 1] = (uint32_t)(fb + step);
        diff -= step;
    }
    if (diff != 0) err |= 2;
    return err;
}

// container header (dwords 0..11 but 10 = payload bytes, written by the scan / compaction stage) + frequency table, by one wave
__device__ __forceinline__ void write_header_wave(uint8_t *out, const uint32_t (&f)[2], unsigned long long s1, unsigned long long s2,
                                                  uint32_t n, uint32_t ns, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c, uint32_t img_w,
                                                  uint32_t img_h, uint32_t wss, int lane)
{
    uint8_t *ft = out + SICN_CODEC_HEADER_BYTES + 4 * lane;
    ft[0] = (uint8_t)f[0]; ft[1] = (uint8_t)(f[0] >> 8); ft[2] = (uint8_t)f[1]; ft[3] = (uint8_t)(f[1] >> 8);
    if (lane < 12) {
        const uint32_t a = (uint32_t)((1 + s1) % ADLER_MOD), b = (uint32_t)((n % ADLER_MOD + s2) % ADLER_MOD);
        const uint32_t words[12] = {0x4C434953u /* "SICL" */, 1u | ((uint32_t)SICN_CODEC_RANSW << 16), img_w, img_h, lat_w, lat_h,
                                    lat_c, n, ns, wss, 0u, (b << 16) | a};
        if (lane != 10) {
            const uint32_t v = words[lane];
            uint8_t *p = out + 4 * lane;
            p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
        }
    }
}

// what the asynchronous encoder's streams need to make the frequency table THEMSELVES from the statistics rows (row mode:
// one launch less — the header kernel was 8.5 us of a 1080p encode, nearly all of it launch latency): every wave sums the rows
// and runs the (deterministic) normalisation; stream 0's wave also writes header, table and status
struct EncSelfHeader {
    const uint32_t *rows;     // [n_rows][STAT_ROW_WORDS] per image (stride s_ws), or nullptr: the table comes from freq_g
    uint32_t n_rows;
    uint8_t *out;             // containers (stride s_slot)
    size_t s_slot;
    uint32_t *status;         // sicn_codec_status[n_images]
    uint32_t lat_w, lat_h, lat_c, img_w, img_h;
};

__global__ __launch_bounds__(64) void k_ransw_encode(const uint8_t *__restrict__ lat_, uint32_t n, uint32_t ns,
                                                     const uint16_t *__restrict__ freq_g_, uint8_t *__restrict__ scratch_,
                                                     uint32_t *__restrict__ lens_, size_t s_lat, size_t s_ws, uint32_t wss,
                                                     EncSelfHeader sh = EncSelfHeader{})
{
    const uint8_t *lat = img_ptr(lat_, s_lat);
    uint8_t *scratch = img_ptr(scratch_, s_ws);
    uint32_t *lens = img_ptr(lens_, s_ws);
    __shared__ __attribute__((aligned(16))) RanswEnt ent[128];
    __shared__ __attribute__((aligned(16))) uint16_t words[RING_WORDS];
    const uint32_t st = blockIdx.x, lane = threadIdx.x;
    if (sh.rows) {
        const uint32_t *rows = img_ptr(sh.rows, s_ws);
        uint32_t hh[2] = {0, 0}, hi = 0;
#pragma unroll 8
        for (uint32_t r = 0; r < sh.n_rows; r++) {
            const uint32_t *row = rows + (size_t)r * STAT_ROW_WORDS;
            const uint2 two = *reinterpret_cast<const uint2 *>(row + 2 * lane);
            hh[0] += two.x;
            hh[1] += two.y;
            hi |= row[128 + lane] | row[192 + lane];
        }
        uint32_t f[2];
        uint32_t err = normalize_wave(hh, n, f, (int)lane) | (hi ? 1u : 0u);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) err |= (uint32_t)__shfl_xor((int)err, d);
        if (err) f[0] = f[1] = 0;   // zero-frequency symbols are skipped below: the kernel stays memory-safe
        ransw_build_ent(ent, f[0], f[1], (int)lane);
        if (st == 0) {              // header, table and encoder status of this image: once
            unsigned long long s1 = 0, s2 = 0;
            for (uint32_t r = lane; r < sh.n_rows; r += 64) {
                const unsigned long long *q = (const unsigned long long *)(rows + (size_t)r * STAT_ROW_WORDS + 256);
                s1 += q[0];
                s2 += q[1];
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                s1 += __shfl_xor(s1, d);
                s2 += __shfl_xor(s2, d);
            }
            write_header_wave(img_ptr(sh.out, sh.s_slot), f, s1, s2, n, ns, sh.lat_w, sh.lat_h, sh.lat_c, sh.img_w, sh.img_h, wss, (int)lane);
            if (lane == 0) sh.status[2 * blockIdx.y] = err;
        }
    } else {
        const uint16_t *freq_g = img_ptr(freq_g_, s_ws);
        ransw_build_ent(ent, freq_g[2 * lane], freq_g[2 * lane + 1], (int)lane);
    }
    __syncthreads();
    const uint32_t wcap = wstream_cap(wss);
    const uint32_t begin = st * wss, cnt = min(wss, n - begin), blocks = (cnt + 255) / 256;
    const bool aligned = (reinterpret_cast<uintptr_t>(lat) & 3) == 0;   // begin is a multiple of the stream length (>= 1024)
    uint16_t *dst = (uint16_t *)(scratch + (size_t)st * wcap);
    uint32_t pos = wcap / 2, top = wcap / 2;   // word indices inside the scratch slot, the same in every lane: [pos, top) is in the ring
    uint32_t x = RANSW_L;
    const unsigned long long below = (1ull << lane) - 1;
    auto load4 = [&](uint32_t q) -> uint32_t {   // the lane's 4 symbols of block q (missing ones read as 0)
        const uint32_t j = q * 256 + lane * 4;
        if (aligned && j + 4 <= cnt) return *reinterpret_cast<const uint32_t *>(lat + begin + j);
        uint32_t v = 0;
        for (int k = 0; k < 4; k++)
            if (j + k < cnt) v |= (uint32_t)lat[begin + j + k] << (8 * k);
        return v;
    };
    const uint4 *ent4 = reinterpret_cast<const uint4 *>(ent);
    // software pipeline, two blocks deep: symbols of block q - 2 in flight from memory, table entries of block q - 1 in flight from
    // LDS, block q's chain on registers (symbols >= 128 are an error the statistics stage flags; they are masked here)
    uint32_t s_next = blocks > 1 ? load4(blocks - 2) : 0;
    uint4 e_cur[4];
    {
        const uint32_t s_cur = blocks ? load4(blocks - 1) : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) e_cur[k] = ent4[(s_cur >> (8 * k)) & 127u];
    }
    for (uint32_t q = blocks; q-- > 0;) {
        const uint32_t s_next2 = q >= 2 ? load4(q - 2) : 0;
        uint4 e_next[4];
#pragma unroll
        for (int k = 0; k < 4; k++) e_next[k] = ent4[(s_next >> (8 * k)) & 127u];
        const int have = (int)min(4u, cnt - min(cnt, q * 256 + lane * 4));   // symbols of this lane in this block (4 but in a short last one)
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const uint32_t f = e_cur[k].x & 0xFFFFu, c = e_cur[k].x >> 16;
            const bool active = k < have && f != 0;          // f == 0 only for a latent the header stage rejected
            const bool emit = active && x > e_cur[k].z;      // x >= f * 2^20
            const unsigned long long mask = __ballot(emit);
            pos -= (uint32_t)__popcll(mask);
            if (emit) words[(pos + (uint32_t)__popcll(mask & below)) & (RING_WORDS - 1)] = (uint16_t)x;   // ascending lane order inside the step
            x = emit ? x >> 16 : x;
            uint32_t qq = __umulhi(x, e_cur[k].y);           // floor(x / f) or one less (ransw_div)
            uint32_t r = x - qq * f;
            const bool fix = r >= f;
            qq += fix ? 1u : 0u;
            r -= fix ? f : 0u;
            x = active ? (qq << PROB_BITS) + r + c : x;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) e_cur[k] = e_next[k];
        s_next = s_next2;
        if (top - pos > RING_WORDS - 4 * 64 - 128) {   // the next 4 steps (and the final states) must still fit
            __syncthreads();
            ring_flush(words, dst, pos, top, lane);
            __syncthreads();
            top = pos;
        }
    }
    pos -= 128;   // the 64 final states, lane 0 first (the word index may be odd: two halves)
    words[(pos + 2 * lane) & (RING_WORDS - 1)] = (uint16_t)x;
    words[(pos + 2 * lane + 1) & (RING_WORDS - 1)] = (uint16_t)(x >> 16);
    __syncthreads();
    ring_flush(words, dst, pos, top, lane);
    if (lane == 0) lens[st] = (wcap / 2 - pos) * 2;
}

// BIGTAB (the latency form, taken when an image's streams are few enough that occupancy does not matter): ONE 4096-entry dword
// table v -> symbol | freq << 7 | (v - cum) << 20 instead of the byte table v -> symbol followed by the symbol's (freq, cum): a
// decode step's chain is table -> multiply-add -> renormalisation word, and the second dependent LDS round trip is gone.  16 KB of
// LDS more per wave (5 waves per CU instead of 12), so large batches keep the two-table form, whose latency the other waves hide.
template <bool BIGTAB>
__global__ __launch_bounds__(64) void k_ransw_decode(const uint8_t *__restrict__ payload_, const uint8_t *__restrict__ freq_bytes_,
                                                     const uint32_t *__restrict__ offsets_, uint32_t n, uint32_t ns,
                                                     uint8_t *__restrict__ lat_, uint32_t *__restrict__ err_, size_t s_slot,
                                                     size_t s_ws, size_t s_lat, const uint8_t *__restrict__ payload_bytes_field_,
                                                     const uint32_t *__restrict__ meta_, unsigned long long *__restrict__ sums_, uint32_t wss,
                                                     const uint32_t *__restrict__ self_valid_ = nullptr, uint32_t self_valid_stride = 0,
                                                     int self = 0)
{
    // self != 0 (the asynchronous path up to SELF_SCAN_MAX streams): NO parse and NO scan kernel ran before this one — the
    // wave bounds the slot itself (valid bytes, header payload field), sums the length table up to its own stream, and reports
    // through per-stream words (err_[st], sums_[2 st .. 2 st + 1], plain stores: nothing has to be cleared beforehand);
    // k_dec_finish_self then validates header and table and folds the per-stream words into the status.
    const uint32_t st = blockIdx.x, lane = threadIdx.x;
    const uint8_t *payload = img_ptr(payload_, s_slot), *freq_bytes = img_ptr(freq_bytes_, s_slot);
    // header field "payload bytes" of this image's container (the host has checked it against the bytes it was
    // given): no stream may reach beyond it, whatever the untrusted length table says
    const uint8_t *pbf = img_ptr(payload_bytes_field_, s_slot);
    uint32_t *err = img_ptr(err_, s_ws);
    if (self) {
        err += st;
        const size_t fixed = SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns;
        const uint32_t valid = self_valid_ ? min(self_valid_[(size_t)blockIdx.y * self_valid_stride], (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu))
                                           : (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu);
        if (valid < fixed) {   // nothing of this slot may be read
            if (lane == 0) *err = 1u;
            return;
        }
    }
    uint32_t payload_bytes = pbf[0] | ((uint32_t)pbf[1] << 8) | ((uint32_t)pbf[2] << 16) | ((uint32_t)pbf[3] << 24);
    if (meta_) payload_bytes = min(payload_bytes, img_ptr(meta_, s_ws)[1]);   // async path: clamped by the parse stage
    if (self) {
        const size_t fixed = SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns;
        const uint32_t valid = self_valid_ ? min(self_valid_[(size_t)blockIdx.y * self_valid_stride], (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu))
                                           : (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu);
        if ((size_t)payload_bytes > (size_t)valid - fixed) payload_bytes = 0;   // what k_dec_parse's meta[1] says
    }
    const uint32_t *offsets = img_ptr(offsets_, s_ws);
    uint8_t *lat = img_ptr(lat_, s_lat);
    __shared__ RanswTab tab;
    __shared__ uint16_t freq[128];
    __shared__ __attribute__((aligned(16))) uint8_t slot[4096];
    __shared__ __attribute__((aligned(16))) uint16_t words[RING_WORDS];
    __shared__ __attribute__((aligned(16))) uint32_t big[BIGTAB ? 4096 : 4];
    freq[2 * lane] = (uint16_t)(freq_bytes[4 * lane] | (freq_bytes[4 * lane + 1] << 8));
    freq[2 * lane + 1] = (uint16_t)(freq_bytes[4 * lane + 2] | (freq_bytes[4 * lane + 3] << 8));
    __syncthreads();
    ransw_build(tab, freq, (int)lane);
    __syncthreads();
    for (int k = 0; k < 2; k++) {   // slot[c .. c + f) = symbol: 16 bytes per store in the middle (one symbol of a ReLU latent owns
        const uint32_t t = tab.fc[2 * lane + k], f = t & 0xFFFFu, c = t >> 16;   // half the table: 2048 byte stores by one lane
        const uint32_t e = min(c + f, 4096u), sy = (uint32_t)(2 * lane + k);     // were a third of a short stream's decode)
        uint32_t v = c;
        for (; v < e && (v & 15u); v++) slot[v] = (uint8_t)sy;
        const uint32_t pat = sy * 0x01010101u;
        for (; v + 16 <= e; v += 16) *reinterpret_cast<uint4 *>(slot + v) = make_uint4(pat, pat, pat, pat);
        for (; v < e; v++) slot[v] = (uint8_t)sy;
    }
    if constexpr (BIGTAB) {   // expand: entry v = symbol | freq << 7 | (v - cum) << 20 (7 + 13 + 12 bits), four consecutive v per lane and round
        __syncthreads();
#pragma unroll 4
        for (uint32_t v0 = 4 * lane; v0 < 4096; v0 += 256) {
            const uint32_t s4 = *reinterpret_cast<const uint32_t *>(slot + v0);
            uint4 o;
            uint32_t *ov = reinterpret_cast<uint32_t *>(&o);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t sy = (s4 >> (8 * k)) & 255u, t = tab.fc[sy & 127u];
                ov[k] = sy | ((t & 0xFFFFu) << 7) | ((v0 + k - (t >> 16)) << 20);
            }
            *reinterpret_cast<uint4 *>(big + v0) = o;
        }
    }
    const uint32_t begin = st * wss, cnt = min(wss, n - begin), blocks = (cnt + 255) / 256;
    uint32_t off, len;
    if (self) {   // exclusive prefix sum of the length table up to this stream, entries above the cap counting as 0 (as in k_scan)
        const uint8_t *table = freq_bytes + 256;
        const uint32_t cap = wstream_cap(wss);
        auto entry = [&](uint32_t i) {
            const uint8_t *q = table + 4 * (size_t)i;
            const uint32_t v = q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
            return v > cap ? 0u : v;
        };
        uint32_t sum = 0;
        if ((reinterpret_cast<uintptr_t>(table) & 3) == 0) {   // the usual case (slots at 4-byte multiples): one load per entry
            const uint32_t *t32 = reinterpret_cast<const uint32_t *>(table);
#pragma unroll 8
            for (uint32_t i = lane; i < st; i += 64) {
                const uint32_t v = t32[i];
                sum += v > cap ? 0u : v;
            }
        } else
            for (uint32_t i = lane; i < st; i += 64) sum += entry(i);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
        off = sum;
        len = entry(st);
    } else {
        off = offsets[st];
        len = offsets[st + 1] - off;
    }
    if (len < 256 || (len & 1) || (off & 1) || len > wstream_cap(wss) ||
        (unsigned long long)off + len > payload_bytes) {   // streams start at even container offsets
        if (lane == 0) {
            if (self) *err = 1u; else atomicOr(err, 1u);
        }
        return;
    }
    const uint32_t nwords = len / 2;
    const uint16_t *src = (const uint16_t *)(payload + off);
    uint32_t loaded = ring_fill(words, src, 0, min(nwords, RING_WORDS), lane);
    __syncthreads();
    uint32_t x = words[2 * lane] | ((uint32_t)words[2 * lane + 1] << 16);
    uint32_t wpos = 128;
    const unsigned long long below = (1ull << lane) - 1;
    const bool aligned = (reinterpret_cast<uintptr_t>(lat) & 3) == 0;
    bool bad = false;
    // sums_ (async path): the two sums adler32 is made of, taken from the symbols as they are decoded — the separate
    // statistics pass over the decoded latent (17 us on a 1080p latent, a tenth of a small image's whole decode) goes away
    unsigned long long s1 = 0, s2 = 0;
    for (uint32_t q = 0; q < blocks; q++) {
        if (loaded < nwords && loaded - min(wpos, loaded) < 4 * 64) {   // top the ring up: the 4 steps below read <= 256 words
            __syncthreads();
            loaded = ring_fill(words, src, loaded, min(nwords, wpos + RING_WORDS), lane);
            __syncthreads();
        }
        uint32_t out4 = 0;
        const int have = (int)min(4u, cnt - min(cnt, q * 256 + lane * 4));   // symbols of this lane in this block
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool active = k < have;
            uint32_t sy, xn;
            if constexpr (BIGTAB) {
                const uint32_t e = big[x & 4095u];
                sy = e & 127u;
                xn = ((e >> 7) & 0x1FFFu) * (x >> PROB_BITS) + (e >> 20);
            } else {
                const uint32_t v = x & 4095u;
                sy = slot[v];
                const uint32_t t = tab.fc[sy];
                xn = (t & 0xFFFFu) * (x >> PROB_BITS) + v - (t >> 16);
            }
            sy = active ? sy : 0u;
            x = active ? xn : x;
            out4 |= sy << (8 * k);
            s1 += sy;
            s2 += (unsigned long long)(n - (begin + q * 256 + lane * 4 + k)) * sy;   // < 2^38 per term, 1024 terms per lane (sy = 0 when inactive)
            const bool need = active && x < RANSW_L;
            const unsigned long long mask = __ballot(need);
            const uint32_t idx = wpos + (uint32_t)__popcll(mask & below);
            const uint32_t wd = words[idx & (RING_WORDS - 1)];   // read unconditionally: no branch on the chain
            bad = bad || (need && idx >= loaded);                // loaded <= nwords; a stream that runs dry is malformed
            x = (need && idx < loaded) ? (x << 16) | wd : x;
            wpos += (uint32_t)__popcll(mask);
        }
        const uint32_t j = q * 256 + lane * 4;
        if (aligned && j + 4 <= cnt)
            *reinterpret_cast<uint32_t *>(lat + begin + j) = out4;
        else
            for (int k = 0; k < 4; k++)
                if (j + k < cnt) lat[begin + j + k] = (uint8_t)(out4 >> (8 * k));
    }
    if (self) {   // this stream's verdict, one plain store
        const unsigned long long any_bad = __ballot(bad || x != RANSW_L || wpos != nwords);
        if (lane == 0) *err = any_bad ? 1u : 0u;
    } else if (bad || x != RANSW_L || wpos != nwords)
        atomicOr(err, 1u);
    if (sums_) {
        unsigned long long *sums = img_ptr(sums_, s_ws);
        s2 %= ADLER_MOD;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            s1 += __shfl_down(s1, d);
            s2 += __shfl_down(s2, d);
        }
        if (lane == 0) {
            if (self) {   // per-stream partial sums: a failed stream leaves garbage here, but then the checksum is never looked at
                sums[2 * st] = s1;
                sums[2 * st + 1] = s2 % ADLER_MOD;
            } else {
                if (s1) atomicAdd(&sums[0], s1);
                if (s2) atomicAdd(&sums[1], s2 % ADLER_MOD);
            }
        }
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
// Entries above `cap` (only possible in an untrusted container) raise `*err` and count as 0, so with
// n * cap < 2^32 (the n_symbols limit of the rANS modes, MAX_RANS_SYMBOLS) the 32-bit sums cannot wrap.
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *__restrict__ in_, const uint8_t *__restrict__ in_bytes_,
                                               uint32_t n, uint32_t *__restrict__ out_, uint8_t *__restrict__ table_out_,
                                               uint8_t *__restrict__ total_out_, size_t s_ws, size_t s_slot, uint32_t cap,
                                               uint32_t *__restrict__ err_, uint32_t *__restrict__ status_bytes_ = nullptr,
                                               uint32_t fixed_bytes = 0, const uint32_t *__restrict__ skip_meta_ = nullptr)
{
    // async decode: a slot the parse stage found shorter than its own fixed part (meta[0] & 0x100) is never read
    const bool skip = skip_meta_ && (img_ptr(skip_meta_, s_ws)[0] & 0x100u);
    const uint32_t *in = in_ ? img_ptr(in_, s_ws) : nullptr;
    const uint8_t *in_bytes = in_bytes_ ? img_ptr(in_bytes_, s_slot) : nullptr;
    uint32_t *out = img_ptr(out_, s_ws);
    uint8_t *table_out = table_out_ ? img_ptr(table_out_, s_slot) : nullptr;
    uint8_t *total_out = total_out_ ? img_ptr(total_out_, s_slot) : nullptr;
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        uint32_t v = 0;
        if (i < n && !skip) {
            if (in_bytes) {
                const uint8_t *p = in_bytes + 4 * (size_t)i;
                v = p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            } else
                v = in[i];
            if (v > cap) {
                if (err_) atomicOr(img_ptr(err_, s_ws), 1u);
                v = 0;
            }
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
        if (status_bytes_) status_bytes_[2 * blockIdx.y + 1] = fixed_bytes + tot;   // sicn_codec_status.bytes
    }
}

// one workgroup per stream: scratch tail -> payload
__global__ __launch_bounds__(256) void k_compact(const uint8_t *__restrict__ scratch_, const uint32_t *__restrict__ lens_,
                                                 const uint32_t *__restrict__ offsets_, uint8_t *__restrict__ payload_, uint32_t cap,
                                                 size_t s_ws, size_t s_slot)
{
    const uint8_t *scratch = img_ptr(scratch_, s_ws);
    const uint32_t *lens = img_ptr(lens_, s_ws), *offsets = img_ptr(offsets_, s_ws);
    uint8_t *payload = img_ptr(payload_, s_slot);
    const uint32_t st = blockIdx.x, len = lens[st];
    const uint8_t *src = scratch + (size_t)st * cap + (cap - len);
    uint8_t *dst = payload + offsets[st];
    for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i];
}

// The same without a scan kernel in front (asynchronous encoder, up to SELF_SCAN_MAX streams per image): every workgroup sums
// the lengths of the streams before its own, copies its stream and writes its entry of the container's length table; the last
// one also writes the header's payload-bytes field and the status' byte count (what k_scan did besides the offsets).
__global__ __launch_bounds__(256) void k_compact_self(const uint8_t *__restrict__ scratch_, const uint32_t *__restrict__ lens_,
                                                      uint8_t *__restrict__ out_, uint32_t cap, uint32_t ns, size_t s_ws, size_t s_slot,
                                                      uint32_t *__restrict__ status_bytes_, uint32_t fixed_bytes)
{
    const uint8_t *scratch = img_ptr(scratch_, s_ws);
    const uint32_t *lens = img_ptr(lens_, s_ws);
    uint8_t *out = img_ptr(out_, s_slot);
    uint8_t *table = out + SICN_CODEC_HEADER_BYTES + 256, *payload = table + 4 * (size_t)ns;
    const uint32_t st = blockIdx.x, len = lens[st];
    __shared__ uint32_t part[4];
    uint32_t sum = 0;
    for (uint32_t i = threadIdx.x; i < st; i += 256) sum += lens[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    const uint32_t off = part[0] + part[1] + part[2] + part[3];
    const uint8_t *src = scratch + (size_t)st * cap + (cap - len);
    uint8_t *dst = payload + off;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 1) == 0) {   // lengths and offsets are even: 2 bytes at least
        const uint16_t *s2 = reinterpret_cast<const uint16_t *>(src);
        uint16_t *d2 = reinterpret_cast<uint16_t *>(dst);
        for (uint32_t i = threadIdx.x; i < len / 2; i += 256) d2[i] = s2[i];
        if ((len & 1) && threadIdx.x == 0) dst[len - 1] = src[len - 1];
    } else
        for (uint32_t i = threadIdx.x; i < len; i += 256) dst[i] = src[i];
    if (threadIdx.x == 0) {
        uint8_t *p = table + 4 * (size_t)st;
        p[0] = (uint8_t)len; p[1] = (uint8_t)(len >> 8); p[2] = (uint8_t)(len >> 16); p[3] = (uint8_t)(len >> 24);
        if (st == ns - 1) {
            const uint32_t tot = off + len;
            out[40] = (uint8_t)tot; out[41] = (uint8_t)(tot >> 8); out[42] = (uint8_t)(tot >> 16); out[43] = (uint8_t)(tot >> 24);
            status_bytes_[2 * blockIdx.y + 1] = fixed_bytes + tot;   // sicn_codec_status.bytes
        }
    }
}

__global__ __launch_bounds__(256) void k_rans_decode(const uint8_t *__restrict__ payload, const uint8_t *__restrict__ freq_bytes,
                                                     const uint32_t *__restrict__ offsets, uint32_t n, uint32_t ns,
                                                     uint8_t *__restrict__ lat, uint32_t *__restrict__ err, uint32_t payload_bytes)
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
    if (offsets[st + 1] > payload_bytes || offsets[st + 1] < offsets[st]) { atomicOr(err, 1u); return; }
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


// ---- asynchronous batch path: everything the host did between two synchronisations, on the device -----------
// blockIdx.y = image everywhere; per-image workspace stride s_ws, container slot stride s_slot.

constexpr uint32_t STATS_WORDS = (1024 + 64 + 256) / 4;   // hist[256] + sums (64 B) + freq (256 B), then meta (64 B): contiguous
__global__ __launch_bounds__(64) void k_clear_stats(uint32_t *__restrict__ hist_, size_t s_ws)
{
    uint32_t *h = img_ptr(hist_, s_ws);
    for (uint32_t i = threadIdx.x; i < STATS_WORDS + 16; i += 64) h[i] = 0;
}
// the decoders' parse kernels are the first thing on the stream that touches the workspace: they clear the block themselves
// (one launch less, about 4.5 us of a small image's decode) — everything but meta[0..3], which lane 0 then writes
__device__ __forceinline__ void clear_stats_in_parse(uint32_t *meta, int lane)
{
    uint32_t *h = meta - STATS_WORDS;
    for (uint32_t i = lane; i < STATS_WORDS + 16; i += 64)
        if (i < STATS_WORDS || i >= STATS_WORDS + 4) h[i] = 0;
}

// Histogram -> 12-bit frequencies exactly as `normalize` / sicl_or_normalize do it (same floor, same "largest first,
// lowest index on ties" correction walk), by one wave: lane l owns symbols 2l, 2l+1.  Then the container header and
// frequency table.  status[0] = error flags (bit 0: symbol >= 128, bit 1: normalisation failed).
__global__ __launch_bounds__(256) void k_enc_header(const uint32_t *__restrict__ hist_, const unsigned long long *__restrict__ sums_,
                                                   uint16_t *__restrict__ freq_, uint8_t *__restrict__ out_, uint32_t *__restrict__ status_,
                                                   uint32_t n, uint32_t ns, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                                                   uint32_t img_w, uint32_t img_h, size_t s_ws, size_t s_slot, uint32_t wss,
                                                   const uint32_t *__restrict__ rows_ = nullptr, uint32_t n_rows = 0)
{
    const uint32_t *hist = img_ptr(hist_, s_ws);
    const unsigned long long *sums = img_ptr(sums_, s_ws);
    uint16_t *freq = img_ptr(freq_, s_ws);
    uint8_t *out = img_ptr(out_, s_slot);
    uint32_t *status = status_ + 2 * blockIdx.y;
    const int lane = threadIdx.x;
    uint32_t err = 0;
    uint32_t hh[2], hi = 0;
    unsigned long long s1 = 0, s2 = 0;
    if (rows_) {   // row mode (launched with 4 waves): the statistics workgroups' rows are summed here, wave w = rows w, w + 4, ..
        const uint32_t *rows = img_ptr(rows_, s_ws);
        __shared__ uint32_t part[4][260];
        const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        unsigned long long p1 = 0, p2 = 0;
#pragma unroll 4
        for (uint32_t r = wv; r < n_rows; r += 4) {
            const uint32_t *row = rows + (size_t)r * STAT_ROW_WORDS;
            a0 += row[l];
            a1 += row[64 + l];
            a2 += row[128 + l];
            a3 += row[192 + l];
            if (l == 0) {
                const unsigned long long *q = (const unsigned long long *)(row + 256);
                p1 += q[0];
                p2 += q[1];
            }
        }
        part[wv][l] = a0; part[wv][64 + l] = a1; part[wv][128 + l] = a2; part[wv][192 + l] = a3;
        if (l == 0) {
            part[wv][256] = (uint32_t)p1; part[wv][257] = (uint32_t)(p1 >> 32);
            part[wv][258] = (uint32_t)p2; part[wv][259] = (uint32_t)(p2 >> 32);
        }
        __syncthreads();
        if (wv) return;
        auto tot = [&](int i) { return part[0][i] + part[1][i] + part[2][i] + part[3][i]; };
        hh[0] = tot(2 * lane);
        hh[1] = tot(2 * lane + 1);
        hi = tot(128 + lane) | tot(192 + lane);
        for (int k = 0; k < 4; k++) {
            s1 += part[k][256] | ((unsigned long long)part[k][257] << 32);
            s2 += part[k][258] | ((unsigned long long)part[k][259] << 32);
        }
    } else {
        hh[0] = hist[2 * lane];
        hh[1] = hist[2 * lane + 1];
        hi = hist[128 + lane] | hist[192 + lane];
        s1 = sums[0];
        s2 = sums[1];
    }
    if (hi) err = 1;
    uint32_t f[2];
    err |= normalize_wave(hh, n, f, lane);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) err |= (uint32_t)__shfl_xor((int)err, d);
    if (err) f[0] = f[1] = 0;   // the encode kernel skips zero-frequency symbols: it stays memory-safe
    freq[2 * lane] = (uint16_t)f[0];
    freq[2 * lane + 1] = (uint16_t)f[1];
    write_header_wave(out, f, s1, s2, n, ns, lat_w, lat_h, lat_c, img_w, img_h, wss, lane);
    if (lane == 0) status[0] = err;
}

// Decoder front end: every header field against the shape the caller expects, the frequency table, and the sizes
// against the bytes the caller vouches for.  meta[0] |= error bits, meta[1] = payload bytes the streams may use.
__global__ __launch_bounds__(64) void k_dec_parse(const uint8_t *__restrict__ containers_, const uint32_t *__restrict__ valid_bytes_,
                                                  uint32_t valid_stride, uint32_t *__restrict__ meta_, uint32_t n, uint32_t ns,
                                                  uint32_t lat_w, uint32_t lat_h, uint32_t lat_c, size_t s_slot, size_t s_ws, uint32_t wss)
{
    const uint8_t *c = img_ptr(containers_, s_slot);
    uint32_t *meta = img_ptr(meta_, s_ws);
    const int lane = threadIdx.x;
    const uint32_t valid = valid_bytes_ ? min(valid_bytes_[(size_t)blockIdx.y * valid_stride], (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu))
                                        : (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu);
    const size_t fixed = SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns;
    uint32_t err = 0;
    clear_stats_in_parse(meta, lane);
    if (valid < fixed) {   // nothing of this slot may be read
        if (lane == 0) { meta[0] = 0x104u; meta[1] = 0; meta[2] = 0; meta[3] = 0; }
        return;
    }
    auto rd32 = [&](int o) { return c[o] | ((uint32_t)c[o + 1] << 8) | ((uint32_t)c[o + 2] << 16) | ((uint32_t)c[o + 3] << 24); };
    const uint32_t expect[10] = {0x4C434953u, 1u | ((uint32_t)SICN_CODEC_RANSW << 16), 0, 0, lat_w, lat_h, lat_c, n, ns, wss};
    if (lane < 10 && lane != 2 && lane != 3 && rd32(4 * lane) != expect[lane]) err = 4;
    uint32_t fsum = c[SICN_CODEC_HEADER_BYTES + 4 * lane] + ((uint32_t)c[SICN_CODEC_HEADER_BYTES + 4 * lane + 1] << 8) +
                    c[SICN_CODEC_HEADER_BYTES + 4 * lane + 2] + ((uint32_t)c[SICN_CODEC_HEADER_BYTES + 4 * lane + 3] << 8);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) fsum += __shfl_xor((int)fsum, d);
    if (n && fsum != 4096) err |= 8;
    const uint32_t pb = rd32(40);
    if ((size_t)pb > (size_t)valid - fixed) err |= 16;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) err |= (uint32_t)__shfl_xor((int)err, d);
    if (lane == 0) {
        meta[0] = err;
        meta[1] = (err & 16) ? 0u : pb;
        meta[2] = rd32(44);
        meta[3] = 0;
    }
}

// Decoder back end: stream-level errors, table total, checksum -> status {error, n_symbols}.
__global__ __launch_bounds__(64) void k_dec_finish(const uint32_t *__restrict__ meta_, const uint32_t *__restrict__ hist_,
                                                   const unsigned long long *__restrict__ sums_, const uint32_t *__restrict__ offsets_,
                                                   uint32_t *__restrict__ status_, uint32_t n, uint32_t ns, size_t s_ws)
{
    if (threadIdx.x) return;
    const uint32_t *meta = img_ptr(meta_, s_ws), *offsets = img_ptr(offsets_, s_ws);
    (void)hist_;
    const unsigned long long *sums = img_ptr(sums_, s_ws);
    uint32_t err = meta[0];
    if (meta[3]) err |= 32;                          // a stream overran / underran (k_ransw_decode, k_scan)
    if (offsets[ns] != meta[1]) err |= 64;           // the length table does not add up to the payload
    const uint32_t a = (uint32_t)((1 + sums[0]) % ADLER_MOD), b = (uint32_t)((n % ADLER_MOD + sums[1]) % ADLER_MOD);
    if (!err && ((b << 16) | a) != meta[2]) err |= 128;   // checksum (SICN_EBADMSG)
    status_[2 * blockIdx.y] = err;
    status_[2 * blockIdx.y + 1] = n;
}

// The asynchronous decoder's ONLY other kernel when the streams were decoded in self mode (k_ransw_decode, self != 0): header,
// frequency table, payload size and length table validated here, AFTER the streams ran (they bound themselves), per-stream
// verdicts and checksum sums folded -> status {error, n_symbols}; the same error bits as k_dec_parse + k_scan + k_dec_finish.
__global__ __launch_bounds__(256) void k_dec_finish_self(const uint8_t *__restrict__ containers_, const uint32_t *__restrict__ valid_bytes_,
                                                         uint32_t valid_stride, const uint32_t *__restrict__ serr_,
                                                         const unsigned long long *__restrict__ ssum_, uint32_t *__restrict__ status_,
                                                         uint32_t n, uint32_t ns, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                                                         size_t s_slot, size_t s_ws, uint32_t wss)
{
    const uint8_t *c = img_ptr(containers_, s_slot);
    const uint32_t *serr = img_ptr(serr_, s_ws);
    const unsigned long long *ssum = img_ptr(ssum_, s_ws);
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t valid = valid_bytes_ ? min(valid_bytes_[(size_t)blockIdx.y * valid_stride], (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu))
                                        : (uint32_t)min(s_slot, (size_t)0xFFFFFFFFu);
    const size_t fixed = SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns;
    __shared__ uint32_t r_err, r_tab;
    __shared__ unsigned long long r_s1, r_s2;
    if (tid == 0) { r_err = 0; r_tab = 0; r_s1 = 0; r_s2 = 0; }
    __syncthreads();
    if (valid < fixed) {   // nothing of this slot was read: bit 8 + bit 2, and bit 5 for the streams that all refused it
        if (tid == 0) {
            status_[2 * blockIdx.y] = 0x104u | (ns ? 32u : 0u);
            status_[2 * blockIdx.y + 1] = n;
        }
        return;
    }
    auto rd32 = [&](size_t o) { return c[o] | ((uint32_t)c[o + 1] << 8) | ((uint32_t)c[o + 2] << 16) | ((uint32_t)c[o + 3] << 24); };
    uint32_t err = 0;
    if (tid < 64) {   // wave 0: the checks of k_dec_parse
        const uint32_t expect[10] = {0x4C434953u, 1u | ((uint32_t)SICN_CODEC_RANSW << 16), 0, 0, lat_w, lat_h, lat_c, n, ns, wss};
        if (lane < 10 && lane != 2 && lane != 3 && rd32(4 * lane) != expect[lane]) err = 4;
        uint32_t fsum = c[SICN_CODEC_HEADER_BYTES + 4 * lane] + ((uint32_t)c[SICN_CODEC_HEADER_BYTES + 4 * lane + 1] << 8) +
                        c[SICN_CODEC_HEADER_BYTES + 4 * lane + 2] + ((uint32_t)c[SICN_CODEC_HEADER_BYTES + 4 * lane + 3] << 8);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) fsum += __shfl_xor((int)fsum, d);
        if (n && fsum != 4096) err |= 8;
    }
    const uint32_t pb = rd32(40);
    const bool pb_bad = (size_t)pb > (size_t)valid - fixed;
    if (tid == 0 && pb_bad) err |= 16;
    const uint32_t bound = pb_bad ? 0u : pb;
    // length table: entries above the cap are an error and count as 0 (k_scan); the per-stream verdicts; the checksum sums
    const uint32_t cap = wstream_cap(wss);
    uint32_t tab = 0;
    unsigned long long s1 = 0, s2 = 0;
    for (uint32_t i = tid; i < ns; i += 256) {
        uint32_t v = rd32(SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)i);
        if (v > cap) { err |= 32; v = 0; }
        tab += v;
        if (serr[i]) err |= 32;
        else { s1 += ssum[2 * i]; s2 += ssum[2 * i + 1]; }
    }
    s2 %= ADLER_MOD;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        err |= (uint32_t)__shfl_xor((int)err, d);
        tab += (uint32_t)__shfl_xor((int)tab, d);
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    if (lane == 0) {
        if (err) atomicOr(&r_err, err);
        if (tab) atomicAdd(&r_tab, tab);
        if (s1) atomicAdd(&r_s1, s1);
        if (s2) atomicAdd(&r_s2, s2 % ADLER_MOD);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t e = r_err;
        if (r_tab != bound) e |= 64;           // the length table does not add up to the payload
        const uint32_t a = (uint32_t)((1 + r_s1) % ADLER_MOD), b = (uint32_t)((n % ADLER_MOD + r_s2) % ADLER_MOD);
        if (!e && ((b << 16) | a) != rd32(44)) e |= 128;   // checksum (SICN_EBADMSG)
        status_[2 * blockIdx.y] = e;
        status_[2 * blockIdx.y + 1] = n;
    }
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

// Host-side exhaustive check of the encoder's divide (the very function the kernel inlines): for every f in
// [f_begin, f_end) and every state the encoder can hold when it divides (x in [2^16, f << 20)), sampled at all
// multiples of f within +-2 of a stride walk, at both ends of the range and around every power of two.
// Returns the number of (x, f) pairs whose quotient or remainder differs from x / f, x % f.
extern "C" long long sicn_codec_selftest_div(uint32_t f_begin, uint32_t f_end, unsigned long long *n_checked)
{
    long long bad = 0;
    unsigned long long cnt = 0;
    for (uint32_t f = f_begin < 1 ? 1 : f_begin; f < f_end && f <= 4096; f++) {
        const uint32_t m = ransw_rcp(f);
        const unsigned long long hi = (unsigned long long)f << 20;        // exclusive
        auto check = [&](unsigned long long xx) {
            if (xx < 65536 || xx >= hi || xx > 0xFFFFFFFFull) return;
            uint32_t r;
            const uint32_t x = (uint32_t)xx, q = ransw_div(x, f, m, r);
            cnt++;
            if (q != x / f || r != x % f) bad++;
        };
        // every 4099th multiple of f (and its neighbours) covers the range with ~256 k..1 M probes per f
        for (unsigned long long k = 65536 / f; k * f < hi + f; k += 4099)
            for (long long d = -2; d <= 2; d++) check(k * f + (unsigned long long)d);
        for (int b = 16; b <= 32; b++)
            for (long long d = -(long long)f - 2; d <= (long long)f + 2; d++) check((1ull << b) + (unsigned long long)d);
        for (long long d = 0; d <= 2 * (long long)f + 2; d++) { check(65536 + d); check(hi - 1 - d); }
    }
    if (n_checked) *n_checked = cnt;
    return bad;
}

// 32-bit stream offsets: ceil(n / wss) scratch slots of wstream_cap(wss) bytes must stay below 2^32 (the static_assert above is
// the 16384 case; 1024-symbol streams allow 1.9 G symbols)
static bool wstream_fits(unsigned long long n, uint32_t wss)
{
    return n <= MAX_RANS_SYMBOLS && ((n + wss - 1) / wss) * (unsigned long long)wstream_cap(wss) < (1ull << 32);
}

extern "C" size_t sicn_codec_max_bytes_sl(uint32_t n, uint32_t wss)
{
    if (!wstream_ok(wss)) return 0;
    const uint32_t ns = (n + wss - 1) / wss;
    return SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns + 2 * (size_t)n + 256 * (size_t)ns;
}

extern "C" size_t sicn_codec_workspace_bytes_sl(uint32_t n, uint32_t wss)
{
    if (!wstream_ok(wss)) return 0;
    Workspace w;
    return carve(w, nullptr, (n + wss - 1) / wss, wstream_cap(wss)) + 64;
}

extern "C" size_t sicn_codec_batch_workspace_bytes_sl(uint32_t n_symbols, uint32_t n_images, uint32_t wss)
{
    if (!wstream_ok(wss)) return 0;
    return (size_t)n_images * align_up(sicn_codec_workspace_bytes_sl(n_symbols, wss), 256) + align_up(16 * (size_t)n_images, 256);
}

extern "C" size_t sicn_codec_max_bytes(int mode, uint32_t n)
{
    const uint32_t ns = (n + stream_symbols(mode) - 1) / stream_symbols(mode);
    if (mode == SICN_CODEC_RANSW) return SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns + 2 * (size_t)n + 256 * (size_t)ns;
    if (mode == SICN_CODEC_RAW8) return SICN_CODEC_HEADER_BYTES + (size_t)n;
    if (mode == SICN_CODEC_PACKED7) return SICN_CODEC_HEADER_BYTES + ((size_t)n + 7) / 8 * 7;
    if (mode == SICN_CODEC_RANS) return SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns + 2 * (size_t)n + 8 * (size_t)ns;
    return 0;
}

extern "C" size_t sicn_codec_workspace_bytes(int mode, uint32_t n)
{
    Workspace w;
    if (mode == SICN_CODEC_RANSW) {   // enough for a container of ANY admissible stream length (sicn_codec_decode reads it from the header)
        size_t mx = 0;
        for (uint32_t wss = 1024; wss <= WSS; wss <<= 1)
            if (wstream_fits(n, wss)) mx = std::max(mx, carve(w, nullptr, (n + wss - 1) / wss, wstream_cap(wss)) + 64);
        return mx;
    }
    const uint32_t ns = (n + stream_symbols(mode) - 1) / stream_symbols(mode);
    return carve(w, nullptr, ns, mode >= SICN_CODEC_RANS ? stream_cap(mode) : 0) + 64;
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
    info->stream_symbols = get32(h + 36);
    if (info->mode > 4) return SICN_EINVAL;
    if (info->mode == SICN_CODEC_RANSW ? !wstream_ok(info->stream_symbols)
                                       : info->stream_symbols != stream_symbols(info->mode == 4 ? 3 : (int)info->mode))
        return SICN_EINVAL;
    if ((unsigned long long)info->lat_w * info->lat_h * info->lat_c != info->n_symbols) return SICN_EINVAL;
    const uint32_t ss = info->stream_symbols;
    if (info->mode == 4) {   // anchors and non-anchors are cut into streams separately (sicn_codec_ctx.inc)
        const uint32_t W = info->lat_w, H = info->lat_h, C = info->lat_c;
        const unsigned long long na = ((unsigned long long)(H / 2) * W + ((H & 1) ? (W + 1) / 2 : 0)) * C;
        const unsigned long long nn = ((unsigned long long)(H / 2) * W + ((H & 1) ? W / 2 : 0)) * C;
        if (info->n_streams != (na + WSS - 1) / WSS + (nn + WSS - 1) / WSS) return SICN_EINVAL;
        return SICN_OK;
    }
    if (info->n_streams != (info->n_symbols + ss - 1) / ss) return SICN_EINVAL;
    return SICN_OK;
}

extern "C" int sicn_codec_encode(int mode, const uint8_t *latent, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c,
                                 uint32_t img_w, uint32_t img_h, uint8_t *out, size_t out_capacity, size_t *out_bytes,
                                 void *workspace, size_t workspace_bytes, void *hip_stream)
{
    if (!out || !out_bytes || mode < 0 || mode > 3) return SICN_EINVAL;
    const unsigned long long n64 = (unsigned long long)lat_w * lat_h * lat_c;
    if (n64 > 0x7fffffffull || (n64 && !latent)) return SICN_EINVAL;
    if (mode >= SICN_CODEC_RANS && n64 > MAX_RANS_SYMBOLS) return SICN_EINVAL;
    const uint32_t n = (uint32_t)n64, ss = stream_symbols(mode), ns = (n + ss - 1) / ss;
    if (out_capacity < sicn_codec_max_bytes(mode, n)) return SICN_ENOSPC;
    if (!workspace || workspace_bytes < sicn_codec_workspace_bytes(mode, n)) return SICN_ENOSPC;
    hipStream_t stream = (hipStream_t)hip_stream;
    Workspace w;
    carve(w, workspace, ns, mode >= SICN_CODEC_RANS ? stream_cap(mode) : 0);

    // statistics (also the checksum and the symbol-range check) -------------------------------
    HIP_TRY(hipMemsetAsync(w.hist, 0, 1024 + 64, stream));
    if (n)
        hipLaunchKernelGGL(k_stats, dim3(stats_blocks(n, 1)), dim3(256), 0, stream, latent, n, w.hist, w.sums, (size_t)0, (size_t)0);
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
    put32(head + 36, ss);
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
        if (ns && mode == SICN_CODEC_RANSW)
            hipLaunchKernelGGL(k_ransw_encode, dim3(ns), dim3(64), 0, stream, latent, n, ns, w.freq, w.scratch, w.lens, (size_t)0, (size_t)0, WSS);
        else if (ns)
            hipLaunchKernelGGL(k_rans_encode, dim3((ns + 255) / 256), dim3(256), 0, stream, latent, n, ns, w.freq, w.scratch, w.lens);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, w.lens, (const uint8_t *)nullptr, ns, w.offsets, table, out + 40, (size_t)0, (size_t)0,
                           0xFFFFFFFFu, (uint32_t *)nullptr);
        if (ns) hipLaunchKernelGGL(k_compact, dim3(ns), dim3(256), 0, stream, w.scratch, w.lens, w.offsets, payload, stream_cap(mode), (size_t)0, (size_t)0);
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
    if (n > 0x7fffffffu || (info.mode >= SICN_CODEC_RANS && n > MAX_RANS_SYMBOLS)) return SICN_EINVAL;
    if (n && (!latent || latent_capacity < n)) return SICN_ENOSPC;
    const bool wmode = info.mode == SICN_CODEC_RANSW;
    if (wmode && !wstream_fits(n, info.stream_symbols)) return SICN_EINVAL;
    // (a caller that sized the workspace for the default stream length needs sicn_codec_workspace_bytes_sl for shorter streams)
    if (!workspace || workspace_bytes < (wmode ? sicn_codec_workspace_bytes_sl(n, info.stream_symbols)
                                               : sicn_codec_workspace_bytes((int)info.mode, n)))
        return SICN_ENOSPC;
    Workspace w;
    carve(w, workspace, ns, 0);
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
                           (uint8_t *)nullptr, (uint8_t *)nullptr, (size_t)0, (size_t)0,
                           wmode ? wstream_cap(info.stream_symbols) : stream_cap((int)info.mode), err);
        uint32_t total = 0;
        HIP_TRY(hipMemcpyAsync(&total, w.offsets + ns, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (total != info.payload_bytes) return SICN_EINVAL;
        if (ns && info.mode == SICN_CODEC_RANSW)
            hipLaunchKernelGGL(k_ransw_decode<false>, dim3(ns), dim3(64), 0, stream, payload, freq_bytes, w.offsets, n, ns, latent, err, (size_t)0, (size_t)0, (size_t)0,
                               container + 40, (const uint32_t *)nullptr, (unsigned long long *)nullptr, info.stream_symbols);
        else if (ns)
            hipLaunchKernelGGL(k_rans_decode, dim3((ns + 255) / 256), dim3(256), 0, stream, payload, freq_bytes, w.offsets, n, ns, latent, err,
                               info.payload_bytes);
    }
    // checksum of what was decoded
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, err, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (flag) return SICN_EINVAL;
    HIP_TRY(hipMemsetAsync(w.hist, 0, 1024 + 64, stream));
    if (n) hipLaunchKernelGGL(k_stats, dim3(stats_blocks(n, 1)), dim3(256), 0, stream, latent, n, w.hist, w.sums, (size_t)0, (size_t)0);
    unsigned long long sums[2];
    HIP_TRY(hipMemcpyAsync(sums, w.sums, sizeof sums, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (hipGetLastError() != hipSuccess) return SICN_ENODEV;
    return adler_from_sums(sums[0], sums[1], n) == info.adler32 ? SICN_OK : SICN_EBADMSG;
}


// ---- batches -------------------------------------------------------------------------------------------------
// The asynchronous pair does everything on the device (statistics -> frequency table -> header -> streams -> offsets
// -> compaction; parse -> offsets -> streams -> checksum) and reports through a device status array: no host
// synchronisation, no allocation, capturable in a hipGraph.  The synchronous batch calls are thin wrappers that add
// the read-back.
extern "C" size_t sicn_codec_batch_workspace_bytes(int mode, uint32_t n_symbols, uint32_t n_images)
{
    // per-image blocks, then room for the synchronous wrappers' status arrays (2 x 8 bytes per image)
    return (size_t)n_images * align_up(sicn_codec_workspace_bytes(mode, n_symbols), 256) + align_up(16 * (size_t)n_images, 256);
}

extern "C" int sicn_codec_encode_batch_async_sl(const uint8_t *latents, uint32_t n_images, uint32_t lat_w, uint32_t lat_h,
                                                uint32_t lat_c, uint32_t img_w, uint32_t img_h, uint8_t *out, size_t slot_bytes,
                                                sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes,
                                                void *hip_stream, uint32_t wss)
{
    if (!out || !status_dev || !wstream_ok(wss)) return SICN_EINVAL;
    const unsigned long long n64 = (unsigned long long)lat_w * lat_h * lat_c;
    if (!wstream_fits(n64, wss) || (n64 && !latents) || n_images > 65535) return SICN_EINVAL;
    if (n_images == 0) return SICN_OK;
    const uint32_t n = (uint32_t)n64, ns = (n + wss - 1) / wss, wcap = wstream_cap(wss);
    if (slot_bytes < sicn_codec_max_bytes_sl(n, wss) || (slot_bytes & 1)) return SICN_ENOSPC;
    const size_t ws1 = align_up(sicn_codec_workspace_bytes_sl(n, wss), 256);
    if (!workspace || workspace_bytes < ws1 * n_images) return SICN_ENOSPC;
    hipStream_t stream = (hipStream_t)hip_stream;
    Workspace w;
    carve(w, workspace, ns, wcap);
    uint32_t *status = (uint32_t *)status_dev;
    uint8_t *table = out + SICN_CODEC_HEADER_BYTES + 256, *payload = table + 4 * (size_t)ns;
    const uint32_t fixed = (uint32_t)(SICN_CODEC_HEADER_BYTES + 256 + 4 * (size_t)ns);
    if (ns && ns <= SELF_SCAN_MAX) {
        // three launches (round 4; four in round 3, six in round 2): statistics in rows (no clear, no atomics) -> streams, every
        // wave making the frequency table from the rows itself and stream 0's wave writing header, table and status ->
        // compaction with its own scan
        const uint32_t n_rows = std::min(std::max(n / 16384u, 1u), STAT_ROWS);
        hipLaunchKernelGGL(k_stats, dim3(n_rows, n_images), dim3(256), 0, stream, latents, n, w.hist, w.sums, (size_t)n, ws1, w.rows);
        const EncSelfHeader sh{w.rows, n_rows, out, slot_bytes, status, lat_w, lat_h, lat_c, img_w, img_h};
        hipLaunchKernelGGL(k_ransw_encode, dim3(ns, n_images), dim3(64), 0, stream, latents, n, ns, w.freq, w.scratch, w.lens,
                           (size_t)n, ws1, wss, sh);
        hipLaunchKernelGGL(k_compact_self, dim3(ns, n_images), dim3(256), 0, stream, w.scratch, w.lens, out, wcap, ns, ws1, slot_bytes,
                           status, fixed);
        return hipGetLastError() == hipSuccess ? SICN_OK : SICN_ENODEV;
    }
    hipLaunchKernelGGL(k_clear_stats, dim3(1, n_images), dim3(64), 0, stream, w.hist, ws1);
    if (n)
        hipLaunchKernelGGL(k_stats, dim3(stats_blocks(n, n_images), n_images), dim3(256), 0, stream, latents, n, w.hist,
                           w.sums, (size_t)n, ws1);
    hipLaunchKernelGGL(k_enc_header, dim3(1, n_images), dim3(64), 0, stream, w.hist, w.sums, w.freq, out, status, n, ns, lat_w,
                       lat_h, lat_c, img_w, img_h, ws1, slot_bytes, wss);
    if (ns)
        hipLaunchKernelGGL(k_ransw_encode, dim3(ns, n_images), dim3(64), 0, stream, latents, n, ns, w.freq, w.scratch, w.lens,
                           (size_t)n, ws1, wss);
    hipLaunchKernelGGL(k_scan, dim3(1, n_images), dim3(1024), 0, stream, w.lens, (const uint8_t *)nullptr, ns, w.offsets, table,
                       out + 40, ws1, slot_bytes, 0xFFFFFFFFu, (uint32_t *)nullptr, status, fixed);
    if (ns)
        hipLaunchKernelGGL(k_compact, dim3(ns, n_images), dim3(256), 0, stream, w.scratch, w.lens, w.offsets, payload, wcap, ws1,
                           slot_bytes);
    return hipGetLastError() == hipSuccess ? SICN_OK : SICN_ENODEV;
}

extern "C" int sicn_codec_encode_batch_async(const uint8_t *latents, uint32_t n_images, uint32_t lat_w, uint32_t lat_h,
                                             uint32_t lat_c, uint32_t img_w, uint32_t img_h, uint8_t *out, size_t slot_bytes,
                                             sicn_codec_status *status_dev, void *workspace, size_t workspace_bytes,
                                             void *hip_stream)
{
    return sicn_codec_encode_batch_async_sl(latents, n_images, lat_w, lat_h, lat_c, img_w, img_h, out, slot_bytes, status_dev, workspace,
                                            workspace_bytes, hip_stream, WSS);
}

extern "C" int sicn_codec_decode_batch_async_sl(const uint8_t *containers, size_t slot_bytes, const sicn_codec_status *valid_dev_or_null,
                                                uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c, uint8_t *latents,
                                                size_t latent_stride, sicn_codec_status *status_dev, void *workspace,
                                                size_t workspace_bytes, void *hip_stream, uint32_t wss)
{
    if (!containers || !status_dev || (slot_bytes & 1) || !wstream_ok(wss)) return SICN_EINVAL;
    const unsigned long long n64 = (unsigned long long)lat_w * lat_h * lat_c;
    if (!wstream_fits(n64, wss) || n_images > 65535) return SICN_EINVAL;
    if (n_images == 0) return SICN_OK;
    const uint32_t n = (uint32_t)n64, ns = (n + wss - 1) / wss;
    if (n && (!latents || latent_stride < n)) return SICN_ENOSPC;
    const size_t ws1 = align_up(sicn_codec_workspace_bytes_sl(n, wss), 256);
    if (!workspace || workspace_bytes < ws1 * n_images) return SICN_ENOSPC;
    hipStream_t stream = (hipStream_t)hip_stream;
    Workspace w;
    carve(w, workspace, ns, 0);
    const uint8_t *freq_bytes = containers + SICN_CODEC_HEADER_BYTES, *table = freq_bytes + 256, *payload = table + 4 * (size_t)ns;
    if (ns <= SELF_SCAN_MAX) {
        // two launches instead of four: the streams bound and place themselves (self mode), the finish kernel validates header
        // and table and folds the per-stream verdicts / checksum sums (a small image's decode is mostly launch latency)
        uint32_t *serr = w.lens;                                           // [ns]
        unsigned long long *ssum = (unsigned long long *)w.scratch;        // [2 ns]; the scratch slots (>= 2304 B each) are idle in a decode
        const uint32_t *vb = valid_dev_or_null ? &valid_dev_or_null->bytes : (const uint32_t *)nullptr;
        if (ns) {
            // the latency form (one 16 KB table, one LDS round trip less per step) while all waves of the batch fit the chip at
            // five per CU; beyond that the two-table form, whose lookups the other eleven waves of a CU hide
            sicn::ChipGeom chip;
            const bool big = sicn::chip_geom(&chip) == SICN_OK && (unsigned long long)ns * n_images <= 5ull * (unsigned)chip.n_cu;
            if (big)
                hipLaunchKernelGGL(k_ransw_decode<true>, dim3(ns, n_images), dim3(64), 0, stream, payload, freq_bytes, (const uint32_t *)nullptr, n, ns,
                                   latents, serr, slot_bytes, ws1, latent_stride, containers + 40, (const uint32_t *)nullptr, ssum, wss, vb, 2u, 1);
            else
                hipLaunchKernelGGL(k_ransw_decode<false>, dim3(ns, n_images), dim3(64), 0, stream, payload, freq_bytes, (const uint32_t *)nullptr, n, ns,
                                   latents, serr, slot_bytes, ws1, latent_stride, containers + 40, (const uint32_t *)nullptr, ssum, wss, vb, 2u, 1);
        }
        hipLaunchKernelGGL(k_dec_finish_self, dim3(1, n_images), dim3(256), 0, stream, containers, vb, 2u, serr, ssum, (uint32_t *)status_dev,
                           n, ns, lat_w, lat_h, lat_c, slot_bytes, ws1, wss);
        return hipGetLastError() == hipSuccess ? SICN_OK : SICN_ENODEV;
    }
    hipLaunchKernelGGL(k_dec_parse, dim3(1, n_images), dim3(64), 0, stream, containers,
                       valid_dev_or_null ? &valid_dev_or_null->bytes : (const uint32_t *)nullptr, 2u, w.meta, n, ns, lat_w, lat_h,
                       lat_c, slot_bytes, ws1, wss);
    // a slot the parse stage rejected as too short is never read: its table counts as empty (ns_eff = 0 via meta[1] = 0 and
    // the per-stream bound off + len <= payload bytes, so every stream of it flags an error and returns)
    hipLaunchKernelGGL(k_scan, dim3(1, n_images), dim3(1024), 0, stream, (const uint32_t *)nullptr, table, ns, w.offsets,
                       (uint8_t *)nullptr, (uint8_t *)nullptr, ws1, slot_bytes, wstream_cap(wss), w.meta + 3, (uint32_t *)nullptr, 0u,
                       (const uint32_t *)w.meta);
    if (ns)
        hipLaunchKernelGGL(k_ransw_decode<false>, dim3(ns, n_images), dim3(64), 0, stream, payload, freq_bytes, w.offsets, n, ns, latents,
                           w.meta + 3, slot_bytes, ws1, latent_stride, containers + 40, w.meta, w.sums, wss);
    hipLaunchKernelGGL(k_dec_finish, dim3(1, n_images), dim3(64), 0, stream, w.meta, w.hist, w.sums, w.offsets, (uint32_t *)status_dev,
                       n, ns, ws1);
    return hipGetLastError() == hipSuccess ? SICN_OK : SICN_ENODEV;
}

extern "C" int sicn_codec_decode_batch_async(const uint8_t *containers, size_t slot_bytes, const sicn_codec_status *valid_dev_or_null,
                                             uint32_t n_images, uint32_t lat_w, uint32_t lat_h, uint32_t lat_c, uint8_t *latents,
                                             size_t latent_stride, sicn_codec_status *status_dev, void *workspace,
                                             size_t workspace_bytes, void *hip_stream)
{
    return sicn_codec_decode_batch_async_sl(containers, slot_bytes, valid_dev_or_null, n_images, lat_w, lat_h, lat_c, latents,
                                            latent_stride, status_dev, workspace, workspace_bytes, hip_stream, WSS);
}

static int status_to_rc(uint32_t err)
{
    if (!err) return SICN_OK;
    return (err & ~128u) ? SICN_EINVAL : SICN_EBADMSG;
}

extern "C" int sicn_codec_encode_batch(int mode, const uint8_t *latents, uint32_t n_images, uint32_t lat_w, uint32_t lat_h,
                                       uint32_t lat_c, uint32_t img_w, uint32_t img_h, uint8_t *out, size_t slot_bytes,
                                       size_t *out_bytes_host, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    if (mode != SICN_CODEC_RANSW || !out || !out_bytes_host) return SICN_EINVAL;
    if (n_images == 0) return SICN_OK;
    // the status array lives behind the per-image blocks of the caller's workspace (sicn_codec_batch_workspace_bytes)
    const unsigned long long n64 = (unsigned long long)lat_w * lat_h * lat_c;
    if (n64 > MAX_RANS_SYMBOLS) return SICN_EINVAL;
    const size_t blocks = (size_t)n_images * align_up(sicn_codec_workspace_bytes(mode, (uint32_t)n64), 256);
    if (!workspace || workspace_bytes < sicn_codec_batch_workspace_bytes(mode, (uint32_t)n64, n_images)) return SICN_ENOSPC;
    hipStream_t stream = (hipStream_t)hip_stream;
    sicn_codec_status *st_dev = (sicn_codec_status *)((uint8_t *)workspace + blocks);
    std::vector<sicn_codec_status> st;
    try { st.resize(n_images); } catch (const std::bad_alloc &) { return SICN_ENOMEM; }
    int rc = sicn_codec_encode_batch_async(latents, n_images, lat_w, lat_h, lat_c, img_w, img_h, out, slot_bytes, st_dev, workspace,
                                           workspace_bytes, hip_stream);
    if (rc == SICN_OK && hipMemcpyAsync(st.data(), st_dev, sizeof(sicn_codec_status) * n_images, hipMemcpyDeviceToHost, stream) != hipSuccess)
        rc = SICN_ENODEV;
    if (hipStreamSynchronize(stream) != hipSuccess) return SICN_ENODEV;
    if (rc) return rc;
    for (uint32_t i = 0; i < n_images; i++) {
        if (st[i].error) return SICN_EINVAL;
        out_bytes_host[i] = st[i].bytes;
    }
    return SICN_OK;
}

extern "C" int sicn_codec_decode_batch(const uint8_t *containers, size_t slot_bytes, const size_t *bytes_host, uint32_t n_images,
                                       uint8_t *latents, size_t latent_stride, sicn_codec_info *infos_or_null, void *workspace,
                                       size_t workspace_bytes, void *hip_stream)
{
    if (!containers || !bytes_host || (slot_bytes & 1)) return SICN_EINVAL;
    if (n_images == 0) return SICN_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    std::vector<uint8_t> head;
    std::vector<sicn_codec_info> info;
    std::vector<sicn_codec_status> st;
    try {
        head.resize((size_t)n_images * SICN_CODEC_HEADER_BYTES);
        info.resize(n_images);
        st.resize(2 * (size_t)n_images);
    } catch (const std::bad_alloc &) { return SICN_ENOMEM; }
    // first synchronisation: the shape comes out of the containers (the async entry point takes it from the caller)
    for (uint32_t i = 0; i < n_images; i++) {
        if (bytes_host[i] < SICN_CODEC_HEADER_BYTES + 256 || bytes_host[i] > slot_bytes) return SICN_EINVAL;
        HIP_TRY(hipMemcpyAsync(&head[(size_t)i * SICN_CODEC_HEADER_BYTES], containers + (size_t)i * slot_bytes, SICN_CODEC_HEADER_BYTES,
                               hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < n_images; i++) {
        int rc = sicn_codec_parse_header(&head[(size_t)i * SICN_CODEC_HEADER_BYTES], SICN_CODEC_HEADER_BYTES, &info[i]);
        if (rc) return rc;
        if (info[i].mode != SICN_CODEC_RANSW) return SICN_EINVAL;
        if (info[i].lat_w != info[0].lat_w || info[i].lat_h != info[0].lat_h || info[i].lat_c != info[0].lat_c) return SICN_EINVAL;
        if (info[i].stream_symbols != info[0].stream_symbols) return SICN_EINVAL;   // one batch = one stream length
        if (info[i].n_symbols > MAX_RANS_SYMBOLS) return SICN_EINVAL;
        if (infos_or_null) infos_or_null[i] = info[i];
        st[i].error = 0;
        st[i].bytes = (uint32_t)bytes_host[i];
    }
    const uint32_t wss = info[0].stream_symbols;
    const size_t blocks = (size_t)n_images * align_up(sicn_codec_workspace_bytes_sl(info[0].n_symbols, wss), 256);
    if (!workspace || workspace_bytes < sicn_codec_batch_workspace_bytes_sl(info[0].n_symbols, n_images, wss)) return SICN_ENOSPC;
    sicn_codec_status *st_dev = (sicn_codec_status *)((uint8_t *)workspace + blocks);   // [0, n): valid bytes in, [n, 2n): verdicts out
    int rc = hipMemcpyAsync(st_dev, st.data(), sizeof(sicn_codec_status) * n_images, hipMemcpyHostToDevice, stream) == hipSuccess
                 ? SICN_OK : SICN_ENODEV;
    if (rc == SICN_OK)
        rc = sicn_codec_decode_batch_async_sl(containers, slot_bytes, st_dev, n_images, info[0].lat_w, info[0].lat_h, info[0].lat_c,
                                              latents, latent_stride, st_dev + n_images, workspace, workspace_bytes, hip_stream, wss);
    if (rc == SICN_OK && hipMemcpyAsync(st.data() + n_images, st_dev + n_images, sizeof(sicn_codec_status) * n_images,
                                        hipMemcpyDeviceToHost, stream) != hipSuccess)
        rc = SICN_ENODEV;
    if (hipStreamSynchronize(stream) != hipSuccess) return SICN_ENODEV;
    if (rc) return rc;
    for (uint32_t i = 0; i < n_images; i++)
        if (int e = status_to_rc(st[n_images + i].error)) return e;
    return SICN_OK;
}

#include "sicn_codec_ctx.inc"   // container mode 4: class-conditional rANS-W with a checkerboard context model
