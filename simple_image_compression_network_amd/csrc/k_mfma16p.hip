// conv2d<> / deconv522<> with 128 / 192 channels — k_mfma16.hip's kernels with an explicit software pipeline.
// Same decomposition (conv: 4 parity planes with rolling refresh; deconv: 4 output phases over a resident patch), same
// 8 x TX-position workgroup tiles (wave w = rows 2w, 2w+1; TX = 32, or 16 for the 192-channel layers and small grids),
// two workgroups per CU = two waves per SIMD, same LDS images, same weight stream; read k_mfma16.hip first.
//
// What changes: in k_mfma16 a pass is  fragment reads -> MFMAs -> counted wait + barrier,  and the only thing that hides a
// wave's LDS round trip and barrier is its partner wave (measured MFMA pipe utilisation 0.61 on layer 6) — and on small
// grids, where a CU holds a single workgroup, nothing does.  The fragments cannot simply be double-buffered there:
// 128 accumulators + 2 x 48 fragment registers + addressing do not fit 256 VGPRs (and hipcc splits a 256-register budget
// 128 / 128 between VGPRs and AGPRs as soon as an AGPR is used).  Here
//   * the MFMAs are inline-asm statements with the accumulator tied ("+v"): volatile, so the order written is the order
//     issued, and hipcc's scheduler no longer needs room to move them;
//   * weight fragment j is RE-READ for the next pass into its own registers right after its MFMAs (it is dead by then): no
//     second set; only the pixel fragments (used by every weight tile of the pass) are double-buffered (+16 VGPRs);
//   * the barrier that ends pass k therefore publishes the weight tiles of pass k+2: the ring runs 4 passes ahead (PFP = 8);
//   * a pass's LDS-DMA requests are issued behind its first MFMA group: the MFMA pipe restarts right after the barrier.
// Also in this file: the output-channel split of the 8 x 16 kernels for grids smaller than the chip (NT16 < NTF, launch_p) and,
// under -DSICN_STAMP only, in-kernel s_memtime stamps for tools/*_stamps.py.  (Round 2's persistent conv kernel k_conv_pp, with
// its per-XCD ticket scheduler in device memory owned by the weights handle, is gone: full-size grids now run the wide persistent
// kernels of k_mfma16x.hip, which deal their tiles statically and keep no scheduler state at all.)
// Hazards hipcc cannot see inside asm are covered by hand: s_nop between a VALU write and an asm MFMA that reads it as C, an
// asm statement that keeps the C operand's registers live (and waits) behind the MFMAs that read them, s_nop before the
// epilogue reads the accumulators.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "k_common.hpp"

namespace sicn {

// PF = weight tiles (K steps) requested ahead of the MFMAs that use them = ring slots (the tile requested in pass k overwrites
// the slot whose fragments pass k - 1 finished reading).  The fetch for pass k+1 runs in pass k, so a tile must have landed at
// the barrier that ends pass k - 1: requested in pass k + 1 - PF/2, it has PF/2 - 2 passes of flight time, and that many passes'
// requests may still be in flight at a barrier.  PF = 8 (2 passes) where LDS is tight; 12 (4 passes) for the 8 x 16 tiles of
// the 128-channel shapes, whose passes are only 16 MFMAs long — shorter than an L2 round trip.
template <int PF>
struct Ring {
    static constexpr int SLOTS = PF, FLIGHT = PF / 2 - 2;
    static_assert(PF % 2 == 0 && FLIGHT >= 1, "");
};

// the deconv's tap loop: a loop over the taps (ROLLED, round 2's form: 20 KB of code) or unrolled completely (round 4: offsets,
// ring slots and parities become constants — in-kernel stamps on one 1080p image had shown 520 - 580 cycles per 16-MFMA pass
// against 360 for the conv kernel, whose 50 passes have always been unrolled).  Measured (us per layer, one image, rolled ->
// unrolled): 1080p layers 4 / 5 / 6 19.9 / 23.1 / 45.3 -> 17.6 / 20.6 / 42.2, 256 x 256 16.9 / 14.4 / 14.9 -> 14.6 / 11.9 / 12.2;
// on FULL grids (two workgroups per CU walking 33 - 47 KB of code each) the unrolled form is 3 % slower (layer 4 of 8 x 4K:
// 0.132 -> 0.136 ms), so the launcher keeps the loop there (profiles/r04_deconv_unroll.txt).
#define SICN_DECONV_TAP_LOOP _Pragma("unroll ROLLED ? 1 : 32")
#define SICN_MFMA_V(ACC, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define SICN_MFMA_V_C(ACC, A, B, C) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(ACC) : "v"(A), "v"(B), "v"(C))

constexpr int PAD_TILES_P = 24;   // zero tiles behind a weight stream (k_mfma16.hip: PAD16)
template <int NT16>
struct WTile {
    static constexpr int TB = NT16 * 16 * KSTEP;     // 4096 / 6144 bytes
    static constexpr int NPB = TB / 1024;            // 1-KiB pieces: 4 / 6
    static constexpr int WR = (NPB + 3) / 4;         // LDS-DMA instructions per wave and tile: 1 / 2
};

// NTF = weight tiles (of 16 channels) of the whole layer: the stride of the stream; NT16 <= NTF = the ones this workgroup
// computes (output-channel split, see launch_p); `wstream` already points at its first row
template <int NT16, int NTF, int PF>
__device__ __forceinline__ void load_wtile_p(uint8_t *ring, const int8_t *wstream, int tile, int lane, int w, int src_tile = -1)
{
    // `tile`: the running number of the K step in THIS workgroup's walk (ring slot = tile % PF); src_tile: where it sits in the
    // layer's stream when the two differ (K split of a deconv: a slice walks every (NQ / NQL)-th group pair), -1 = the same
    constexpr int TB = WTile<NT16>::TB, NPB = WTile<NT16>::NPB, WR = WTile<NT16>::WR;
    const int8_t *src = wstream + (size_t)(src_tile < 0 ? tile : src_tile) * WTile<NTF>::TB + lane * 16;
    uint8_t *dst = ring + (tile % PF) * TB;
#pragma unroll
    for (int r = 0; r < WR; r++) {
        int piece = r * 4 + w;
        if (piece >= NPB) piece -= 2;   // NPB == 6 / 2: waves 2, 3 re-load pieces 4, 5 / 0, 1 (same bytes): every wave issues WR requests
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + piece * 1024), LDS_PTR(dst + piece * 1024), 16, 0, 0);
    }
}

// One pass: NC * NT16 MFMAs (weight tile j outer, the wave's NC column tiles inner) on the CURRENT fragments (wc, pc); the
// fragments of the NEXT pass are fetched in between.
//   TX = 32 (32 MFMAs per pass, registers tight): weight fragment j is re-read INTO ITS OWN registers right after its MFMAs
//            (wn aliases wc); only the pixel fragments have a second buffer.  The last re-read is issued behind the last
//            MFMA group, so the barrier's lgkmcnt(0) exposes one LDS round trip per pass.
//   TX = 16 (16-24 MFMAs per pass, registers to spare): the weight fragments are double-buffered too and all reads are
//            issued two per MFMA group from the start of the pass — on small grids a CU holds one workgroup and nothing
//            else would hide that round trip.  (Measured worth only 2-5 % on 1080p: the small-grid passes are bound by
//            something else — the barrier among waves that have no partner workgroup; kept because it is free.)
//   FIRST: the accumulators start here, C operand = the bias (bias4[J] = this lane's 16 bias bytes of weight tiles 4J..4J+3)
//   dma(): this pass's LDS-DMA requests, issued behind the first MFMA group (their targets are free for the whole pass)
template <int TX, int NT16, int VMCNT, int EXTRA, bool FIRST, bool WDB, typename Dma>
__device__ __forceinline__ void pass_p(v4i (&acc)[Geo<TX>::NC][NT16], const v4i (&wc)[NT16], v4i (&wn)[NT16], const v4i (&pc)[Geo<TX>::NC],
                                       v4i (&pn)[Geo<TX>::NC], const uint8_t *pix_next, const uint8_t *wt_next, bool extra,
                                       const v4i (&bias4)[NT16 / 4], Dma dma, unsigned long long *st = nullptr)
{
    constexpr int PX = Geo<TX>::PX, NC = Geo<TX>::NC, XT = Geo<TX>::XT;
#if defined(SICN_STAMP) && SICN_STAMP > 1   // diagnostic build only (tools/pass_stamps.py): st = {last stamp, sum of issue phases, sum of wait + barrier phases}; level 2 stamps every pass (and slows the kernel 2.4 x: an s_memtime drains lgkmcnt)
    if (st) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        st[2] += now - st[0];
        st[0] = now;
    }
#endif
    // WDB: weights double-buffered (wn != wc)
    auto fetch = [&](auto r_tag) {           // read number r of the next pass: the NC pixel fragments first, then the weights
        constexpr int r = decltype(r_tag)::value;
        if constexpr (r < NC)
            pn[r] = *(const v4i *)(pix_next + ((r / XT) * PX + (r % XT) * 16) * 32);
        else if constexpr (r < NC + NT16)
            wn[r - NC] = *(const v4i *)(wt_next + (r - NC) * 16 * 32);
    };
#pragma unroll
    for (int j = 0; j < NT16; j++) {
        v4i cin;
        if constexpr (FIRST) {
#pragma unroll
            for (int r = 0; r < 4; r++) cin[r] = (int)(int8_t)((uint32_t)bias4[j >> 2][j & 3] >> (8 * r));
            // VALU write -> MFMA read of the same VGPR needs wait states hipcc would insert for a builtin MFMA but cannot for
            // an asm one (measured: without them the group's first MFMA saw a stale cin[3])
            asm volatile("s_nop 3" : "+v"(cin));
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if constexpr (FIRST)
                SICN_MFMA_V_C(acc[c][j], wc[j], pc[c], cin);
            else
                SICN_MFMA_V(acc[c][j], wc[j], pc[c]);
        }
        // An MFMA reads its C operand a few cycles AFTER it issues, and hipcc cannot see through the asm that `cin` is an
        // MFMA operand: any VALU write into those 4 registers right behind the group (the next group's unpacking, address
        // arithmetic of the next pass) would be a write-after-read hazard.  This statement keeps `cin` live — so its registers
        // cannot be given away earlier — and spends the wait states.
        if constexpr (FIRST) asm volatile("s_nop 7" ::"v"(cin));
        if constexpr (WDB) {
            // two reads per group: (2j, 2j+1) of [pixel 0..NC-1, weight 0..NT16-1]
            switch (j) {
#define SICN_F(J) case J: fetch(std::integral_constant<int, 2 * J>{}); fetch(std::integral_constant<int, 2 * J + 1>{}); break;
                SICN_F(0) SICN_F(1) SICN_F(2) SICN_F(3) SICN_F(4) SICN_F(5) SICN_F(6) SICN_F(7) SICN_F(8) SICN_F(9) SICN_F(10) SICN_F(11)
#undef SICN_F
            }
        } else {
            switch (j) {   // in place: weight j (dead now), plus pixel fragment j while there are any
#define SICN_F(J) case J: fetch(std::integral_constant<int, NC + J>{}); if (J < NC) fetch(std::integral_constant<int, (J < NC ? J : 0)>{}); break;
                SICN_F(0) SICN_F(1) SICN_F(2) SICN_F(3) SICN_F(4) SICN_F(5) SICN_F(6) SICN_F(7) SICN_F(8) SICN_F(9) SICN_F(10) SICN_F(11)
#undef SICN_F
            }
        }
#ifndef SICN_X_DMA_AT
#define SICN_X_DMA_AT 0   // the MFMA group behind which a pass issues its LDS-DMA requests (experiment knob: 2 / 4 / 7 measured within 1 % of 0)
#endif
        if (j == (SICN_X_DMA_AT < NT16 ? SICN_X_DMA_AT : NT16 - 1)) dma();
    }
#if defined(SICN_STAMP) && SICN_STAMP > 1
    if (st) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        st[1] += now - st[0];
        st[0] = now;
    }
#endif
    if (EXTRA > 0 && extra)
        wait_vmcnt<VMCNT + EXTRA>();
    else
        wait_vmcnt<VMCNT>();
#if defined(SICN_X_NOBAR)      // experiment (wrong results): what does the per-pass workgroup barrier cost?
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(SICN_X_NOWAIT)   // experiment (wrong results): neither the counted wait's effect nor the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    block_barrier();
#endif
}

#ifdef SICN_STAMP
__device__ unsigned long long *g_sicn_stamp = nullptr;   // [workgroup][wave][4]: total, prologue, issue, wait (cycles)
__device__ unsigned long long g_sicn_stagger = 0;
extern "C" int sicn_debug_stamp_buffer(void *p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sicn_stamp), &p, sizeof p);
}
extern "C" int sicn_debug_stagger(unsigned long long cycles)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sicn_stagger), &cycles, sizeof cycles);
}
#endif

// ReLU / pack / store of one accumulator set (as store_tiles16 of k_mfma16.hip); always NC * NT16 / 4 stores per wave
// THROUGH: the stores bypass the caches (sc0 sc1: written through to memory) — the K-split partial tensors, which a workgroup on
// ANOTHER XCD reads in the same launch (ksplit_finish_tile, with cache-bypassing loads): measured necessary — with ordinary
// stores and loads plus agent-scope release / acquire fences a finishing workgroup could still hit a line in its XCD's L2 that a
// neighbour's finisher had fetched before the other half of the line (the other channel slice's bytes) was written back
template <int TX, int NT16, bool THROUGH = false>
__device__ __forceinline__ void store_tiles_p(v4i (&acc)[Geo<TX>::NC][NT16], uint8_t *out_img, int out_img_bytes, const TensorMap &om,
                                              int MW, int MH, int Y0, int X0, int w, int pos, int g, bool deconv, int py, int px,
                                              uint32_t act_floor, uint32_t cg0)
{
    constexpr int NC = Geo<TX>::NC, XT = Geo<TX>::XT;   // cg0: first 32-channel group of this workgroup's output channels
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last MFMAs' results -> VALU
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int j = 0; j < NT16; j++) asm volatile("" : "+v"(acc[c][j]));   // keeps the packs below behind the s_nop
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, out_img_bytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int gy = Y0 + 2 * w + c / XT, gx = X0 + (c % XT) * 16 + pos;
        const bool ok = gy < MH && gx < MW;
        const int oy_ = deconv ? 2 * gy + py : gy, ox_ = deconv ? 2 * gx + px : gx;
        const uint32_t off0 = tensor_offset(om, oy_, ox_, (uint32_t)(g >> 1) + cg0) + 16u * (g & 1);
#pragma unroll
        for (int J = 0; J < NT16 / 4; J++) {
            v4i v;
#pragma unroll
            for (int d = 0; d < 4; d++)
                v[d] = (int)pack4_relu7(acc[c][4 * J + d][0], acc[c][4 * J + d][1], acc[c][4 * J + d][2], acc[c][4 * J + d][3],
                                        act_floor & ACT_FLOOR_MASK);
            if constexpr (THROUGH)
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 17);   // sc0 sc1
            else if (act_floor & ACT_NT_STORE)
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 2);
            else
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 0);
        }
    }
}

// ---- K split (round 4): the channel-group pairs of a layer over KS workgroups (blockIdx.z) -----------------------------------
// Small grids leave most CUs idle while every workgroup walks a serial chain of 50 - 75 passes.  The M tiles cannot multiply and the
// output channels are already split (launch_p), so K is: slice z computes the partial sums of channel groups [z NQ / KS, (z + 1) NQ / KS)
// — slice 0 starts from the bias, the others from 0 — and stores their LOW BYTES (Z -> Z/256 is a ring homomorphism: the truncation
// may come before the final sum, conv_nonsquare_top.cpp:272 / mvau.hpp:160-170 only define the result mod 256) as a partial
// tensor of the layer's output shape in the caller's workspace.  Who finishes a tile is decided by ONE 64-bit word per (tile,
// channel slice): upper 56 bits = the net's random tag, low bits = arrival mask, updated by compare-and-swap (a single location,
// so the updates are totally ordered).  Whoever completes the mask adds the KS partial tensors byte-wise mod 256, applies the
// activation, stores the tile and clears the word.  A word that does not carry the tag (uninitialised workspace) counts as
// empty: no initialisation pass, no counter that must start at zero.
struct KSplitArgs {
    uint8_t *partials;            // [KS][n_images * out_img_bytes]
    unsigned long long stride;    // bytes between two slices' tensors
    unsigned long long *flags;    // [tiles * n_images][gridDim.y]
    unsigned long long tag;       // random, low 8 bits zero, never 0
};

template <int KS>
__device__ __forceinline__ bool ksplit_arrive(unsigned long long *word, unsigned long long tag, int z, uint8_t *lds_word)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this thread's partial stores are visible device-wide ...
    __syncthreads();                                     // ... and so are everybody else's of this workgroup
    if (threadIdx.x == 0) {
        unsigned long long cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want;
        do {
            const unsigned long long mask = ((cur ^ tag) >> 8) == 0 ? (cur & 0xffull) : 0ull;   // foreign content = nobody has arrived
            want = tag | mask | (1ull << z);
        } while (!__hip_atomic_compare_exchange_strong(word, &cur, want, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT));
        *(volatile uint32_t *)lds_word = (want & 0xffull) == ((1ull << KS) - 1) ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *(volatile uint32_t *)lds_word != 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other slices' partial stores are visible to this workgroup
    return last;
}

__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t b)   // four independent byte sums mod 256
{
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ uint32_t relu7_bytes(uint32_t x)   // bytes >= 128 (negative as int8) -> 0
{
    const uint32_t m = (x >> 7) & 0x01010101u;
    return x & ~((m << 8) - m);
}

// the finishing pass of a tile: the same pieces (16 bytes per lane) a normal epilogue stores, read from the KS partial tensors
template <int TX, int NT16, int KS>
__device__ __forceinline__ void ksplit_finish_tile(const uint8_t *part_img, unsigned long long stride, uint8_t *out_img, int out_img_bytes,
                                                   const TensorMap &om, int MW, int MH, int Y0, int X0, int w, int pos, int g, bool deconv,
                                                   uint32_t act_floor, uint32_t cg0)
{
    constexpr int NC = Geo<TX>::NC, XT = Geo<TX>::XT;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, out_img_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rp[KS];
#pragma unroll
    for (int z = 0; z < KS; z++) rp[z] = __builtin_amdgcn_make_buffer_rsrc((void *)(part_img + (size_t)z * stride), 0, out_img_bytes, 0x00020000);
    const bool relu = (act_floor & ACT_FLOOR_MASK) == ACT_FLOOR_RELU;
    const int nph = deconv ? 4 : 1;
    for (int ph = 0; ph < nph; ph++) {
        const int py = ph >> 1, px = ph & 1;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int gy = Y0 + 2 * w + c / XT, gx = X0 + (c % XT) * 16 + pos;
            const bool ok = gy < MH && gx < MW;
            const int oy_ = deconv ? 2 * gy + py : gy, ox_ = deconv ? 2 * gx + px : gx;
            const uint32_t off0 = tensor_offset(om, oy_, ox_, (uint32_t)(g >> 1) + cg0) + 16u * (g & 1);
#pragma unroll
            for (int J = 0; J < NT16 / 4; J++) {
                const uint32_t off = ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB;
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                // sc0 sc1: past the caches, straight from memory (see store_tiles_p<.., THROUGH>)
                v4u sum = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rp[0], off, 0, 17));
#pragma unroll
                for (int z = 1; z < KS; z++) {
                    const v4u v = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rp[z], off, 0, 17));
#pragma unroll
                    for (int d = 0; d < 4; d++) sum[d] = add_bytes(sum[d], v[d]);
                }
                if (relu) {
#pragma unroll
                    for (int d = 0; d < 4; d++) sum[d] = relu7_bytes(sum[d]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(sum, ro, off, 0, 0);
            }
        }
    }
}

struct TileCoord {
    int img, Y0, X0;
    bool valid;
    int item;   // logical index in the (tile x, tile y, image) list
};
template <int TX>
__device__ __forceinline__ TileCoord tile_coord(int tiles_x, int n_tiles, int n_images, int n_xcd)
{
    const int item = xcd_logical_index(n_tiles * n_images, n_xcd);
    if (item < 0) return TileCoord{0, 0, 0, false, 0};
    const int img = item / n_tiles, tile = item - img * n_tiles;
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    return TileCoord{img, tile_y * TILE_Y, tile_x * TX, true, item};
}

// =====================================================================================================================
// deconv522<>: NQ / 2 passes per tap (channel pairs q, q+1 of one tap), the tap loop stays a loop
// =====================================================================================================================
template <int NQ, int NT16, int NTF, int TX, int PF, int KS, bool ROLLED>
__global__ __launch_bounds__(256, 2) void k_deconv_p(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                     const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias, int IW, int IH,
                                                     int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout,
                                                     int out_layout, uint32_t act_floor, int n_xcd, KSplitArgs ks)
{
    // KS > 1: K split — this workgroup (blockIdx.z) walks NQL = NQ / KS = 2 of the layer's NQ channel groups (one pass per tap)
    constexpr int NQL = NQ / KS;
    static_assert(NQ % KS == 0 && (KS == 1 || NQL == 2), "a K slice is one channel-group pair");
    constexpr int CIN = NQ * 32, COUT = NTF * 16, COUTW = NT16 * 16, NC = Geo<TX>::NC, PPT = NQL / 2;   // passes per tap
    const int zs = KS > 1 ? (int)blockIdx.z : 0;
    // position of the workgroup's K step `L` (tap-major, its own groups) in the layer's stream (tap-major, all NQ groups)
    auto src_tile = [&](int L) { return KS == 1 ? -1 : (L / NQL) * NQ + zs * NQL + L % NQL; };
    constexpr int PX = Geo<TX>::PX, ALLOC = Geo<TX>::ALLOC, SLOTS = Geo<TX>::SLOTS;
    constexpr int TB = WTile<NT16>::TB, WR = WTile<NT16>::WR;
    const int split = blockIdx.y;                    // this workgroup's COUTW output channels start at split * COUTW
    wstream += split * TB;
    bias += split * COUTW;
    constexpr int NSTORE = NC * NT16 / 4;            // output stores per wave and phase
    constexpr int FL = Ring<PF>::FLIGHT;
    constexpr int VM = FL * 2 * WR;                  // requests of the last FL passes
    constexpr int BIAS_LDS = (Geo<TX>::PIX * KSTEP + 1023) / 1024 * 1024;   // the padding pieces behind sub-patch 0's positions
    static_assert(BIAS_LDS + COUTW <= ALLOC, "no room for the bias behind the patch");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem, *ring = smem + NQL * ALLOC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, hi = g >> 1, half = g & 1;
    const TileCoord tc = tile_coord<TX>(tiles_x, n_tiles, n_images, n_xcd);
    if (!tc.valid) return;   // before any LDS-DMA is issued
#ifdef SICN_STAMP
    unsigned long long dst[10];   // start, loop start, then (passes done, epilogue done) per phase
    dst[0] = __builtin_amdgcn_s_memtime();
#endif
    const int Y0 = tc.Y0, X0 = tc.X0;
    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const uint8_t *in_img = in + (size_t)tc.img * in_img_bytes;
    uint8_t *out_img = out + (size_t)tc.img * out_img_bytes;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);

    // this lane's fragment 0: pixel (sub-patch of its K half, rows 2w..), weight (ring slot 0)
    const uint8_t *lane_pix = patch + (uint32_t)(((2 * w) * PX + pos) * 32 + half * 16 + hi * ALLOC);
    const uint8_t *lane_wt = ring + (uint32_t)(pos * 32 + half * 16);

    // ---- prologue: the whole patch (NQ channel groups) + PFP weight tiles + the bias ---------------------------------------
#pragma unroll
    for (int slot = 0; slot < SLOTS; slot++) {
        const PieceSrc ps = piece_src<TX>(im, slot * 4 + w, lane, Y0 - 1, X0 - 1, 1, 0, 0, IW, IH);
#pragma unroll
        for (int sub = 0; sub < NQL; sub++)
            load_piece<ALLOC>(patch, in_img, in_img_bytes, sub, slot * 4 + w, ps.ok ? ps.off + (uint32_t)(zs * NQL + sub) * im.grp : OOB);
    }
#pragma unroll
    for (int s = 0; s < PF; s++) load_wtile_p<NT16, NTF, PF>(ring, wstream, s, lane, w, src_tile(s));
    // the bias goes to LDS (padding behind sub-patch 0): reading it from global memory at the start of every phase would make
    // hipcc drain vmcnt there, i.e. wait for the previous phase's stores.  (K split: only slice 0 starts from the bias.)
    uint32_t bias_dw = 0;
    if (tid < COUTW / 4 && zs == 0) bias_dw = ((const uint32_t *)bias)[tid];
    wait_vmcnt<0>();
    block_barrier();   // every wave's DMA has landed — including the padding piece (zeros) the bias is about to replace
    if (tid < COUTW / 4) ((uint32_t *)(patch + BIAS_LDS))[tid] = bias_dw;
    block_barrier();

    // a pass covers steps (s0, s0 + 1) = channel groups (q, q + 1) of one tap: fragments at
    //   pixel : lane_pix + tap offset + q * ALLOC            (the lane's K half already selects sub-patch q + hi)
    //   weight: lane_wt + ((s0 + hi) % RINGP) * TB
    // weight fragment buffers (pass_p); with an odd number of passes per tap the buffer parity is a run-time value and the
    // second weight set costs more registers than there are (256 + scratch, layer 4 +25 %): in place there
    constexpr int NWB = (TX == 16 && PPT % 2 == 0) ? 2 : 1;
    v4i wbuf[NWB][NT16], pbuf[2][NC];
    v4i acc[NC][NT16];
    int step = 0;
    {   // fragments of the first pass: phase 0, tap 0 (offset 0), q = 0, steps 0 / 1
#pragma unroll
        for (int r = 0; r < NT16; r++) wbuf[0][r] = *(const v4i *)(lane_wt + hi * TB + r * 16 * 32);
#pragma unroll
        for (int r = 0; r < NC; r++) pbuf[0][r] = *(const v4i *)(lane_pix + ((r / Geo<TX>::XT) * PX + (r % Geo<TX>::XT) * 16) * 32);
    }
    // Pass k's weight requests overwrite the ring slots of pass k's OWN tiles (prefetch distance = ring size): safe because every
    // wave read those fragments in pass k - 1, in front of the barrier that ended it — except for pass 0, whose fragments are the
    // reads just above.  Without this barrier a wave that runs ahead requests tiles 8 / 9 into slots 0 / 1 while a slower wave of
    // the workgroup has not read tiles 0 / 1 yet.  Two workgroups per CU never showed it (a request takes longer to land than
    // waves drift apart); the K-split kernels, whose smaller LDS footprint puts four to five workgroups on a CU, did: wrong bytes
    // in phase 0, weight piece 1 (the piece the waves that run ahead write), timing dependent (round 4).
    block_barrier();
    // one tap = PPT passes, the first one on pixel buffer PAR
    auto tap = [&](auto par_tag, int ph, int t, uint32_t toff, uint32_t toff_next) {
        constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
        for (int qi = 0; qi < PPT; qi++) {
            const int q = 2 * qi, s0 = step + q;
            const bool last = qi + 1 == PPT;
            const uint8_t *pixn = lane_pix + (last ? toff_next : toff + (uint32_t)((q + 2) * ALLOC));
            const uint8_t *wtn = lane_wt + (uint32_t)(((s0 + 2 + hi) % PF) * TB);
            auto dma = [&]() {
                load_wtile_p<NT16, NTF, PF>(ring, wstream, s0 + PF, lane, w, src_tile(s0 + PF));
                load_wtile_p<NT16, NTF, PF>(ring, wstream, s0 + 1 + PF, lane, w, src_tile(s0 + 1 + PF));
            };
            // the previous phase's NSTORE stores are younger than the tiles awaited in the first FL passes of a phase: they
            // are counted, not waited for
            const bool extra = ph > 0 && (t * PPT + qi) < FL;
            v4i(&pc)[NC] = pbuf[(PAR + qi) & 1];
            v4i(&pn)[NC] = pbuf[(PAR + qi + 1) & 1];
            v4i(&wc)[NT16] = wbuf[(PAR + qi) & (NWB - 1)];
            v4i(&wn)[NT16] = wbuf[(PAR + qi + 1) & (NWB - 1)];
            if (qi == 0 && t == 0) {
                v4i bias4[NT16 / 4];
#pragma unroll
                for (int J = 0; J < NT16 / 4; J++) bias4[J] = *(const v4i *)(patch + BIAS_LDS + 64 * J + 16 * g);
                pass_p<TX, NT16, VM, NSTORE, true, NWB == 2>(acc, wc, wn, pc, pn, pixn, wtn, extra, bias4, dma);
            } else {
                const v4i none[NT16 / 4] = {};
                pass_p<TX, NT16, VM, NSTORE, false, NWB == 2>(acc, wc, wn, pc, pn, pixn, wtn, extra, none, dma);
            }
        }
        step += NQL;
    };
#ifdef SICN_STAMP
    dst[1] = __builtin_amdgcn_s_memtime();
#endif
    // With an odd number of passes per tap (192 input channels) the two pixel buffers swap roles from tap to tap.  The parity
    // is kept STATIC — taps are walked in pairs, and a phase's first parity is a constant of the phase (9 / 6 / 6 / 4 taps) —
    // because a run-time parity made hipcc reconcile the two register assignments with v_mov copies right in front of and
    // right behind the asm MFMAs, inside their hazard windows (tools/isa_hazards.py, rules A and D; round 3).
#pragma unroll
    for (int ph = 0; ph < 4; ph++) {
        const int py = ph >> 1, px = ph & 1;
        const int nkx = 3 - px, ntap = (3 - py) * nkx;
        const int par0 = (PPT % 2) ? (ph == 0 ? 0 : ph == 1 ? (9 & 1) : ph == 2 ? ((9 + 6) & 1) : ((9 + 6 + 6) & 1)) : 0;
        auto offsets = [&](int t, uint32_t &toff, uint32_t &toff_next) {
            const int iy = t / nkx, ix = t - iy * nkx;
            toff = (uint32_t)(((iy + py) * PX + ix + px) * 32);
            if (t + 1 < ntap) {   // the tap after this one (the next phase's first tap at the end of a phase)
                const int t1 = t + 1, iy1 = t1 / nkx, ix1 = t1 - iy1 * nkx;
                toff_next = (uint32_t)(((iy1 + py) * PX + ix1 + px) * 32);
            } else {
                const int ph1 = (ph + 1) & 3;
                toff_next = (uint32_t)(((ph1 >> 1) * PX + (ph1 & 1)) * 32);
            }
        };
        auto one = [&](auto par_tag, int t) {
            uint32_t toff, toff_next;
            offsets(t, toff, toff_next);
            tap(par_tag, ph, t, toff, toff_next);
        };
        if constexpr (PPT % 2 == 0) {
SICN_DECONV_TAP_LOOP
            for (int t = 0; t < ntap; t++) one(std::integral_constant<int, 0>{}, t);   // an even number of passes: the parity never changes
        } else {
            int t = 0;
            if (par0 == 0) {
SICN_DECONV_TAP_LOOP
                for (; t + 1 < ntap; t += 2) {
                    one(std::integral_constant<int, 0>{}, t);
                    one(std::integral_constant<int, 1>{}, t + 1);
                }
                if (t < ntap) one(std::integral_constant<int, 0>{}, t);
            } else {
SICN_DECONV_TAP_LOOP
                for (; t + 1 < ntap; t += 2) {
                    one(std::integral_constant<int, 1>{}, t);
                    one(std::integral_constant<int, 0>{}, t + 1);
                }
                if (t < ntap) one(std::integral_constant<int, 1>{}, t);
            }
        }
#ifdef SICN_STAMP
        dst[2 + 2 * ph] = __builtin_amdgcn_s_memtime();
#endif
        if (ph == 3) wait_vmcnt<0>();  // the padded tail of the weight prefetch must land before exit
        if constexpr (KS == 1)
            store_tiles_p<TX, NT16>(acc, out_img, out_img_bytes, om, IW, IH, Y0, X0, w, pos, g, true, py, px, act_floor,
                                    (uint32_t)(split * (COUTW / 32)));
        else   // this slice's partial sums, low bytes, no activation, into its own tensor of the output's shape
            store_tiles_p<TX, NT16, true>(acc, ks.partials + (size_t)zs * ks.stride + (size_t)tc.img * out_img_bytes, out_img_bytes, om, IW, IH,
                                          Y0, X0, w, pos, g, true, py, px, ACT_FLOOR_RAW, (uint32_t)(split * (COUTW / 32)));
#ifdef SICN_STAMP
        dst[3 + 2 * ph] = __builtin_amdgcn_s_memtime();
#endif
    }
    if constexpr (KS > 1) {
        unsigned long long *word = ks.flags + (size_t)tc.item * gridDim.y + blockIdx.y;
        if (ksplit_arrive<KS>(word, ks.tag, zs, smem)) {
            ksplit_finish_tile<TX, NT16, KS>(ks.partials + (size_t)tc.img * out_img_bytes, ks.stride, out_img, out_img_bytes, om, IW, IH, Y0, X0,
                                             w, pos, g, true, act_floor, (uint32_t)(split * (COUTW / 32)));
            if (tid == 0) __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // empty again for the next launch
        }
    }
#ifdef SICN_STAMP
    if (g_sicn_stamp && lane == 0) {
        unsigned long long *o = g_sicn_stamp + ((size_t)blockIdx.x * 4 + w) * 12;
#pragma unroll
        for (int i = 0; i < 10; i++) o[i] = dst[i];
    }
#endif
}

// =====================================================================================================================
// conv2d<>: K walk, parity planes and their rolling refresh as in k_mfma16.hip; all 25 NQ / 2 passes of a tile unrolled (tap
// offsets and refresh slots are compile-time, and the two pixel buffers alternate while a 25-pass window is odd)
// =====================================================================================================================
__host__ __device__ constexpr int refresh_start_p(int plane) { return plane == 3 ? 1 : plane == 0 ? 10 : plane == 1 ? 16 : 22; }
__host__ __device__ constexpr int refresh_plane_p(int t, int S)
{
    for (int pl = 0; pl < 4; pl++)
        if (t >= refresh_start_p(pl) && t < refresh_start_p(pl) + S) return pl;
    return -1;
}
__host__ __device__ constexpr int refresh_slot_p(int t, int S) { return refresh_plane_p(t, S) < 0 ? 0 : t - refresh_start_p(refresh_plane_p(t, S)); }
template <int TX>
__host__ __device__ constexpr uint32_t tap_off_p(int t)   // byte offset of tap t (plane-ordered walk) inside the patch
{
    const Tap a = conv_tap(t);
    return (uint32_t)(((a.ky & 1) * 2 + (a.kx & 1)) * Geo<TX>::ALLOC + ((a.ky >> 1) * Geo<TX>::PX + (a.kx >> 1)) * 32);
}
template <int TX, int NT16>
__host__ __device__ constexpr int conv_requests_p(int P)   // LDS-DMA instructions a wave issues in pass P (steps 2P, 2P+1)
{
    if (P < 0) return 0;
    return 2 * WTile<NT16>::WR + (refresh_plane_p((2 * P) % 25, Geo<TX>::SLOTS) >= 0) + (refresh_plane_p((2 * P + 1) % 25, Geo<TX>::SLOTS) >= 0);
}
template <int TX, int NT16>
__host__ __device__ constexpr int conv_in_flight_p(int P, int n)   // requests of passes P, P-1, .. (n of them)
{
    int s = 0;
    for (int i = 0; i < n; i++) s += conv_requests_p<TX, NT16>(P - i);
    return s;
}

struct ConvPCtx {
    uint8_t *patch, *ring;
    const int8_t *wstream;
    const uint8_t *in_img;
    int in_img_bytes;
    const uint8_t *lane_pix, *lane_wt;
    uint32_t qstride;
    int lane, w, hi;
    unsigned long long *st;   // SICN_STAMP builds: this wave's stamp sums (nullptr otherwise)
};

template <int TX, int NT16, int NTF, int PF, int P, int NPASS>
__device__ __forceinline__ void conv_pass_p(v4i (&acc)[Geo<TX>::NC][NT16], const v4i (&wc)[NT16], v4i (&wn)[NT16],
                                            const v4i (&pc)[Geo<TX>::NC], v4i (&pn)[Geo<TX>::NC], const ConvPCtx &c,
                                            const uint32_t (&poff)[4][Geo<TX>::SLOTS])
{
    constexpr int ALLOC = Geo<TX>::ALLOC, S = Geo<TX>::SLOTS, TB = WTile<NT16>::TB;
    constexpr int tA = (2 * P) % 25, tB = (2 * P + 1) % 25, qA = (2 * P) / 25, qB = (2 * P + 1) / 25;
    constexpr int rpA = refresh_plane_p(tA, S), rpB = refresh_plane_p(tB, S);
    auto dma = [&]() {   // plane refresh pieces scheduled for these two steps, weight tiles PFP steps ahead
        if constexpr (rpA >= 0)
            load_piece<ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpA, refresh_slot_p(tA, S) * 4 + c.w,
                              poff[rpA][refresh_slot_p(tA, S)] + (uint32_t)((rpA == 3) ? qA : qA + 1) * c.qstride);
        if constexpr (rpB >= 0)
            load_piece<ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpB, refresh_slot_p(tB, S) * 4 + c.w,
                              poff[rpB][refresh_slot_p(tB, S)] + (uint32_t)((rpB == 3) ? qB : qB + 1) * c.qstride);
        load_wtile_p<NT16, NTF, PF>(c.ring, c.wstream, 2 * P + PF, c.lane, c.w);
        load_wtile_p<NT16, NTF, PF>(c.ring, c.wstream, 2 * P + 1 + PF, c.lane, c.w);
    };
    constexpr int PN = (P + 1) % NPASS;   // the pass whose fragments are fetched now
    constexpr uint32_t offNA = tap_off_p<TX>((2 * PN) % 25), offNB = tap_off_p<TX>((2 * PN + 1) % 25);
    const uint8_t *pixn = c.lane_pix + (c.hi ? offNB : offNA);
    const uint8_t *wtn = c.lane_wt + (uint32_t)((c.hi ? ((2 * PN + 1) % PF) : ((2 * PN) % PF)) * TB);
    const v4i none[NT16 / 4] = {};
    // everything but the requests of the last FLIGHT passes has landed at the barrier (= what pass P+2's fetch needs; a plane
    // refresh piece is requested at least 8 passes before its first read)
    pass_p<TX, NT16, conv_in_flight_p<TX, NT16>(P, Ring<PF>::FLIGHT), 0, false, TX == 16>(acc, wc, wn, pc, pn, pixn, wtn, false, none, dma, c.st);
}

template <int TX, int NT16, int NTF, int PF, int P, int NPASS>
__device__ __forceinline__ void conv_passes_p(v4i (&acc)[Geo<TX>::NC][NT16], v4i (&wbuf)[TX == 16 ? 2 : 1][NT16], v4i (&pa)[Geo<TX>::NC],
                                              v4i (&pb)[Geo<TX>::NC], const ConvPCtx &c, const uint32_t (&poff)[4][Geo<TX>::SLOTS])
{
    constexpr int NWB = TX == 16 ? 2 : 1;
    if constexpr ((P & 1) == 0)
        conv_pass_p<TX, NT16, NTF, PF, P, NPASS>(acc, wbuf[0], wbuf[NWB - 1], pa, pb, c, poff);
    else
        conv_pass_p<TX, NT16, NTF, PF, P, NPASS>(acc, wbuf[NWB - 1], wbuf[0], pb, pa, c, poff);
    if constexpr (P + 1 < NPASS) conv_passes_p<TX, NT16, NTF, PF, P + 1, NPASS>(acc, wbuf, pa, pb, c, poff);
}

template <int NQ, int NT16, int NTF, int TX, int PF, int KS>
__global__ __launch_bounds__(256, 2) void k_conv_p(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                   const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias, int IW, int IH,
                                                   int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout, int out_layout,
                                                   uint32_t act_floor, int n_xcd, KSplitArgs ks)
{
    // KS > 1: K split — this workgroup (blockIdx.z) walks channel groups [zs NQL, (zs + 1) NQL) only: one 25-pass window
    constexpr int NQL = NQ / KS;
    static_assert(NQ % KS == 0 && (KS == 1 || NQL == 2), "a K slice is one channel-group pair");
    constexpr int CIN = NQ * 32, COUT = NTF * 16, COUTW = NT16 * 16, NC = Geo<TX>::NC, NPASS = 25 * NQL / 2;
    static_assert(NQL % 2 == 0, "channel groups are consumed in pairs");
    constexpr int PX = Geo<TX>::PX, ALLOC = Geo<TX>::ALLOC, SLOTS = Geo<TX>::SLOTS, TB = WTile<NT16>::TB;
    const int split = blockIdx.y;                    // this workgroup's COUTW output channels start at split * COUTW
    const int zs = KS > 1 ? (int)blockIdx.z : 0;
    wstream += split * TB + (size_t)(zs * NQL * 25) * WTile<NTF>::TB;   // the stream is group-major: a slice's tiles are contiguous
    bias += split * COUTW;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem, *ring = smem + 4 * ALLOC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, hi = g >> 1, half = g & 1;
    const TileCoord tc = tile_coord<TX>(tiles_x, n_tiles, n_images, n_xcd);
    if (!tc.valid) return;
    const int Y0 = tc.Y0, X0 = tc.X0;
    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const uint8_t *in_img = in + (size_t)tc.img * in_img_bytes;
    uint8_t *out_img = out + (size_t)tc.img * out_img_bytes;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);

    uint32_t poff[4][SLOTS];   // channel group zs * NQL (the slice's first); out-of-image pieces stay beyond any image after the adds
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
#pragma unroll
        for (int slot = 0; slot < SLOTS; slot++) {
            const PieceSrc ps = piece_src<TX>(im, slot * 4 + w, lane, Y0 - 1, X0 - 1, 2, pl >> 1, pl & 1, IW, IH);
            poff[pl][slot] = ps.ok ? ps.off + (uint32_t)(zs * NQL) * im.grp : OOB;
        }
#ifdef SICN_STAMP
    unsigned long long st[3] = {0, 0, 0};
    if (g_sicn_stagger && blockIdx.x >= 256 && blockIdx.x < 512) {   // experiment: put a CU's two workgroups out of phase
        const unsigned long long t_wait = __builtin_amdgcn_s_memtime() + g_sicn_stagger;
        while (__builtin_amdgcn_s_memtime() < t_wait) __builtin_amdgcn_s_sleep(32);
    }
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    unsigned long long *stp = st;
#else
    unsigned long long *stp = nullptr;
#endif
    const ConvPCtx ctx{patch, ring, wstream, in_img, in_img_bytes, patch + (uint32_t)(((2 * w) * PX + pos) * 32 + half * 16),
                       ring + (uint32_t)(pos * 32 + half * 16), im.grp, lane, w, hi, stp};
    // ---- prologue: planes 0..2 of group 0 (plane 3 arrives in steps 1..) + PFP weight tiles ------------------------------
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int slot = 0; slot < SLOTS; slot++) load_piece<ALLOC>(patch, in_img, in_img_bytes, pl, slot * 4 + w, poff[pl][slot]);
#pragma unroll
    for (int s = 0; s < PF; s++) load_wtile_p<NT16, NTF, PF>(ring, wstream, s, lane, w);
    // accumulators start at the bias: register r of tile (c, j) is channel 64 (j>>2) + 16 g + 4 (j&3) + r
    v4i acc[NC][NT16];
#pragma unroll
    for (int J = 0; J < NT16 / 4; J++) {
        v4i b4 = *(const v4i *)(bias + 64 * J + 16 * g);
        if (KS > 1 && zs != 0) b4 = v4i{0, 0, 0, 0};   // K split: only slice 0 starts from the bias
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            v4i v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = (int)(int8_t)((uint32_t)b4[jj] >> (8 * r));
#pragma unroll
            for (int c = 0; c < NC; c++) acc[c][4 * J + jj] = v;
        }
    }
    wait_vmcnt<0>();
    block_barrier();
    v4i wbuf[TX == 16 ? 2 : 1][NT16], pa[NC], pb[NC];
    {   // fragments of pass 0: taps 0 / 1 of group 0, ring slots 0 / 1
        const uint8_t *p0 = ctx.lane_pix + (hi ? tap_off_p<TX>(1) : tap_off_p<TX>(0));
        const uint8_t *w0 = ctx.lane_wt + hi * TB;
#pragma unroll
        for (int r = 0; r < NT16; r++) wbuf[0][r] = *(const v4i *)(w0 + r * 16 * 32);
#pragma unroll
        for (int r = 0; r < NC; r++) pa[r] = *(const v4i *)(p0 + ((r / Geo<TX>::XT) * PX + (r % Geo<TX>::XT) * 16) * 32);
    }
    block_barrier();   // pass 0's requests go into the slots just read: every wave must have read them first (see k_deconv_p)
    // VALU-initialised accumulators -> asm MFMA reading them as C: the wait states hipcc cannot know about
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int j = 0; j < NT16; j++) asm volatile("" : "+v"(acc[c][j]));
    asm volatile("s_nop 3" ::: "memory");
#ifdef SICN_STAMP
    st[0] = __builtin_amdgcn_s_memtime();
    const unsigned long long t_loop = st[0], r_loop = __builtin_amdgcn_s_memrealtime();
#endif
    conv_passes_p<TX, NT16, NTF, PF, 0, NPASS>(acc, wbuf, pa, pb, ctx, poff);
#ifdef SICN_STAMP
    const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime(), r_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
    wait_vmcnt<0>();
    if constexpr (KS == 1) {
        store_tiles_p<TX, NT16>(acc, out_img, out_img_bytes, om, OW, OH, Y0, X0, w, pos, g, false, 0, 0, act_floor,
                                (uint32_t)(split * (COUTW / 32)));
    } else {
        // this slice's partial sums, low bytes, no activation, into its own tensor of the output's shape; the last slice to arrive
        // adds the KS tensors mod 256 and finishes the tile
        store_tiles_p<TX, NT16, true>(acc, ks.partials + (size_t)zs * ks.stride + (size_t)tc.img * out_img_bytes, out_img_bytes, om, OW, OH, Y0,
                                      X0, w, pos, g, false, 0, 0, ACT_FLOOR_RAW, (uint32_t)(split * (COUTW / 32)));
        unsigned long long *word = ks.flags + (size_t)tc.item * gridDim.y + blockIdx.y;
        if (ksplit_arrive<KS>(word, ks.tag, zs, smem)) {
            ksplit_finish_tile<TX, NT16, KS>(ks.partials + (size_t)tc.img * out_img_bytes, ks.stride, out_img, out_img_bytes, om, OW, OH, Y0, X0,
                                             w, pos, g, false, act_floor, (uint32_t)(split * (COUTW / 32)));
            if (tid == 0) __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // empty again for the next launch
        }
    }
#ifdef SICN_STAMP
    if (g_sicn_stamp && lane == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = g_sicn_stamp + ((size_t)blockIdx.x * 4 + w) * 8;
        o[6] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ..
        o[7] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
        o[0] = t_end - t_start;      // whole workgroup life of this wave
        o[1] = t_loop - t_start;     // prologue (patch + weight ring requests, their latency, first fragments)
#if SICN_STAMP > 1
        o[2] = st[1];                // passes: first MFMA .. last fragment read issued
        o[3] = st[2] + (t_end - st[0]);   // passes: vmcnt wait + barrier; + the epilogue (ReLU, pack, stores issued)
        o[4] = t_end - st[0];        // the epilogue alone
#else
        o[2] = t_loop_end - t_loop;  // all passes (level 1: two stamps around the loop, no perturbation inside)
        o[3] = r_loop_end - r_loop;  // the loop again, in 100 MHz ticks (-> the clock the chip held)
        o[4] = t_end - t_loop_end;   // epilogue: last barrier .. stores issued
#endif
        o[5] = t_start;
    }
#endif
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
// NT16 < NTF: output-channel split — blockIdx.y = which NT16 * 16 channels a workgroup computes.  Used on grids smaller
// than the chip (fewer than two workgroups per CU): the tile count cannot grow, the channel dimension can; each workgroup
// then runs the same number of passes with NT16 / NTF of the MFMAs, and the CUs that had no tile get one.
template <int NQ, int NT16, int NTF, bool DECONV, int TX, int KS = 1>
static hipError_t launch_p(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images, hipStream_t stream,
                           int in_layout, int out_layout, bool relu, const ChipGeom &chip, const KSplitScratch *ks = nullptr)
{
    constexpr int NSUB = DECONV ? NQ / KS : 4;
    // PF = 12 (4 passes of flight time) fits the 8 x 16 tiles of the 128-channel shapes but measured no gain on small grids
    // (1080p: layer 6 51 vs 48 us; round 4, the channel-split kernels at 256^2 ... 1080p with 12 / 16 / 24 slots: no gain either):
    // the weight stream's latency is not what bounds a pass there
    constexpr int PF = 8;
    static_assert(PF <= PAD_TILES_P, "prefetch would run off the weight stream");
    // K split of a deconv: a slice's prefetch runs up to tap 25 + PF / 2 of the tap-major stream
    static_assert(KS == 1 || !DECONV || ((49 + PF) / 2 + 1) * NQ - 1 < 25 * NQ + PAD_TILES_P, "K-split prefetch would run off the weight stream");
    const int MW = DECONV ? g.IW : g.OW, MH = DECONV ? g.IH : g.OH;
    const int tiles_x = (MW + TX - 1) / TX, tiles_y = (MH + TILE_Y - 1) / TILE_Y;
    constexpr size_t lds_need = (size_t)NSUB * Geo<TX>::ALLOC + (size_t)PF * WTile<NT16>::TB;
    static_assert(lds_need <= 80 * 1024, "two workgroups per CU");
#ifdef SICN_STAMP   // diagnostic build: SICN_X_LDS_PAD bytes of unused LDS force one workgroup per CU (a wave's solo pass rate)
    static const size_t lds_pad = getenv("SICN_X_LDS_PAD") ? (size_t)atoi(getenv("SICN_X_LDS_PAD")) : 0;
    const size_t lds = lds_need + lds_pad;
#else
    constexpr size_t lds = lds_need;
#endif
    const uint32_t flags = (relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW) |
                           (nt_store_wanted((size_t)g.OH * g.OW * g.COUT * n_images) ? ACT_NT_STORE : 0u);
    const dim3 grid(xcd_grid_size((long)tiles_x * tiles_y * n_images, chip.n_xcd), NTF / NT16, KS);
    KSplitArgs ka{nullptr, 0, nullptr, 0};
    if constexpr (KS > 1) {
        if (!ks || !ks->partials || !ks->flags || ks->partial_stride < (size_t)g.OH * g.OW * g.COUT * (size_t)n_images ||
            ks->n_flags < (size_t)grid.x * grid.y)
            return hipErrorInvalidValue;
        ka = KSplitArgs{ks->partials, (unsigned long long)ks->partial_stride, ks->flags, ks->nonce};
    }
    if constexpr (DECONV) {
        // the tap loop stays a loop where every CU holds two workgroups (full grids), and is unrolled on smaller ones (see
        // SICN_DECONV_TAP_LOOP); the K-split form (an option) is a small-grid form: unrolled.  (Its unrolled build was what exposed
        // the missing barrier behind the prologue's fragment reads — see the comment there.)
        const bool rolled = KS == 1 && (unsigned long long)grid.x * grid.y >= 2ull * (unsigned)chip.n_cu;
        auto go = [&](auto kern) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, in, out, w.d_w_mfma16, w.d_bias, g.IW, g.IH, g.OW, g.OH, tiles_x,
                               tiles_x * tiles_y, n_images, in_layout, out_layout, flags, chip.n_xcd, ka);
            return hipGetLastError();
        };
        if constexpr (KS == 1) {
            if (rolled) return go(&k_deconv_p<NQ, NT16, NTF, TX, PF, KS, true>);
        }
        return go(&k_deconv_p<NQ, NT16, NTF, TX, PF, KS, false>);
    } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_p<NQ, NT16, NTF, TX, PF, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_conv_p<NQ, NT16, NTF, TX, PF, KS>), grid, dim3(256), lds, stream, in, out, w.d_w_mfma16, w.d_bias, g.IW, g.IH, g.OW,
                           g.OH, tiles_x, tiles_x * tiles_y, n_images, in_layout, out_layout, flags, chip.n_xcd, ka);
    }
    return hipGetLastError();
}

// Which (shape, tile width) combinations exist here: every shape at TX = 16 (LDS <= 80 KB), the 128 -> 128 shapes also at TX = 32.
bool pipelined_supported(const LayerGeom &g, int tx)
{
    const bool io = (g.CIN == 128 || g.CIN == 192) && (g.COUT == 128 || g.COUT == 192) && !(g.CIN == 192 && g.COUT == 192);
    if (!io) return false;
    return tx == 16 || (g.CIN == 128 && g.COUT == 128);
}

hipError_t launch_pipelined(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images, hipStream_t stream,
                            int in_layout, int out_layout, bool relu, int tx, bool split, const ChipGeom &chip, int split_k,
                            const KSplitScratch *ks)
{
    if (!pipelined_supported(g, tx)) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB || (size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;
#define SICN_P(NQ, NT, NTF, D, TX) return launch_p<NQ, NT, NTF, D, TX>(g, w, in, out, n_images, stream, in_layout, out_layout, relu, chip)
#define SICN_PK(NQ, NT, NTF, D, TX) return launch_p<NQ, NT, NTF, D, TX, NQ / 2>(g, w, in, out, n_images, stream, in_layout, out_layout, relu, chip, ks)
    // The K split (round 4) measured a LOSS on this chip: its instantiations live in the ALT build only (libsicn_alt.so, with
    // their tests: tests/alt_kernels_check.py); the product library rejects sicn_options.split_k > 1 (SICN_EINVAL).
#ifndef SICN_ALT_KERNELS
    if (split_k > 1) return hipErrorInvalidValue;
    (void)ks;
#else
    if (split && tx == 16 && split_k > 1) {   // 64 output channels and one channel-group pair per workgroup (K split, round 4)
        if (split_k != g.CIN / 64) return hipErrorInvalidValue;
        if (g.transposed) {
            if (g.CIN == 128 && g.COUT == 128) SICN_PK(4, 4, 8, true, 16);
            if (g.CIN == 192 && g.COUT == 128) SICN_PK(6, 4, 8, true, 16);
            if (g.CIN == 128 && g.COUT == 192) SICN_PK(4, 4, 12, true, 16);
        } else {
            if (g.CIN == 128 && g.COUT == 128) SICN_PK(4, 4, 8, false, 16);
            if (g.CIN == 128 && g.COUT == 192) SICN_PK(4, 4, 12, false, 16);
            if (g.CIN == 192 && g.COUT == 128) SICN_PK(6, 4, 8, false, 16);
        }
    }
#endif
    if (split && tx == 16) {   // 64 output channels per workgroup
        if (g.transposed) {
            if (g.CIN == 128 && g.COUT == 128) SICN_P(4, 4, 8, true, 16);
            if (g.CIN == 192 && g.COUT == 128) SICN_P(6, 4, 8, true, 16);
            if (g.CIN == 128 && g.COUT == 192) SICN_P(4, 4, 12, true, 16);
        } else {
            if (g.CIN == 128 && g.COUT == 128) SICN_P(4, 4, 8, false, 16);
            if (g.CIN == 128 && g.COUT == 192) SICN_P(4, 4, 12, false, 16);
            if (g.CIN == 192 && g.COUT == 128) SICN_P(6, 4, 8, false, 16);
        }
    }
    if (g.transposed) {
        if (g.CIN == 128 && g.COUT == 128) {
            if (tx == 32) SICN_P(4, 8, 8, true, 32);
            SICN_P(4, 8, 8, true, 16);
        }
        if (g.CIN == 192 && g.COUT == 128) SICN_P(6, 8, 8, true, 16);
        if (g.CIN == 128 && g.COUT == 192) SICN_P(4, 12, 12, true, 16);
    } else {
        if (g.CIN == 128 && g.COUT == 128) {
            if (tx == 32) SICN_P(4, 8, 8, false, 32);
            SICN_P(4, 8, 8, false, 16);
        }
        if (g.CIN == 128 && g.COUT == 192) SICN_P(4, 12, 12, false, 16);
        if (g.CIN == 192 && g.COUT == 128) SICN_P(6, 8, 8, false, 16);
    }
#undef SICN_P
#undef SICN_PK
    return hipErrorInvalidValue;
}

}  // namespace sicn
