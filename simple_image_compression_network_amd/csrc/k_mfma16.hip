// conv2d<> / deconv522<> implicit GEMM on v_mfma_i32_16x16x64_i8 — same decomposition, LDS patch,
// weight ring and layouts as k_mfma.hip (read its header first), different MFMA shape.
//
// Why: on this power-limited chip the 32x32x32 int8 MFMA loop tops out at ~2.93 POP/s and the
// 16x16x64 loop at ~3.45 POP/s with identical LDS traffic per MAC (tools/microbench/mfma_shape.hip,
// profiles/r01_microbench_mfma_shape.txt); k_mfma.hip already sits on the 32x32x32 ceiling.
//
// One "pass" = TWO K steps of k_mfma.hip (2 x 32 channel bytes = the 64-deep K of the instruction):
// lanes 0..31 (K bytes 0..31 of the MFMA) read step A's operands, lanes 32..63 step B's.
//   conv  : A, B = two consecutive taps of the plane-ordered walk (possibly two different planes)
//   deconv: A, B = channel groups q, q+1 of the same tap
// Lane roles (l = lane): pos/row = l & 15, g = l >> 4: step = g >> 1, 16-byte half = g & 1.
//   pixel fragment c (NC = 2 * TX/16 per wave: row i = c / (TX/16) of the wave's two rows, column tile c % (TX/16)):
//       16 bytes at patch[sub_step][(2w + i + oy_step) * (TX+2) + 16*(c % (TX/16)) + pos + ox_step][half]
//   weight fragment j (COUT/16 per pass): 16 bytes at ring[step][row j*16 + pos][half]
// Neither image is swizzled: the hardware's 16-lane ds_read_b128 groups pair positions {0-3,12-15}
// of one half with positions {4-11} of the other, which already covers 16 distinct 16-byte slots.
// C/D layout (col = l & 15 = position, row = 4g + r): weight row (4g + r) of tile j holds channel
// 64*(j>>2) + 16g + 4*(j&3) + r, so the four accumulators of tiles 4J..4J+3 of a lane are 16
// consecutive channels of one pixel: one 16-byte store per (column tile, J), no transpose.
// A barrier per pass (not per step) halves the barrier count of k_mfma.hip.
#include <cstdlib>

#include "k_common.hpp"

namespace sicn {

#ifndef SICN_PF16
#define SICN_PF16 6
#endif
#ifndef SICN_RING16
#define SICN_RING16 8
#endif
#ifndef SICN_WAIT_PASSES
#define SICN_WAIT_PASSES 2
#endif
constexpr int PF16 = SICN_PF16;      // weight tiles (K steps) requested ahead of the consumer, even
constexpr int RING16 = SICN_RING16;  // >= PF16 + 2
// A pass needs the tiles of the NEXT pass at its barrier; they were requested PF16/2 = 3 passes
// earlier, so the requests of the last WAITP = 2 passes may still be in flight there (1 = the
// stricter wait of the first version: only the pass's own requests).
constexpr int WAITP = SICN_WAIT_PASSES;
static_assert(WAITP >= 1 && WAITP <= PF16 / 2 - 1, "tiles of pass k+1 were requested in pass k+1-PF16/2");

template <int TB>
__device__ __forceinline__ void load_wtile16(uint8_t *ring, const int8_t *wstream, int tile, int lane, int w)
{
    constexpr int NPB = TB / 1024, WR = (NPB + 3) / 4;
    const int8_t *src = wstream + (size_t)tile * TB + lane * 16;
    uint8_t *dst = ring + (tile % RING16) * TB;
#pragma unroll
    for (int r = 0; r < WR; r++) {
        int piece = r * 4 + w;
        if (piece >= NPB) piece -= 2;  // NPB == 6: waves 2,3 re-load pieces 4,5 (same bytes)
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + piece * 1024), LDS_PTR(dst + piece * 1024), 16, 0, 0);
    }
}

// accumulators start at the bias: register r of tile (c, j) is channel 64*(j>>2) + 16g + 4*(j&3) + r
template <int NT16, int NC>
__device__ __forceinline__ void init_acc16(v4i (&acc)[NC][NT16], const int8_t *bias, int g)
{
#pragma unroll
    for (int J = 0; J < NT16 / 4; J++) {
        const v4i b4 = *(const v4i *)(bias + 64 * J + 16 * g);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            v4i v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = (int)(int8_t)((uint32_t)b4[jj] >> (8 * r));
#pragma unroll
            for (int c = 0; c < NC; c++) acc[c][4 * J + jj] = v;
        }
    }
}

// One pass: fragments of steps A (lanes 0..31) and B (lanes 32..63), then 4 * NT16 MFMAs, then the
// counted wait + barrier that publishes the next pass's weight tiles.
//   pix_off : byte offset of this lane's fragment 0 inside the patch, already including the step
//             (sub-patch, tap offset) selected by the lane's K half
//   wt_off  : byte offset of this lane's fragment 0 inside the ring, including the step's slot
//   FIRST   : the accumulators start here: C operand = the bias (unpacked per weight tile into 4
//             scratch registers) instead of 4 * NT16 * 4 register initialisations before the pass
template <int TX, int NT16, int VMCNT, int EXTRA, bool FIRST = false>
__device__ __forceinline__ void pass16(v4i (&acc)[Geo<TX>::NC][NT16], const uint8_t *patch, const uint8_t *ring, uint32_t pix_off,
                                       uint32_t wt_off, bool extra, const v4i (&bias4)[NT16 / 4] = {})
{
    constexpr int NC = Geo<TX>::NC, XT = Geo<TX>::XT;
    v4i pf[NC], wf[NT16];
    // The fragment addresses pass through a volatile asm that follows the previous pass's barrier in
    // program order, so the reads below cannot be scheduled above that barrier.  (hipcc did hoist them:
    // the LDS reads of pass k+1 were issued before the barrier of pass k, which is only safe while the
    // tiles land a whole pass early.)
    asm volatile("" : "+v"(pix_off), "+v"(wt_off));
#pragma unroll
    for (int c = 0; c < NC; c++) pf[c] = *(const v4i *)(patch + pix_off + ((c / XT) * Geo<TX>::PX + (c % XT) * 16) * 32);
#pragma unroll
    for (int j = 0; j < NT16; j++) wf[j] = *(const v4i *)(ring + wt_off + j * 16 * 32);
#pragma unroll
    for (int j = 0; j < NT16; j++) {
        v4i cin;
        if constexpr (FIRST) {   // register r of tile j is channel 64*(j>>2) + 16g + 4*(j&3) + r: byte r of dword j&3 of bias4[j>>2]
#pragma unroll
            for (int r = 0; r < 4; r++) cin[r] = (int)(int8_t)((uint32_t)bias4[j >> 2][j & 3] >> (8 * r));
        }
#pragma unroll
        for (int c = 0; c < NC; c++)
            acc[c][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[j], pf[c], FIRST ? cin : acc[c][j], 0, 0, 0);
    }
#ifndef SICN_NO_SCHED16
    __builtin_amdgcn_sched_group_barrier(0x100, NC + NT16, 0);   // all fragment reads first
    if constexpr (!FIRST) __builtin_amdgcn_sched_group_barrier(0x008, NC * NT16, 0);   // then the MFMA cluster
#endif
#ifdef SICN_EXP_NOWAIT   // timing experiment only (results are wrong): never block on vmcnt inside the loop
    wait_vmcnt<63>();
#else
    if (EXTRA > 0 && extra)
        wait_vmcnt<VMCNT + EXTRA>();
    else
        wait_vmcnt<VMCNT>();
#endif
#ifndef SICN_EXP_NO_PASS_BARRIER   // timing experiment only (races): what does the per-pass barrier cost?
    block_barrier();
#endif
}

// Always issues NC * NT16/4 stores per wave (positions outside the image go to an out-of-range offset
// of a buffer descriptor, which drops them): the counted waits of the following passes rely on it.
template <int TX, int NT16>
__device__ __forceinline__ void store_tiles16(const v4i (&acc)[Geo<TX>::NC][NT16], uint8_t *out_img, int out_img_bytes,
                                              const TensorMap &om, int MW, int MH, int Y0, int X0, int w, int pos, int g,
                                              bool deconv, int py, int px, uint32_t act_floor)
{
    constexpr int NC = Geo<TX>::NC, XT = Geo<TX>::XT;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, out_img_bytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int gy = Y0 + 2 * w + c / XT, gx = X0 + (c % XT) * 16 + pos;
        const bool ok = gy < MH && gx < MW;
        const int oy_ = deconv ? 2 * gy + py : gy, ox_ = deconv ? 2 * gx + px : gx;
        // channels 64J + 16g .. +15 = channel group 2J + (g>>1), second half iff g odd
        const uint32_t off0 = tensor_offset(om, oy_, ox_, (uint32_t)(g >> 1)) + 16u * (g & 1);
#pragma unroll
        for (int J = 0; J < NT16 / 4; J++) {
            v4i v;
            v[0] = (int)pack4_relu7(acc[c][4 * J + 0][0], acc[c][4 * J + 0][1], acc[c][4 * J + 0][2], acc[c][4 * J + 0][3], act_floor & ACT_FLOOR_MASK);
            v[1] = (int)pack4_relu7(acc[c][4 * J + 1][0], acc[c][4 * J + 1][1], acc[c][4 * J + 1][2], acc[c][4 * J + 1][3], act_floor & ACT_FLOOR_MASK);
            v[2] = (int)pack4_relu7(acc[c][4 * J + 2][0], acc[c][4 * J + 2][1], acc[c][4 * J + 2][2], acc[c][4 * J + 2][3], act_floor & ACT_FLOOR_MASK);
            v[3] = (int)pack4_relu7(acc[c][4 * J + 3][0], acc[c][4 * J + 3][1], acc[c][4 * J + 3][2], acc[c][4 * J + 3][3], act_floor & ACT_FLOOR_MASK);
            // outputs far larger than the caches are stored non-temporal (aux bit 1): measured on 8 x 4K (A/B in one process,
            // profiles/r02_ab_nt_stores.txt) layer 6 -6 % and — its consumer finds less of its own input evicted — layer 7 -8 %
            if (act_floor & ACT_NT_STORE)
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 2);
            else
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 0);
        }
    }
}

// conv: refresh schedule of the 4 parity planes in a 25-step channel group, S = Geo::SLOTS steps per
// plane (one piece per wave and step).  A plane may only be re-filled from the pass AFTER the one that
// holds its last read, whatever the step parity of the group is, i.e. from (last read step + 2):
// plane 0 (last read 8) from step 10, plane 1 (14) from 16, plane 2 (20) from 22, plane 3 (24) from
// step 1 of the NEXT group.
__host__ __device__ constexpr int refresh16_start(int plane) { return plane == 3 ? 1 : plane == 0 ? 10 : plane == 1 ? 16 : 22; }
__host__ __device__ constexpr int refresh16_plane(int t, int S)
{
    for (int pl = 0; pl < 4; pl++)
        if (t >= refresh16_start(pl) && t < refresh16_start(pl) + S) return pl;
    return -1;
}
__host__ __device__ constexpr int refresh16_slot(int t, int S)
{
    const int pl = refresh16_plane(t, S);
    return pl < 0 ? 0 : t - refresh16_start(pl);
}

struct Conv16Ctx {
    uint8_t *patch;
    uint8_t *ring;
    const int8_t *wstream;
    const uint8_t *in_img;
    int in_img_bytes;
    uint32_t lane_pix;   // ((2w) * 34 + pos) * 32 + half * 16
    uint32_t lane_wt;    // pos * 32 + half * 16
    int lane, w, hi;     // hi = this lane serves step B (lanes 32..63)
};

// pass P of a 50-step window (two channel groups q0, q0+1): steps 2P, 2P+1
template <int TX, int NT16, int P>
__device__ __forceinline__ void conv_passes16(v4i (&acc)[Geo<TX>::NC][NT16], const Conv16Ctx &c,
                                              const uint32_t (&poff)[4][Geo<TX>::SLOTS], int q0, uint32_t qstride)
{
    constexpr int PATCH_X = Geo<TX>::PX, SUB_ALLOC = Geo<TX>::ALLOC, S = Geo<TX>::SLOTS;   // this tile width's geometry
    constexpr int TB = NT16 * 16 * KSTEP, WR = (TB / 1024 + 3) / 4;
    constexpr int SA = 2 * P, SB = 2 * P + 1;              // steps inside the 50-step window
    constexpr int TA = SA % 25, TBs = SB % 25;             // tap index inside the channel group
    constexpr Tap tapA = conv_tap(TA), tapB = conv_tap(TBs);
    constexpr int planeA = (tapA.ky & 1) * 2 + (tapA.kx & 1), planeB = (tapB.ky & 1) * 2 + (tapB.kx & 1);
    constexpr uint32_t offA = planeA * SUB_ALLOC + ((tapA.ky >> 1) * PATCH_X + (tapA.kx >> 1)) * 32;
    constexpr uint32_t offB = planeB * SUB_ALLOC + ((tapB.ky >> 1) * PATCH_X + (tapB.kx >> 1)) * 32;
    const int qA = q0 + SA / 25, qB = q0 + SB / 25;
    const int stepA = qA * 25 + TA, stepB = qB * 25 + TBs;   // = stepA + 1
    // (1) plane refresh pieces scheduled for these two steps
    constexpr int rpA = refresh16_plane(TA, S), rpB = refresh16_plane(TBs, S);
    if constexpr (rpA >= 0) {
        constexpr int slot = refresh16_slot(TA, S);
        load_piece<SUB_ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpA, slot * 4 + c.w,
                   poff[rpA][slot] + (uint32_t)((rpA == 3) ? qA : qA + 1) * qstride);
    }
    if constexpr (rpB >= 0) {
        constexpr int slot = refresh16_slot(TBs, S);
        load_piece<SUB_ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpB, slot * 4 + c.w,
                   poff[rpB][slot] + (uint32_t)((rpB == 3) ? qB : qB + 1) * qstride);
    }
    // (2) weight tiles of the pass after next (the stream is padded with PF16 zero tiles)
    load_wtile16<TB>(c.ring, c.wstream, stepA + PF16, c.lane, c.w);
    load_wtile16<TB>(c.ring, c.wstream, stepB + PF16, c.lane, c.w);
    // (3) MFMAs; the wait leaves exactly this pass's own loads in flight
    const uint32_t pix = c.lane_pix + (c.hi ? offB : offA);
    const uint32_t wt = c.lane_wt + (uint32_t)(((c.hi ? stepB : stepA) % RING16) * TB);
    // requests of this pass, plus (WAITP == 2) those of the previous one: steps SA-2, SA-1 (the previous
    // window's last pass for P == 0; before the first window nothing is outstanding, which only helps)
    constexpr int TPa = (SA + 48) % 25, TPb = (SA + 49) % 25;
    constexpr int own = 2 * WR + (rpA >= 0) + (rpB >= 0);
    constexpr int prev = 2 * WR + (refresh16_plane(TPa, S) >= 0) + (refresh16_plane(TPb, S) >= 0);
    pass16<TX, NT16, own + (WAITP - 1) * prev, 0>(acc, c.patch, c.ring, pix, wt, false);
    if constexpr (P + 1 < 25) conv_passes16<TX, NT16, P + 1>(acc, c, poff, q0, qstride);
}

template <int NQ, int NT16, bool DECONV, int MINW, int TX>
__global__ __launch_bounds__(256, MINW) void k_mfma16_t(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ wstream,
    const int8_t *__restrict__ bias, int IW, int IH, int OW, int OH, int MW, int MH, int tiles_x, int n_tiles, int n_images,
    int in_layout, int out_layout, uint32_t act_floor, int n_xcd)
{
    static_assert(NQ % 2 == 0, "channel groups are consumed in pairs");
    constexpr int CIN = NQ * 32, COUT = NT16 * 16;
    constexpr int NSUB = DECONV ? NQ : 4;
    constexpr int TB = COUT * KSTEP, WR = (TB / 1024 + 3) / 4;
    constexpr int PATCH_X = Geo<TX>::PX, SUB_ALLOC = Geo<TX>::ALLOC, SLOTS = Geo<TX>::SLOTS, NC = Geo<TX>::NC;
    constexpr int NSTORE = NC * NT16 / 4;   // output stores per wave and tile (phase)

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem;
    uint8_t *ring = smem + NSUB * SUB_ALLOC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, hi = g >> 1, half = g & 1;
    // logical work list: tile x (fastest), tile y, image; an XCD gets a contiguous range of it, so the
    // tiles that share halo pixels share an L2 (k_common.hpp)
    const int item = xcd_logical_index(n_tiles * n_images, n_xcd);
    if (item < 0) return;   // before any LDS-DMA is issued
    const int img = item / n_tiles, tile = item - img * n_tiles;
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int Y0 = tile_y * TILE_Y, X0 = tile_x * TX;

    const int in_img_bytes = IH * IW * CIN;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * OH * OW * COUT;
    const uint32_t lane_pix = (uint32_t)(((2 * w) * PATCH_X + pos) * 32 + half * 16);
    const uint32_t lane_wt = (uint32_t)(pos * 32 + half * 16);

    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);
    const int out_img_bytes = OH * OW * COUT;

    v4i acc[NC][NT16];

    if constexpr (DECONV) {
        // ---- prologue: the whole patch (NQ channel groups) + PF16 weight tiles -------------------
#pragma unroll
        for (int slot = 0; slot < SLOTS; slot++) {
            const PieceSrc ps = piece_src<TX>(im, slot * 4 + w, lane, Y0 - 1, X0 - 1, 1, 0, 0, IW, IH);
#pragma unroll
            for (int sub = 0; sub < NQ; sub++)
                load_piece<SUB_ALLOC>(patch, in_img, in_img_bytes, sub, slot * 4 + w, ps.ok ? ps.off + (uint32_t)sub * im.grp : OOB);
        }
#pragma unroll
        for (int s = 0; s < PF16; s++) load_wtile16<TB>(ring, wstream, s, lane, w);
        // this lane's 16 bias bytes per group of 4 weight tiles (channels 64J + 16g .. +15)
        v4i bias4[NT16 / 4];
#pragma unroll
        for (int J = 0; J < NT16 / 4; J++) bias4[J] = *(const v4i *)(bias + 64 * J + 16 * g);
        wait_vmcnt<0>();
        block_barrier();

        int step = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
            const int py = ph >> 1, px = ph & 1;
            const int nkx = 3 - px, ntap = (3 - py) * nkx;
#pragma unroll 1
            for (int t = 0; t < ntap; t++) {
                const int iy = t / nkx, ix = t - iy * nkx;
                const uint32_t tap_off = (uint32_t)(((iy + py) * PATCH_X + ix + px) * 32);
                int q0 = 0;
                if (t == 0) {
                    // first pass of the phase: the accumulators start at the bias (C operand).  The previous
                    // phase's NT16 stores are younger than the awaited tiles in the first WAITP passes of a
                    // phase — count them instead of waiting for them
                    load_wtile16<TB>(ring, wstream, step + PF16, lane, w);
                    load_wtile16<TB>(ring, wstream, step + 1 + PF16, lane, w);
                    const uint32_t pix = lane_pix + tap_off + (uint32_t)(hi * SUB_ALLOC);
                    const uint32_t wt = lane_wt + (uint32_t)(((step + hi) % RING16) * TB);
                    pass16<TX, NT16, 2 * WR * WAITP, NSTORE, true>(acc, patch, ring, pix, wt, ph > 0, bias4);
                    q0 = 2;
                }
#pragma unroll NQ <= 4 ? 2 : 1
                for (int q = q0; q < NQ; q += 2) {
                    load_wtile16<TB>(ring, wstream, step + q + PF16, lane, w);
                    load_wtile16<TB>(ring, wstream, step + q + 1 + PF16, lane, w);
                    const uint32_t pix = lane_pix + tap_off + (uint32_t)((q + hi) * SUB_ALLOC);
                    const uint32_t wt = lane_wt + (uint32_t)(((step + q + hi) % RING16) * TB);
                    pass16<TX, NT16, 2 * WR * WAITP, NSTORE>(acc, patch, ring, pix, wt, ph > 0 && t == 0 && q < 2 * WAITP);
                }
                step += NQ;
            }
            if (ph == 3) wait_vmcnt<0>();  // the padded tail of the weight prefetch must land before exit
            store_tiles16<TX, NT16>(acc, out_img, out_img_bytes, om, MW, MH, Y0, X0, w, pos, g, true, py, px, act_floor);
        }
    } else {
        // ---- per-lane source offsets of the 4 planes x 3 refresh slots (channel group 0) --------
        uint32_t poff[4][SLOTS];
#pragma unroll
        for (int pl = 0; pl < 4; pl++)
#pragma unroll
            for (int slot = 0; slot < SLOTS; slot++) {
                const PieceSrc ps = piece_src<TX>(im, slot * 4 + w, lane, Y0 - 1, X0 - 1, 2, pl >> 1, pl & 1, IW, IH);
                poff[pl][slot] = ps.ok ? ps.off : OOB;
            }
        const uint32_t qstride = im.grp;   // next channel group
        // ---- prologue: planes 0..2 of group 0 (plane 3 arrives in steps 1..3) + PF16 weight tiles -
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int slot = 0; slot < SLOTS; slot++) load_piece<SUB_ALLOC>(patch, in_img, in_img_bytes, pl, slot * 4 + w, poff[pl][slot]);
#pragma unroll
        for (int s = 0; s < PF16; s++) load_wtile16<TB>(ring, wstream, s, lane, w);
        wait_vmcnt<0>();
        block_barrier();

        const Conv16Ctx ctx{patch, ring, wstream, in_img, in_img_bytes, lane_pix, lane_wt, lane, w, hi};
        init_acc16<NT16, NC>(acc, bias, g);
#pragma unroll 1
        for (int q0 = 0; q0 < NQ; q0 += 2) conv_passes16<TX, NT16, 0>(acc, ctx, poff, q0, qstride);
        wait_vmcnt<0>();
        store_tiles16<TX, NT16>(acc, out_img, out_img_bytes, om, MW, MH, Y0, X0, w, pos, g, false, 0, 0, act_floor);
    }
}

// workgroups per CU the registers / LDS of a variant allow: 8 x 32 tiles of the 192-channel layers only 1
constexpr int minw16(int NQ, int NT16, int TX) { return (TX == 16 || (NT16 <= 8 && NQ <= 4)) ? 2 : 1; }

#define SICN_INST16(NQ, NT16, D, TX)                                                                             \
    template __global__ void k_mfma16_t<NQ, NT16, D, minw16(NQ, NT16, TX), TX>(                                    \
        const uint8_t *__restrict__, uint8_t *__restrict__, const int8_t *__restrict__, const int8_t *__restrict__, \
        int, int, int, int, int, int, int, int, int, int, int, uint32_t, int);
SICN_INST16(4, 8, true, 32)
SICN_INST16(6, 8, true, 32)
SICN_INST16(4, 8, false, 32)
SICN_INST16(4, 12, false, 32)
SICN_INST16(4, 8, true, 16)
SICN_INST16(6, 8, true, 16)
SICN_INST16(4, 8, false, 16)
SICN_INST16(4, 12, false, 16)
// shapes of the hyperprior stacks (extension): conv 192 -> 128, deconv 128 -> 192
SICN_INST16(6, 8, false, 32)
SICN_INST16(6, 8, false, 16)
SICN_INST16(4, 12, true, 32)
SICN_INST16(4, 12, true, 16)
#undef SICN_INST16

template <int NQ, int NT16, bool DECONV, int TX>
static hipError_t launch16_tx(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images,
                              hipStream_t stream, int in_layout, int out_layout, bool relu, const ChipGeom &chip)
{
    constexpr int NSUB = DECONV ? NQ : 4;
    constexpr int MINW = minw16(NQ, NT16, TX);
    const int MW = DECONV ? g.IW : g.OW, MH = DECONV ? g.IH : g.OH;
    const int tiles_x = (MW + TX - 1) / TX, tiles_y = (MH + TILE_Y - 1) / TILE_Y;
    const size_t lds = (size_t)NSUB * Geo<TX>::ALLOC + (size_t)RING16 * NT16 * 16 * KSTEP;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mfma16_t<NQ, NT16, DECONV, MINW, TX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid(xcd_grid_size((long)tiles_x * tiles_y * n_images, chip.n_xcd));
    hipLaunchKernelGGL((k_mfma16_t<NQ, NT16, DECONV, MINW, TX>), grid, dim3(256), lds, stream, in, out, w.d_w_mfma16, w.d_bias,
                       g.IW, g.IH, g.OW, g.OH, MW, MH, tiles_x, tiles_x * tiles_y, n_images, in_layout, out_layout,
                       (relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW) | (nt_store_wanted((size_t)g.OH * g.OW * g.COUT * n_images) ? ACT_NT_STORE : 0u),
                       chip.n_xcd);
    return hipGetLastError();
}

// Which kernel family, tile width and splits a conv / deconv layer runs with (pure: sicn_debug_plan walks it without a GPU).
//   family 2: the wide persistent form (k_mfma16x.hip: one workgroup of 4 waves per CU, 16 x 32 positions, 128 x 128 outputs per
//             wave) where every CU gets at least WIDE_MIN_TILES_PER_CU tiles; sicn_options.wave_tile = 128 forces it, 64 forbids it
//   family 1: the software-pipelined 8 x 16 / 8 x 32 kernels (k_mfma16p.hip), the default wherever they exist
//   family 0: the plain kernels of this file
// Tile width: 8 x 32 positions by default; 8 x 16 where the wide tile allows only one workgroup per CU (the 192-channel layers)
// or leaves most of the chip without a tile (small images).  sicn_options.tile_x = 16 | 32 forces one (experiments, tests).
MfmaPlan plan_mfma(const LayerGeom &g, int n_images, const sicn_options &o, const ChipGeom &chip)
{
    const bool deconv = g.transposed != 0;
    const int nq = g.CIN / 32, nt16 = g.COUT / 16;
    const int MW = deconv ? g.IW : g.OW, MH = deconv ? g.IH : g.OH;
    MfmaPlan p{0, 32, 1, 1, 0, 1, 1};
    if (wide_supported(g) && o.tile_x != 16) {
        const long tiles_w = (long)((MW + 31) / 32) * ((MH + 15) / 16) * n_images;
        // measured r03 (tools/ab_options.py, 1080p x 4 and 4K x 1 = 1020 tiles of 256 CUs): layer 1 131 - 140 against 144 - 149 us;
        // at 255 / 510 tiles the wide conv loses, 44 / 39 and 70 / 68 us; the deconv is a wash below 1024: 45 / 47, 75 / 75, 146 / 142 us
        if (o.wave_tile == 128 || (o.wave_tile == 0 && o.prefetch == 0 && wide_automatic(tiles_w, deconv, chip))) {
            p.family = 2;
            p.grid_x = wide_grid(tiles_w, o.persistent_grid, chip);
            p.deal = wide_deal_pays(tiles_w, p.grid_x, chip);
            return p;
        }
    }
    const long tiles32 = (long)((MW + 31) / 32) * ((MH + TILE_Y - 1) / TILE_Y) * n_images;
    // 8 x 32 tiles from about 0.8 of one residency (2 workgroups per CU) on: below that the 8 x 16 tiles' second, partly filled
    // round is still cheaper; measured r03 on one image of 1440 x 810 ... 1920 x 1080 (312 ... 506 tiles of 8 x 32), layer 1 / 6:
    // 8 x 16 tiles 30 31 | 38 39 39 / 39 40 | 47 47 46 us, 8 x 32 tiles 35 35 | 35 36 36 / 40 40 | 41 42 41 us (the bar at 434 tiles)
    bool narrow = minw16(nq, nt16, 32) == 1 || narrow_tile_wanted(tiles32, chip);
    if (o.tile_x == 16) narrow = true;
    if (o.tile_x == 32) narrow = false;
    p.tile_x = narrow ? 16 : 32;
    const long tiles = (long)((MW + p.tile_x - 1) / p.tile_x) * ((MH + TILE_Y - 1) / TILE_Y) * n_images;
    p.grid_x = xcd_grid_size(tiles, chip.n_xcd);
    if (o.prefetch != 1 && pipelined_supported(g, p.tile_x)) {
        p.family = 1;
        // grids that leave half of the CUs without a workgroup: split the output channels over 2 / 3 workgroups
        // (measured, r02: at 192 - 255 tiles the split is a wash or a loss; at <= 72 it takes 20 - 35 % off the layer)
        if (p.tile_x == 16 && (o.split_n > 1 || (o.split_n == 0 && split_n_automatic(tiles, chip)))) p.split_n = nt16 / 4;
        // ... and the ones that are still small then: split K as well (round 4; only the channel-split 8 x 16 kernels have the form)
        if (p.split_n > 1) {
            const size_t out_bytes = (size_t)g.OH * g.OW * g.COUT * n_images;
            int ks = o.split_k > 1 ? nq / 2 : (o.split_k == 0 ? split_k_automatic(tiles * p.split_n, nq, deconv, out_bytes, chip) : 1);
            p.split_k = ks;
        }
        p.grid_y = p.split_n;
        p.grid_z = p.split_k;
    }
    return p;
}

template <int NQ, int NT16, bool DECONV>
static hipError_t launch16(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images,
                           hipStream_t stream, int in_layout, int out_layout, const sicn_options &o, const ChipGeom &chip, bool relu,
                           const KSplitScratch *ks)
{
    MfmaPlan p = plan_mfma(g, n_images, o, chip);
    if (p.family == 2) return launch_wide(g, w, in, out, n_images, stream, in_layout, out_layout, relu, o.persistent_grid, chip, ks ? ks->deal : nullptr);
    if (p.family == 1) {
        // (round 2 sent the deconv 192 -> 128 on full grids back to the plain kernel: the pipelined one was 7 % slower there.  The
        // reason was the v_mov copies hipcc made for its run-time buffer parity — right around the asm MFMAs, where
        // tools/isa_hazards.py found them; with the parity static the pipelined form is 17 % FASTER: layer 4 0.158 -> 0.131 ms.)
        if (p.split_k > 1 && !ksplit_scratch_fits(g, n_images, p, ks)) p.split_k = 1;   // no scratch (single-layer entry points): unsplit
        return launch_pipelined(g, w, in, out, n_images, stream, in_layout, out_layout, relu, p.tile_x, p.split_n > 1, chip, p.split_k, ks);
    }
    return p.tile_x == 16 ? launch16_tx<NQ, NT16, DECONV, 16>(g, w, in, out, n_images, stream, in_layout, out_layout, relu, chip)
                          : launch16_tx<NQ, NT16, DECONV, 32>(g, w, in, out, n_images, stream, in_layout, out_layout, relu, chip);
}

// shapes the 16x16x64 kernels serve: the reference net's L1-L6 plus the hyperprior stacks' conv 192 -> 128 and deconv 128 -> 192
bool mfma_supported(int cin, int cout, int transposed)
{
    (void)transposed;
    return (cin == 128 || cin == 192) && (cout == 128 || cout == 192) && !(cin == 192 && cout == 192);
}

hipError_t launch_mfma16(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images,
                         hipStream_t stream, int in_layout, int out_layout, const sicn_options &o, const ChipGeom &chip, bool relu,
                         const KSplitScratch *ks)
{
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB) return hipErrorInvalidValue;          // 31-bit patch offsets
    if ((size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;         // buffer-descriptor stores
    if (g.transposed) {
        if (g.CIN == 128 && g.COUT == 128) return launch16<4, 8, true>(g, w, in, out, n_images, stream, in_layout, out_layout, o, chip, relu, ks);
        if (g.CIN == 192 && g.COUT == 128) return launch16<6, 8, true>(g, w, in, out, n_images, stream, in_layout, out_layout, o, chip, relu, ks);
        if (g.CIN == 128 && g.COUT == 192) return launch16<4, 12, true>(g, w, in, out, n_images, stream, in_layout, out_layout, o, chip, relu, ks);
    } else {
        if (g.CIN == 192 && g.COUT == 128) return launch16<6, 8, false>(g, w, in, out, n_images, stream, in_layout, out_layout, o, chip, relu, ks);
        if (g.CIN == 128 && g.COUT == 128) return launch16<4, 8, false>(g, w, in, out, n_images, stream, in_layout, out_layout, o, chip, relu, ks);
        if (g.CIN == 128 && g.COUT == 192) return launch16<4, 12, false>(g, w, in, out, n_images, stream, in_layout, out_layout, o, chip, relu, ks);
    }
    return hipErrorInvalidValue;
}

// ---- host-side weight packing: the same tile sequence as pack_mfma_stream, rows in the 16x16 C/D
// ---- order (LDS row j*16 + rho holds channel 64*(j>>2) + 16*(rho>>2) + 4*(j&3) + (rho&3)), no swizzle
// zero tiles behind the stream: the deepest prefetch of any kernel that walks it (k_mfma16: PF16 = 6, k_mfma16w: 8, k_mfma16p: 12)
constexpr int PAD16 = 24;
static_assert(PAD16 >= PF16, "prefetch would run off the stream");
size_t mfma16_stream_bytes(int cin, int cout) { return (size_t)(25 * (cin / 32) + PAD16) * cout * KSTEP; }

static void pack_tile16(const int8_t *w_okc, int cin, int cout, int tap, int q, int8_t *tile)
{
    const int kk = 25 * cin;
    for (int row = 0; row < cout; row++) {
        const int j = row >> 4, rho = row & 15;
        const int ch = 64 * (j >> 2) + 16 * (rho >> 2) + 4 * (j & 3) + (rho & 3);
        const int8_t *src = w_okc + (size_t)ch * kk + tap * cin + q * 32;
        for (int b = 0; b < 32; b++) tile[row * 32 + b] = src[b];
    }
}

void pack_mfma16_stream(const int8_t *w_okc, int cin, int cout, int transposed, int8_t *dst)
{
    const int nq = cin / 32;
    const size_t tb = (size_t)cout * KSTEP;
    size_t step = 0;
    if (!transposed) {
        for (int q = 0; q < nq; q++)
            for (int t = 0; t < 25; t++) {
                const Tap tap = conv_tap(t);
                pack_tile16(w_okc, cin, cout, tap.ky * 5 + tap.kx, q, dst + (step++) * tb);
            }
    } else {
        for (int ph = 0; ph < 4; ph++) {
            const int py = ph >> 1, px = ph & 1;
            for (int iy = 0; iy < 3 - py; iy++)
                for (int ix = 0; ix < 3 - px; ix++) {
                    const int ky = 2 * iy + py, kx = 2 * ix + px;
                    for (int q = 0; q < nq; q++) pack_tile16(w_okc, cin, cout, ky * 5 + kx, q, dst + (step++) * tb);
                }
        }
    }
    for (size_t i = step * tb; i < (step + PAD16) * tb; i++) dst[i] = 0;
}

}  // namespace sicn
