// Implicit-GEMM conv2d<> / deconv522<> for the 32-channel-multiple layers (L1..L6 of the net),
// hand-written for gfx950: int8 MFMA (v_mfma_i32_32x32x32_i8), LDS-resident input patch filled by
// LDS-DMA (buffer_load ... lds, zero padding = the descriptor's range check), weights streamed
// through an LDS ring by LDS-DMA, bias/wrap-mod-256/ReLU fused into the epilogue.
//
// What it replaces (reference, conv_nonsquare_top.cpp):
//   conv2d<>   :198-280  = FMPadding_nonsquare -> width converter -> stride-1 sliding window
//                          (slidingwindow.h:1254-1353) -> keep even rows/cols (:243-259) ->
//                          Matrix_Vector_Activate_Batch (mvau.hpp:87-179) -> bias + ReLU (:267-278)
//   deconv522<>:71-195   = zero-insert (:110-150) -> pad -> sliding window -> MVAU -> bias + ReLU
// None of those streams is materialised here.  The 8-bit wrapping accumulator of the reference
// (mvau.hpp:112 + activations.hpp:127-134) is reproduced by accumulating in int32 and truncating
// once (Z -> Z/256 is a ring homomorphism; nothing is ever clamped).
//
// Decomposition (SURVEY.md §7 design notes, re-derived in DESIGN.md §3):
//   conv, stride 2 : the input is split into 4 parity planes P[a][b][i][j] = in[2i+a][2j+b]; tap
//                    (ky,kx) of output (y,x) reads plane (ky&1,kx&1) at (y+(ky>>1)-1, x+(kx>>1)-1),
//                    i.e. a unit-stride access.  K is walked as 32-channel groups (outer) x 25
//                    taps (inner, ordered plane by plane); the 4 planes of one channel group sit
//                    in LDS and each plane is re-filled with the NEXT channel group as soon as its
//                    taps are done, under the MFMAs of the other planes.
//   deconv         : 4 output phases (py,px); phase (py,px) of input position (y,x) is output
//                    (2y+py,2x+px) and only uses taps ky=py, kx=px (mod 2), reading input
//                    (y+((ky+py)>>1)-1, x+((kx+px)>>1)-1).  The inserted zeros are never touched
//                    (4x fewer MACs than the reference dataflow; skipped terms are exact zeros).
//
// Work split: one workgroup = 4 waves = one M tile of 8 x 32 positions x ALL output channels.
// Wave w owns rows 2w, 2w+1 of the tile (two 32-position MFMA tiles) x NTJ channel tiles of 32.
// MFMA orientation: A operand = weights (row = output channel), B operand = pixels (column =
// position), so a lane's 16 accumulators of one tile are 16 CONSECUTIVE output channels of ONE
// pixel (weight rows are stored permuted: LDS row (a + 4h + 8d) holds channel 16h + 4d + a), and
// the epilogue stores 16 bytes per lane straight from registers.
//
// Tensor layouts: NHWC (ABI) or GROUP = [C/32][H][W][32] (between two layers of a chain), chosen
// per side by run-time flags; see k_common.hpp.
#include <cstdlib>

#include "k_common.hpp"

namespace sicn {

// ---- device helpers -------------------------------------------------------------------------
// (plain function templates on purpose: lambdas inside a __global__ template make hipcc's
//  host-side instantiation of the kernel fail with a silent substitution failure, ROCm 7.2)

template <int TB>
__device__ __forceinline__ void load_wtile(uint8_t *ring, const int8_t *wstream, int tile, int seq, int lane, int w)
{
    constexpr int NPB = TB / 1024, WR = (NPB + 3) / 4;
    const int8_t *src = wstream + (size_t)tile * TB + lane * 16;   // `tile` = index in the weight stream
    uint8_t *dst = ring + (seq % RING) * TB;                       // `seq` = position in this block's K walk
#pragma unroll
    for (int r = 0; r < WR; r++) {
        int piece = r * 4 + w;
        if (piece >= NPB) piece -= 2;  // NPB == 6: waves 2,3 re-load pieces 4,5 (same bytes)
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + piece * 1024), LDS_PTR(dst + piece * 1024), 16, 0, 0);
    }
}

// accumulators start at the bias: register r of tile j is channel j*32 + 16*kh + r
template <int NTJ>
__device__ __forceinline__ void init_acc(v16i (&acc)[2][NTJ], const int8_t *bias, int kh)
{
#pragma unroll
    for (int j = 0; j < NTJ; j++) {
        const v4i b4 = *(const v4i *)(bias + j * 32 + 16 * kh);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int bv = (int)(int8_t)((uint32_t)b4[r >> 2] >> (8 * (r & 3)));
            acc[0][j][r] = bv;
            acc[1][j][r] = bv;
        }
    }
}

// The operand fragments of one K step: 2 pixel fragments (this wave's two 32-position rows) and
// NTJ weight fragments.  Two sets live in registers: the MFMAs of step s run on one while the
// ds_reads of step s+1 fill the other.
template <int NTJ>
struct Frags {
    v4i pf[2];
    v4i wf[NTJ];
};

template <int NTJ>
__device__ __forceinline__ void load_frags(Frags<NTJ> &f, const uint8_t *sub_patch, const uint8_t *wt,
                                           const uint32_t (&wrow)[NTJ], int p_lane, int kh, int oy, int ox)
{
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int p = p_lane + (i + oy) * PATCH_X + ox;
        f.pf[i] = *(const v4i *)(sub_patch + p * 32 + ((((p >> 3) & 1) ^ kh) << 4));
    }
#pragma unroll
    for (int j = 0; j < NTJ; j++) f.wf[j] = *(const v4i *)(wt + wrow[j]);
}

// half H of a step's MFMAs: pixel row H against every weight tile
template <int NTJ, int H>
__device__ __forceinline__ void mma_half(v16i (&acc)[2][NTJ], const Frags<NTJ> &f)
{
#pragma unroll
    for (int j = 0; j < NTJ; j++)
        acc[H][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.wf[j], f.pf[H], acc[H][j], 0, 0, 0);
}

// The software-pipelined K step.  On entry `cur` holds the fragments of step s
// (read during step s-1) and weight tiles <= s+1 are landed for THIS wave's pieces.
//   [LDS-DMA issue]  MFMA half 0  | counted vmcnt + s_barrier: tile s+1 (and every patch piece
//   issued before it) is now visible to all waves | ds_read fragments of step s+1 | MFMA half 1
// so the barrier and the LDS round trip sit between MFMAs of the same wave instead of behind them.
// VARIANT 1 (pipelined): MFMA half 0 | vmcnt + barrier | ds_read fragments of step s+1 | MFMA half 1.
// VARIANT 0 (simple)   : ds_read fragments of step s | all MFMAs | vmcnt + barrier.
// Both are kept for in-process A/B runs (SICN_MFMA_VARIANT): on a power-limited chip the simpler
// stream is not necessarily the slower one.
template <int VMCNT, int EXTRA>
__device__ __forceinline__ void wait_tiles(bool extra)
{
    if (EXTRA > 0 && extra)
        wait_vmcnt<VMCNT + EXTRA>();
    else
        wait_vmcnt<VMCNT>();
}

template <int NTJ, int VMCNT, int VARIANT, int EXTRA = 0>
__device__ __forceinline__ void k_step(v16i (&acc)[2][NTJ], Frags<NTJ> &cur, Frags<NTJ> &nxt,
                                       const uint8_t *cur_sub_patch, const uint8_t *cur_wt, int cur_oy, int cur_ox,
                                       const uint8_t *nxt_sub_patch, const uint8_t *nxt_wt, int nxt_oy, int nxt_ox,
                                       const uint32_t (&wrow)[NTJ], int p_lane, int kh, bool extra = false)
{
    if constexpr (VARIANT == 1) {
        mma_half<NTJ, 0>(acc, cur);
        __builtin_amdgcn_sched_barrier(0);
        wait_tiles<VMCNT, EXTRA>(extra);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frags<NTJ>(nxt, nxt_sub_patch, nxt_wt, wrow, p_lane, kh, nxt_oy, nxt_ox);
        __builtin_amdgcn_sched_barrier(0);
        mma_half<NTJ, 1>(acc, cur);
    } else {
        // the per-lane address parts pass through a volatile asm that follows the previous step's barrier in
        // program order: the fragment reads cannot be scheduled above that barrier (hipcc otherwise hoists
        // them, and tile s is only published by the barrier of step s-1)
        int pl = p_lane;
        uint32_t wr[NTJ];
        asm volatile("" : "+v"(pl));
#pragma unroll
        for (int j = 0; j < NTJ; j++) {
            wr[j] = wrow[j];
            asm volatile("" : "+v"(wr[j]));
        }
        load_frags<NTJ>(cur, cur_sub_patch, cur_wt, wr, pl, kh, cur_oy, cur_ox);
#pragma unroll
        for (int j = 0; j < NTJ; j++) {
            acc[0][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.wf[j], cur.pf[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur.wf[j], cur.pf[1], acc[1][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + NTJ, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTJ, 0);
        wait_tiles<VMCNT, EXTRA>(extra);
        block_barrier();
    }
}

// bias is already in the accumulator: truncate mod 256, relu7, 16 consecutive channels per lane
template <int NTJ>
__device__ __forceinline__ void store_tiles(const v16i (&acc)[2][NTJ], uint8_t *out_img, int OW, int OH, int MW,
                                            int MH, int Y0, int X0, int w, int m, int kh, bool deconv, int py,
                                            int px, int out_layout, int dbg = 0)
{
    constexpr int COUT = NTJ * 32;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int gy = Y0 + 2 * w + i, gx = X0 + m;
        if (gy < MH && gx < MW && !(dbg & 1)) {  // dbg: timing diagnostics only
            const int oy_ = deconv ? 2 * gy + py : gy, ox_ = deconv ? 2 * gx + px : gx;
#pragma unroll
            for (int j = 0; j < NTJ; j++) {
                const v16i a = acc[i][j];
                uint4 v;
                v.x = pack4_relu7(a[0], a[1], a[2], a[3]);
                v.y = pack4_relu7(a[4], a[5], a[6], a[7]);
                v.z = pack4_relu7(a[8], a[9], a[10], a[11]);
                v.w = pack4_relu7(a[12], a[13], a[14], a[15]);
                uint32_t off = tensor_offset(out_layout, oy_, ox_, (uint32_t)j, COUT, OW, OH) + 16 * kh;
                if (dbg & 4) off &= 0xFFFFu;   // diagnostic: all stores land in one 64 KiB window (stay in L2)
                *(uint4 *)(out_img + off) = v;
            }
        }
    }
}

// Everything one K step needs, so that the conv steps can be expanded by template recursion
// (their vmcnt immediates and plane-refresh slots are compile-time functions of the tap index).
template <int NTJ>
struct StepCtx {
    v16i (&acc)[2][NTJ];
    Frags<NTJ> (&fr)[2];
    uint8_t *patch;
    uint8_t *ring;
    const int8_t *wstream;
    const uint32_t (&wrow)[NTJ];
    const uint8_t *in_img;
    int in_img_bytes;
    int p_lane, kh, lane, w;
};

// conv: which plane does step t (0..24, plane order) refresh, if any?  slot i = pieces 4i+w
__host__ __device__ constexpr int refresh_plane(int t)
{
    return (t >= 0 && t < 3) ? 3 : (t >= 9 && t < 12) ? 0 : (t >= 15 && t < 18) ? 1 : (t >= 21 && t < 24) ? 2 : -1;
}
__host__ __device__ constexpr int refresh_slot(int t) { return t < 3 ? t : t < 12 ? t - 9 : t < 18 ? t - 15 : t - 21; }
__host__ __device__ constexpr int has_refresh(int t) { return refresh_plane((t + 50) % 25) >= 0 ? 1 : 0; }
// patch pieces issued in the PF-1 steps up to and including step t (they are younger than the
// weight tile the step waits for)
__host__ __device__ constexpr int refresh_count(int t)
{
    int n = 0;
    for (int k = 0; k < PF - 1; k++) n += has_refresh(t - k);
    return n;
}

// Conv: 50 steps = two channel groups (q0, q0+1) per expansion, so that the fragment set of a step
// is a compile-time function (T & 1) of its index.
template <int NTJ, int VARIANT, int T>
__device__ __forceinline__ void conv_steps(const StepCtx<NTJ> &c, const uint32_t (&poff)[4][3], int q0, uint32_t qstride)
{
    constexpr int TB = NTJ * 32 * KSTEP, WR = (TB / 1024 + 3) / 4;
    constexpr int TT = T % 25;                      // tap index inside the channel group
    constexpr Tap tap = conv_tap(TT);
    constexpr int plane = (tap.ky & 1) * 2 + (tap.kx & 1);
    constexpr Tap nxt = conv_tap((T + 1) % 25);     // the step whose fragments the pipelined variant fetches
    constexpr int nxt_plane = (nxt.ky & 1) * 2 + (nxt.kx & 1);
    const int q = q0 + T / 25;
    const int step = q * 25 + TT;
    // (1) plane refresh: plane 3 takes THIS group's data (it was last used by the previous group's
    //     final taps), planes 0..2 take the NEXT group's as soon as their own taps are done.
    //     A refresh with group == NQ reads past the last group: never consumed.
    constexpr int rp = refresh_plane(TT);
    if constexpr (rp >= 0) {
        constexpr int slot = refresh_slot(TT);
        const int qq = (rp == 3) ? q : q + 1;
        load_piece(c.patch, c.in_img, c.in_img_bytes, rp, slot * 4 + c.w, poff[rp][slot] + (uint32_t)qq * qstride);
    }
    // (2) weight ring top-up (the stream is padded with PF dummy tiles)
    load_wtile<TB>(c.ring, c.wstream, step + PF, step + PF, c.lane, c.w);
    // (3) MFMAs of this step around the barrier that publishes weight tile step+1 (issued 2 steps
    //     ago, before that step's B pieces) and every patch piece issued before it
    k_step<NTJ, (PF - 1) * WR + refresh_count(TT), VARIANT>(
        c.acc, c.fr[T & 1], c.fr[(T + 1) & 1], c.patch + plane * SUB_ALLOC, c.ring + (step % RING) * TB, tap.ky >> 1,
        tap.kx >> 1, c.patch + nxt_plane * SUB_ALLOC, c.ring + ((step + 1) % RING) * TB, nxt.ky >> 1, nxt.kx >> 1,
        c.wrow, c.p_lane, c.kh);
    if constexpr (T + 1 < 50) conv_steps<NTJ, VARIANT, T + 1>(c, poff, q0, qstride);
}

struct DeconvIo {
    uint8_t *out_img;
    const int8_t *bias;
    int OW, OH, MW, MH, Y0, X0, m, out_layout, dbg;
};

__host__ __device__ constexpr int phase_first_tap(int ph) { return ph == 0 ? 0 : ph == 1 ? 9 : ph == 2 ? 15 : 21; }
__host__ __device__ constexpr int phase_taps(int ph) { return (3 - (ph >> 1)) * (3 - (ph & 1)); }

// The 4 deconv phases in the order (ROT, ROT+1, ROT+2, ROT+3) mod 4.  The weight stream is stored in
// phase order 0,1,2,3, so the tile consumed at position `seq` of this walk is a compile-time base plus a
// run-time offset, and the prefetch PF steps ahead may already belong to the next phase of the walk.
template <int NTJ, int NQ, int VARIANT, int ROT>
__device__ __forceinline__ void deconv_phases(const StepCtx<NTJ> &c, const DeconvIo &io)
{
    constexpr int TB = NTJ * 32 * KSTEP, WR = (TB / 1024 + 3) / 4, STEPS = 25 * NQ;
    // first PF tiles of the walk (phase ROT has >= 16 steps >= PF)
#pragma unroll
    for (int s = 0; s < PF; s++) load_wtile<TB>(c.ring, c.wstream, phase_first_tap(ROT) * NQ + s, s, c.lane, c.w);
    wait_vmcnt<0>();
    block_barrier();
    if constexpr (VARIANT == 1)
        load_frags<NTJ>(c.fr[0], c.patch, c.ring, c.wrow, c.p_lane, c.kh, ROT >> 1, ROT & 1);  // step 0
    int seq = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        constexpr int dummy = 0;
        (void)dummy;
        const int ph = (ROT + k) & 3, nph = (ROT + k + 1) & 3;
        const int py = ph >> 1, px = ph & 1;
        const int nkx = 3 - px, ntap = (3 - py) * nkx;
        const int npy = nph >> 1, npx = nph & 1;   // first tap of the next phase of this walk
        const int len = ntap * NQ;
        const int base = phase_first_tap(ph) * NQ;                       // stream tile of local step 0
        const int nbase = k < 3 ? phase_first_tap(nph) * NQ : STEPS;     // next phase, or the zero padding
        init_acc<NTJ>(c.acc, io.bias, c.kh);
#pragma unroll 1
        for (int t = 0; t < ntap; t++) {
            const int iy = t / nkx, ix = t - iy * nkx;
            int noy, nox;   // coordinates of the tap after this one (for the last group's prefetch)
            if (t + 1 < ntap) {
                const int t1 = t + 1, iy1 = t1 / nkx;
                noy = iy1 + py;
                nox = t1 - iy1 * nkx + px;
            } else {
                noy = npy;
                nox = npx;
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int j = t * NQ + q + PF;   // local index of the tile to prefetch
                load_wtile<TB>(c.ring, c.wstream, j < len ? base + j : nbase + (j - len), seq + q + PF, c.lane, c.w);
                const bool last = (q == NQ - 1);
                // the 2*NTJ output stores of the previous phase sit between the awaited weight tile
                // and this step for the first PF-1 steps of a phase: count them, do not wait for them
                k_step<NTJ, (PF - 1) * WR, VARIANT, 2 * NTJ>(c.acc, c.fr[q & 1], c.fr[(q + 1) & 1], c.patch + q * SUB_ALLOC,
                                                             c.ring + ((seq + q) % RING) * TB, iy + py, ix + px,
                                                             c.patch + (last ? 0 : q + 1) * SUB_ALLOC,
                                                             c.ring + ((seq + q + 1) % RING) * TB, last ? noy : iy + py,
                                                             last ? nox : ix + px, c.wrow, c.p_lane, c.kh,
                                                             k > 0 && t == 0 && q < PF - 1);
            }
            seq += NQ;
        }
        if (k == 3) wait_vmcnt<0>();  // the padded tail of the weight prefetch must land before exit
        store_tiles<NTJ>(c.acc, io.out_img, io.OW, io.OH, io.MW, io.MH, io.Y0, io.X0, c.w, io.m, c.kh, true, py, px,
                         io.out_layout, io.dbg);
    }
}

template <int NQ, int NTJ, bool DECONV, int MINW, int VARIANT>
__global__ __launch_bounds__(256, MINW) void k_mfma_t(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ wstream,
    const int8_t *__restrict__ bias, int IW, int IH, int OW, int OH, int MW, int MH, int tiles_x, int in_layout,
    int out_layout, int dbg)
{
    constexpr int CIN = NQ * 32, COUT = NTJ * 32;
    constexpr int NSUB = DECONV ? NQ : 4;
    constexpr int TB = COUT * KSTEP;  // weight tile bytes

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem;
    uint8_t *ring = smem + NSUB * SUB_ALLOC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int img = blockIdx.z;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int Y0 = tile_y * TILE_Y, X0 = tile_x * TILE_X;

    const int in_img_bytes = IH * IW * CIN;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * OH * OW * COUT;

    // weight rows: LDS row = j*32 + m, logical K half kh at physical half kh ^ ((row>>3)&1)
    uint32_t wrow[NTJ];
#pragma unroll
    for (int j = 0; j < NTJ; j++) wrow[j] = (uint32_t)((j * 32 + m) * 32 + ((kh ^ ((m >> 3) & 1)) << 4));

    v16i acc[2][NTJ];
    Frags<NTJ> fr[2];
    const int p_lane = (2 * w) * PATCH_X + m;
    const StepCtx<NTJ> ctx{acc, fr, patch, ring, wstream, wrow, in_img, in_img_bytes, p_lane, kh, lane, w};
    (void)bias;

    if constexpr (DECONV) {
        // ---- prologue: the whole patch (NQ channel groups) + PF weight tiles --------------------
#pragma unroll
        for (int sub = 0; sub < NQ; sub++)
#pragma unroll
            for (int slot = 0; slot < 3; slot++)
                load_piece(patch, in_img, in_img_bytes, sub, slot * 4 + w,
                           piece_src_offset(slot * 4 + w, lane, Y0 - 1, X0 - 1, 1, 0, 0, IW, IH, in_layout,
                                            (uint32_t)sub, CIN));
#pragma unroll
        for (int s = 0; s < PF; s++) load_wtile<TB>(ring, wstream, s, s, lane, w);
        wait_vmcnt<0>();
        block_barrier();
        if constexpr (VARIANT == 1) load_frags<NTJ>(fr[0], patch, ring, wrow, p_lane, kh, 0, 0);  // step 0

        // Phase order: rotation 0 or 2, chosen per workgroup when dbg bit 1 is set (experiment:
        // co-resident workgroups then reach their store bursts at different times).
        const DeconvIo io{out_img, bias, OW, OH, MW, MH, Y0, X0, m, out_layout, dbg};
        if ((dbg & 2) && (blockIdx.x & 1))
            deconv_phases<NTJ, NQ, VARIANT, 2>(ctx, io);
        else
            deconv_phases<NTJ, NQ, VARIANT, 0>(ctx, io);
    } else {
        static_assert(DECONV || NQ % 2 == 0, "conv walks channel groups in pairs");
        // ---- per-lane source offsets of the 4 planes x 3 refresh slots (channel group 0) --------
        uint32_t poff[4][3];
#pragma unroll
        for (int pl = 0; pl < 4; pl++)
#pragma unroll
            for (int slot = 0; slot < 3; slot++)
                poff[pl][slot] = piece_src_offset(slot * 4 + w, lane, Y0 - 1, X0 - 1, 2, pl >> 1, pl & 1, IW, IH,
                                                  in_layout, 0u, CIN);
        const uint32_t qstride = in_layout == LAYOUT_GROUP ? (uint32_t)(IW * IH * 32) : 32u;  // conv: NHWC or GROUP
        // ---- prologue: planes 0..2 of group 0 (plane 3 arrives in steps 0..2) + PF weight tiles --
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int slot = 0; slot < 3; slot++)
                load_piece(patch, in_img, in_img_bytes, pl, slot * 4 + w, poff[pl][slot]);
#pragma unroll
        for (int s = 0; s < PF; s++) load_wtile<TB>(ring, wstream, s, s, lane, w);
        wait_vmcnt<0>();
        block_barrier();
        if constexpr (VARIANT == 1) load_frags<NTJ>(fr[0], patch, ring, wrow, p_lane, kh, 0, 0);  // step 0

        init_acc<NTJ>(acc, bias, kh);
#pragma unroll 1
        for (int q0 = 0; q0 < NQ; q0 += 2) conv_steps<NTJ, VARIANT, 0>(ctx, poff, q0, qstride);
        wait_vmcnt<0>();
        store_tiles<NTJ>(acc, out_img, OW, OH, MW, MH, Y0, X0, w, m, kh, false, 0, 0, out_layout, dbg);
    }
}

// Explicit instantiations: the host stubs of a __global__ template that is only named inside
// another template are not emitted by hipcc (ROCm 7.2) otherwise.
#define SICN_INST(NQ, NTJ, D, V)                                                                            \
    template __global__ void k_mfma_t<NQ, NTJ, D, ((NTJ <= 4 && NQ <= 4) ? 2 : 1), V>(                       \
        const uint8_t *__restrict__, uint8_t *__restrict__, const int8_t *__restrict__,                      \
        const int8_t *__restrict__, int, int, int, int, int, int, int, int, int, int);
SICN_INST(4, 4, true, 0)
SICN_INST(4, 4, true, 1)
SICN_INST(6, 4, true, 0)
SICN_INST(6, 4, true, 1)
SICN_INST(4, 4, false, 0)
SICN_INST(4, 4, false, 1)
SICN_INST(4, 6, false, 0)
SICN_INST(4, 6, false, 1)
#undef SICN_INST

static int mfma_variant() { return debug_env().mfma_variant; }   // environment read once at load (sicn_abi.hip)

template <int NQ, int NTJ, bool DECONV, int VARIANT>
static hipError_t launch_var(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                             int n_images, hipStream_t stream, int in_layout, int out_layout)
{
    constexpr int NSUB = DECONV ? NQ : 4;
    constexpr int MINW = ((NTJ <= 4 && NQ <= 4) ? 2 : 1);
    const int MW = DECONV ? g.IW : g.OW, MH = DECONV ? g.IH : g.OH;
    const int tiles_x = (MW + TILE_X - 1) / TILE_X, tiles_y = (MH + TILE_Y - 1) / TILE_Y;
    size_t lds = (size_t)NSUB * SUB_ALLOC + (size_t)RING * NTJ * 32 * KSTEP;
    lds += (size_t)debug_env().extra_lds;  // occupancy experiments only (SICN_DEBUG_EXTRA_LDS at load time)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mfma_t<NQ, NTJ, DECONV, MINW, VARIANT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(tiles_x * tiles_y), 1, (unsigned)n_images);
    hipLaunchKernelGGL((k_mfma_t<NQ, NTJ, DECONV, MINW, VARIANT>), grid, dim3(256), lds, stream, in, out, w.d_w_mfma,
                       w.d_bias, g.IW, g.IH, g.OW, g.OH, MW, MH, tiles_x, in_layout, out_layout,
                       debug_env().debug_kernel);
    return hipGetLastError();
}

template <int NQ, int NTJ, bool DECONV>
static hipError_t launch_one(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                             int n_images, hipStream_t stream, int in_layout, int out_layout)
{
    return mfma_variant() == 1 ? launch_var<NQ, NTJ, DECONV, 1>(g, w, in, out, n_images, stream, in_layout, out_layout)
                               : launch_var<NQ, NTJ, DECONV, 0>(g, w, in, out, n_images, stream, in_layout, out_layout);
}

// shapes the 32x32x32 kernels of this file are instantiated for (the reference net only)
bool mfma32_supported(int cin, int cout, int transposed)
{
    if (transposed) return (cin == 128 || cin == 192) && cout == 128;
    return cin == 128 && (cout == 128 || cout == 192);
}

hipError_t launch_mfma(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                       int n_images, hipStream_t stream, int in_layout, int out_layout)
{
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB) return hipErrorInvalidValue;          // 31-bit patch offsets
    if ((size_t)g.OH * g.OW * g.COUT >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit store offsets
    if (g.transposed) {
        if (g.CIN == 128 && g.COUT == 128) return launch_one<4, 4, true>(g, w, in, out, n_images, stream, in_layout, out_layout);
        if (g.CIN == 192 && g.COUT == 128) return launch_one<6, 4, true>(g, w, in, out, n_images, stream, in_layout, out_layout);
    } else {
        if (g.CIN == 128 && g.COUT == 128) return launch_one<4, 4, false>(g, w, in, out, n_images, stream, in_layout, out_layout);
        if (g.CIN == 128 && g.COUT == 192) return launch_one<4, 6, false>(g, w, in, out, n_images, stream, in_layout, out_layout);
    }
    return hipErrorInvalidValue;
}

// ---- host-side weight packing --------------------------------------------------------------
// The stream is padded with PF zero tiles: the kernels prefetch PF tiles past the last step.
int mfma_stream_steps(int cin) { return 25 * (cin / 32); }
size_t mfma_stream_bytes(int cin, int cout) { return (size_t)(mfma_stream_steps(cin) + PF) * cout * KSTEP; }

static void pack_tile(const int8_t *w_okc, int cin, int cout, int tap, int q, int8_t *tile)
{
    const int kk = 25 * cin;
    for (int row = 0; row < cout; row++) {
        const int j = row >> 5, rho = row & 31;
        const int ch = j * 32 + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);  // sigma
        const int8_t *src = w_okc + (size_t)ch * kk + tap * cin + q * 32;
        const int g = (row >> 3) & 1;
        for (int h = 0; h < 2; h++)
            for (int b = 0; b < 16; b++) tile[row * 32 + ((h ^ g) << 4) + b] = src[h * 16 + b];
    }
}

void pack_mfma_stream(const int8_t *w_okc, int cin, int cout, int transposed, int8_t *dst)
{
    const int nq = cin / 32;
    const size_t tb = (size_t)cout * KSTEP;
    size_t step = 0;
    if (!transposed) {
        for (int q = 0; q < nq; q++)
            for (int t = 0; t < 25; t++) {
                const Tap tap = conv_tap(t);
                pack_tile(w_okc, cin, cout, tap.ky * 5 + tap.kx, q, dst + (step++) * tb);
            }
    } else {
        for (int ph = 0; ph < 4; ph++) {
            const int py = ph >> 1, px = ph & 1;
            for (int iy = 0; iy < 3 - py; iy++)
                for (int ix = 0; ix < 3 - px; ix++) {
                    const int ky = 2 * iy + py, kx = 2 * ix + px;
                    for (int q = 0; q < nq; q++) pack_tile(w_okc, cin, cout, ky * 5 + kx, q, dst + (step++) * tb);
                }
        }
    }
    for (size_t i = step * tb; i < (step + PF) * tb; i++) dst[i] = 0;
}

}  // namespace sicn
