// deconv522<> 128 -> 128 channels (layers 5 and 6) — k_mfma16.hip's deconvolution with an explicit software pipeline.
// Same decomposition (4 output phases, 8 x 32 input positions per workgroup, wave w = rows 2w, 2w+1, two workgroups per CU
// = two waves per SIMD), same LDS images, same weight stream; read k_mfma16.hip first.
//
// What changes: in k_mfma16 a pass is  fragment reads -> 32 MFMAs -> counted wait + barrier,  and the only thing that hides a
// wave's LDS round trip and barrier is its partner wave (measured MFMA pipe utilisation 0.61 on layer 6).  The fragments
// cannot simply be double-buffered there: 128 accumulators + 2 x 48 fragment registers + addressing do not fit 256 VGPRs
// (and hipcc splits a 256-register budget 128 / 128 between VGPRs and AGPRs as soon as an AGPR is used).  Here
//   * the MFMAs are inline-asm statements with the accumulator tied ("+v"): volatile, so the order written is the order
//     issued, and hipcc's scheduler no longer needs room to move them;
//   * weight fragment j is RE-READ for the next pass into its own registers right after its 4 MFMAs (it is dead by then): no
//     second set; only the 4 pixel fragments (used by all 8 weight tiles of the pass) are double-buffered (+16 VGPRs);
//   * the barrier that ends pass k therefore publishes the weight tiles of pass k+2: the ring runs 4 passes ahead (PFP = 8).
// Hazards hipcc cannot see inside asm are covered by hand (s_nop before the epilogue reads the accumulators).
#include "k_common.hpp"

namespace sicn {

constexpr int PFP = 8;    // weight tiles (K steps) requested ahead: 4 passes
constexpr int RINGP = 8;
constexpr int TBP = 128 * KSTEP;

#define SICN_MFMA_V(ACC, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define SICN_MFMA_V_C(ACC, A, B, C) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(ACC) : "v"(A), "v"(B), "v"(C))

__device__ __forceinline__ void load_wtile_p(uint8_t *ring, const int8_t *wstream, int tile, int lane, int w)
{
    const int8_t *src = wstream + (size_t)tile * TBP + lane * 16 + w * 1024;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(ring + (tile % RINGP) * TBP + w * 1024), 16, 0, 0);
}

// One pass: 32 MFMAs (weight tile j outer, the wave's 4 column tiles inner) on the weight fragments `wf` and the CURRENT pixel
// fragments `pc`; the fragments of the NEXT pass are fetched in between: weight fragment j is re-read into its own registers
// right after its 4 MFMAs (it is dead by then), the 4 pixel fragments go to the other buffer `pn`.  Then counted wait + barrier.
//   FIRST: the accumulators start here, C operand = the bias
//   dma(): this pass's LDS-DMA requests; issued behind the first MFMA group, so that the MFMA pipe starts right after the
//   barrier instead of waiting for ~10-20 address / request instructions (their targets are free for the whole pass)
template <int VMCNT, int EXTRA, bool FIRST, typename Dma>
__device__ __forceinline__ void pass_p(v4i (&acc)[4][8], v4i (&wf)[8], const v4i (&pc)[4], v4i (&pn)[4], const uint8_t *pix_next,
                                       const uint8_t *wt_next, bool extra, const v4i (&bias4)[2], Dma dma)
{
    constexpr int PX = Geo<32>::PX;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        v4i cin;
        if constexpr (FIRST) {
#pragma unroll
            for (int r = 0; r < 4; r++) cin[r] = (int)(int8_t)((uint32_t)bias4[j >> 2][j & 3] >> (8 * r));
            // VALU write -> MFMA read of the same VGPR needs wait states hipcc would insert for a builtin MFMA but cannot for
            // an asm one (measured: without them the group's first MFMA saw a stale cin[3])
            asm volatile("s_nop 3" : "+v"(cin));
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if constexpr (FIRST)
                SICN_MFMA_V_C(acc[c][j], wf[j], pc[c], cin);
            else
                SICN_MFMA_V(acc[c][j], wf[j], pc[c]);
        }
        // An MFMA reads its C operand a few cycles AFTER it issues, and hipcc cannot see through the asm that `cin` is an
        // MFMA operand: any VALU write into those 4 registers right behind the group (the next group's unpacking, address
        // arithmetic of the next pass) would be a write-after-read hazard.  This statement keeps `cin` live — so its registers
        // cannot be given away earlier — and spends the wait states.
        if constexpr (FIRST) asm volatile("s_nop 7" ::"v"(cin));
        wf[j] = *(const v4i *)(wt_next + j * 16 * 32);
        if (j < 4) pn[j] = *(const v4i *)(pix_next + ((j / 2) * PX + (j % 2) * 16) * 32);
        if (j == 0) dma();
    }
    if (EXTRA > 0 && extra)
        wait_vmcnt<VMCNT + EXTRA>();
    else
        wait_vmcnt<VMCNT>();
    block_barrier();
}

__global__ __launch_bounds__(256, 2) void k_deconv128p(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                       const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias, int IW, int IH,
                                                       int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout,
                                                       int out_layout, uint32_t act_floor)
{
    constexpr int NQ = 4, CIN = 128, COUT = 128, NT16 = 8, NC = 4, XT = 2;
    constexpr int PX = Geo<32>::PX, ALLOC = Geo<32>::ALLOC, SLOTS = Geo<32>::SLOTS;
    constexpr int NSTORE = NC * NT16 / 4;   // output stores per wave and phase
    static_assert(Geo<32>::PIX * KSTEP <= 11 * 1024, "the bias would overwrite patch positions");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem, *ring = smem + NQ * ALLOC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, hi = g >> 1, half = g & 1;
    const int item = xcd_logical_index(n_tiles * n_images);
    if (item < 0) return;
    const int img = item / n_tiles, tile = item - img * n_tiles;
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int Y0 = tile_y * TILE_Y, X0 = tile_x * 32;
    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * out_img_bytes;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);

    // this lane's fragment 0: pixel (sub-patch of its K half, rows 2w..), weight (ring slot 0)
    const uint8_t *lane_pix = patch + (uint32_t)(((2 * w) * PX + pos) * 32 + half * 16 + hi * ALLOC);
    const uint8_t *lane_wt = ring + (uint32_t)(pos * 32 + half * 16);

    // ---- prologue: the whole patch (NQ channel groups) + PFP weight tiles -------------------------------------------------
#pragma unroll
    for (int slot = 0; slot < SLOTS; slot++) {
        const PieceSrc ps = piece_src<32>(im, slot * 4 + w, lane, Y0 - 1, X0 - 1, 1, 0, 0, IW, IH);
#pragma unroll
        for (int sub = 0; sub < NQ; sub++)
            load_piece<ALLOC>(patch, in_img, in_img_bytes, sub, slot * 4 + w, ps.ok ? ps.off + (uint32_t)sub * im.grp : OOB);
    }
#pragma unroll
    for (int s = 0; s < PFP; s++) load_wtile_p(ring, wstream, s, lane, w);
    // the bias goes to LDS (the 1 KiB behind the last position of sub-patch 0 only ever holds padding): reading it from global
    // memory at the start of every phase would make hipcc drain vmcnt there, i.e. wait for the previous phase's stores
    constexpr int BIAS_LDS = 11 * 1024;
    uint32_t bias_dw = 0;
    if (tid < COUT / 4) bias_dw = ((const uint32_t *)bias)[tid];
    wait_vmcnt<0>();
    if (tid < COUT / 4) ((uint32_t *)(patch + BIAS_LDS))[tid] = bias_dw;
    block_barrier();

    // pass numbering: step = 4 * (taps so far) + q; a pass covers steps (step + q, step + q + 1), q in {0, 2}
    // address of the fragments of the pass (tap offset `toff`, channel pair q, first step s0):
    //   pixel : lane_pix + toff + q * ALLOC                    (the lane's K half already selects sub-patch q + hi)
    //   weight: lane_wt + ((s0 + hi) % RINGP) * TBP
    v4i wf[8], pa[4], pb[4];
    v4i acc[NC][NT16];
    int step = 0;
    {   // fragments of the first pass: phase 0, tap 0 (offset 0), q = 0, steps 0 / 1
#pragma unroll
        for (int r = 0; r < 8; r++) wf[r] = *(const v4i *)(lane_wt + hi * TBP + r * 16 * 32);
#pragma unroll
        for (int r = 0; r < 4; r++) pa[r] = *(const v4i *)(lane_pix + ((r / 2) * PX + (r % 2) * 16) * 32);
    }
#pragma unroll
    for (int ph = 0; ph < 4; ph++) {
        const int py = ph >> 1, px = ph & 1;
        const int nkx = 3 - px, ntap = (3 - py) * nkx;
#pragma unroll 1
        for (int t = 0; t < ntap; t++) {
            const int iy = t / nkx, ix = t - iy * nkx;
            const uint32_t toff = (uint32_t)(((iy + py) * PX + ix + px) * 32);
            // the tap after this one (the next phase's first tap at the end of a phase; anything valid at the very end)
            uint32_t toff_next;
            if (t + 1 < ntap) {
                const int t1 = t + 1, iy1 = t1 / nkx, ix1 = t1 - iy1 * nkx;
                toff_next = (uint32_t)(((iy1 + py) * PX + ix1 + px) * 32);
            } else {
                const int ph1 = (ph + 1) & 3;
                toff_next = (uint32_t)((((ph1 >> 1)) * PX + (ph1 & 1)) * 32);
            }
            // pass A: steps step, step+1 (q = 0) on pa; fetches pb / wf for steps step+2, step+3 (q = 2), same tap
            {
                const uint8_t *pixn = lane_pix + toff + 2u * ALLOC;
                const uint8_t *wtn = lane_wt + (uint32_t)(((step + 2 + hi) % RINGP) * TBP);
                auto dma = [&]() {
                    load_wtile_p(ring, wstream, step + PFP, lane, w);
                    load_wtile_p(ring, wstream, step + 1 + PFP, lane, w);
                };
                if (t == 0) {   // the accumulators start at the bias: this lane's 16 bytes per group of 4 weight tiles
                    v4i bias4[2];
#pragma unroll
                    for (int J = 0; J < 2; J++) bias4[J] = *(const v4i *)(patch + BIAS_LDS + 64 * J + 16 * g);
                    // the previous phase's NSTORE stores are younger than the tiles awaited in the first two passes of a
                    // phase: they are counted, not waited for
                    pass_p<4, NSTORE, true>(acc, wf, pa, pb, pixn, wtn, ph > 0, bias4, dma);
                } else {
                    const v4i none[2] = {};
                    pass_p<4, NSTORE, false>(acc, wf, pa, pb, pixn, wtn, false, none, dma);
                }
            }
            // pass B: steps step+2, step+3 (q = 2) on pb; fetches pa / wf for the next tap's q = 0
            {
                auto dma = [&]() {
                    load_wtile_p(ring, wstream, step + 2 + PFP, lane, w);
                    load_wtile_p(ring, wstream, step + 3 + PFP, lane, w);
                };
                const uint8_t *pixn = lane_pix + toff_next;
                const uint8_t *wtn = lane_wt + (uint32_t)(((step + 4 + hi) % RINGP) * TBP);
                const v4i none[2] = {};
                pass_p<4, NSTORE, false>(acc, wf, pb, pa, pixn, wtn, ph > 0 && t == 0, none, dma);
            }
            step += 4;
        }
        if (ph == 3) wait_vmcnt<0>();  // the padded tail of the weight prefetch must land before exit
        // ---- epilogue of the phase (as store_tiles16 of k_mfma16.hip) ------------------------------------------------------
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last MFMAs' results -> VALU
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int j = 0; j < NT16; j++) asm volatile("" : "+v"(acc[c][j]));   // keeps the packs below behind the s_nop
        __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, out_img_bytes, 0x00020000);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int gy = Y0 + 2 * w + c / XT, gx = X0 + (c % XT) * 16 + pos;
            const bool ok = gy < IH && gx < IW;
            const uint32_t off0 = tensor_offset(om, 2 * gy + py, 2 * gx + px, (uint32_t)(g >> 1)) + 16u * (g & 1);
#pragma unroll
            for (int J = 0; J < NT16 / 4; J++) {
                v4i v;
#pragma unroll
                for (int d = 0; d < 4; d++)
                    v[d] = (int)pack4_relu7(acc[c][4 * J + d][0], acc[c][4 * J + d][1], acc[c][4 * J + d][2], acc[c][4 * J + d][3],
                                            act_floor & ACT_FLOOR_MASK);
                // this kernel only serves full-size grids, whose outputs never fit a cache: always non-temporal
                // (profiles/r02_ab_nt_stores.txt)
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 2);
            }
        }
    }
}

bool deconv128p_supported(const LayerGeom &g) { return g.transposed && g.CIN == 128 && g.COUT == 128; }

hipError_t launch_deconv128p(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images,
                             hipStream_t stream, int in_layout, int out_layout, bool relu)
{
    if (!deconv128p_supported(g)) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB || (size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;
    const int tiles_x = (g.IW + 31) / 32, tiles_y = (g.IH + TILE_Y - 1) / TILE_Y;
    const size_t lds = (size_t)4 * Geo<32>::ALLOC + (size_t)RINGP * TBP;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_deconv128p), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_deconv128p, dim3(xcd_grid_size(tiles_x * tiles_y * n_images)), dim3(256), lds, stream, in, out, w.d_w_mfma16,
                       w.d_bias, g.IW, g.IH, g.OW, g.OH, tiles_x, tiles_x * tiles_y, n_images, in_layout, out_layout,
                       (relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW) | (nt_store_wanted((size_t)g.OH * g.OW * g.COUT * n_images) ? ACT_NT_STORE : 0u));
    return hipGetLastError();
}

// =====================================================================================================================
// conv2d<> 128 -> 128 (layers 1 and 2), the same pipeline.  K walk, parity planes and their rolling refresh as in
// k_mfma16.hip; the 50 passes of a tile are unrolled (tap offsets and refresh slots are compile-time, and the two pixel
// buffers alternate while a 25-pass channel-group window is odd).
// =====================================================================================================================
__host__ __device__ constexpr int refresh_start_p(int plane) { return plane == 3 ? 1 : plane == 0 ? 10 : plane == 1 ? 16 : 22; }
__host__ __device__ constexpr int refresh_plane_p(int t)
{
    for (int pl = 0; pl < 4; pl++)
        if (t >= refresh_start_p(pl) && t < refresh_start_p(pl) + 3) return pl;
    return -1;
}
__host__ __device__ constexpr int refresh_slot_p(int t) { return refresh_plane_p(t) < 0 ? 0 : t - refresh_start_p(refresh_plane_p(t)); }
__host__ __device__ constexpr uint32_t tap_off_p(int t)   // byte offset of tap t (plane-ordered walk) inside the patch
{
    const Tap a = conv_tap(t);
    return (uint32_t)(((a.ky & 1) * 2 + (a.kx & 1)) * Geo<32>::ALLOC + ((a.ky >> 1) * Geo<32>::PX + (a.kx >> 1)) * 32);
}
__host__ __device__ constexpr int conv_requests_p(int P)   // LDS-DMA instructions a wave issues in pass P (steps 2P, 2P+1)
{
    if (P < 0) return 0;
    return 2 + (refresh_plane_p((2 * P) % 25) >= 0) + (refresh_plane_p((2 * P + 1) % 25) >= 0);
}

struct ConvPCtx {
    uint8_t *patch, *ring;
    const int8_t *wstream;
    const uint8_t *in_img;
    int in_img_bytes;
    const uint8_t *lane_pix, *lane_wt;
    uint32_t qstride;
    int lane, w, hi;
};

template <int P, int NPASS>
__device__ __forceinline__ void conv_pass_p(v4i (&acc)[4][8], v4i (&wf)[8], const v4i (&pc)[4], v4i (&pn)[4], const ConvPCtx &c,
                                            const uint32_t (&poff)[4][3])
{
    constexpr int ALLOC = Geo<32>::ALLOC;
    constexpr int tA = (2 * P) % 25, tB = (2 * P + 1) % 25, qA = (2 * P) / 25, qB = (2 * P + 1) / 25;
    // (1) plane refresh pieces scheduled for these two steps, weight tiles PFP steps ahead
    constexpr int rpA = refresh_plane_p(tA), rpB = refresh_plane_p(tB);
    auto dma = [&]() {
        if constexpr (rpA >= 0)
            load_piece<ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpA, refresh_slot_p(tA) * 4 + c.w,
                              poff[rpA][refresh_slot_p(tA)] + (uint32_t)((rpA == 3) ? qA : qA + 1) * c.qstride);
        if constexpr (rpB >= 0)
            load_piece<ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpB, refresh_slot_p(tB) * 4 + c.w,
                              poff[rpB][refresh_slot_p(tB)] + (uint32_t)((rpB == 3) ? qB : qB + 1) * c.qstride);
        load_wtile_p(c.ring, c.wstream, 2 * P + PFP, c.lane, c.w);
        load_wtile_p(c.ring, c.wstream, 2 * P + 1 + PFP, c.lane, c.w);
    };
    // (2) this pass's MFMAs, the next pass's fragments
    constexpr int PN = (P + 1) % NPASS;
    constexpr uint32_t offNA = tap_off_p((2 * PN) % 25), offNB = tap_off_p((2 * PN + 1) % 25);
    const uint8_t *pixn = c.lane_pix + (c.hi ? offNB : offNA);
    const uint8_t *wtn = c.lane_wt + (uint32_t)((c.hi ? ((2 * PN + 1) % RINGP) : ((2 * PN) % RINGP)) * TBP);
    const v4i none[2] = {};
    // (3) inside pass_p: everything but this pass's and the previous pass's requests has landed (= what pass P+2 reads)
    pass_p<conv_requests_p(P) + conv_requests_p(P - 1), 0, false>(acc, wf, pc, pn, pixn, wtn, false, none, dma);
}

template <int P, int NPASS>
__device__ __forceinline__ void conv_passes_p(v4i (&acc)[4][8], v4i (&wf)[8], v4i (&pa)[4], v4i (&pb)[4], const ConvPCtx &c,
                                              const uint32_t (&poff)[4][3])
{
    if constexpr ((P & 1) == 0)
        conv_pass_p<P, NPASS>(acc, wf, pa, pb, c, poff);
    else
        conv_pass_p<P, NPASS>(acc, wf, pb, pa, c, poff);
    if constexpr (P + 1 < NPASS) conv_passes_p<P + 1, NPASS>(acc, wf, pa, pb, c, poff);
}

__global__ __launch_bounds__(256, 2) void k_conv128p(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                     const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias, int IW, int IH,
                                                     int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout, int out_layout,
                                                     uint32_t act_floor)
{
    constexpr int NQ = 4, CIN = 128, COUT = 128, NPASS = 25 * NQ / 2;
    constexpr int PX = Geo<32>::PX, ALLOC = Geo<32>::ALLOC, SLOTS = Geo<32>::SLOTS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem, *ring = smem + 4 * ALLOC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, hi = g >> 1, half = g & 1;
    const int item = xcd_logical_index(n_tiles * n_images);
    if (item < 0) return;
    const int img = item / n_tiles, tile = item - img * n_tiles;
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int Y0 = tile_y * TILE_Y, X0 = tile_x * 32;
    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * out_img_bytes;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);

    uint32_t poff[4][3];
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
#pragma unroll
        for (int slot = 0; slot < SLOTS; slot++) {
            const PieceSrc ps = piece_src<32>(im, slot * 4 + w, lane, Y0 - 1, X0 - 1, 2, pl >> 1, pl & 1, IW, IH);
            poff[pl][slot] = ps.ok ? ps.off : OOB;
        }
    const ConvPCtx ctx{patch, ring, wstream, in_img, in_img_bytes, patch + (uint32_t)(((2 * w) * PX + pos) * 32 + half * 16),
                       ring + (uint32_t)(pos * 32 + half * 16), im.grp, lane, w, hi};
    // ---- prologue: planes 0..2 of group 0 (plane 3 arrives in steps 1..3) + PFP weight tiles ------------------------------
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int slot = 0; slot < SLOTS; slot++) load_piece<ALLOC>(patch, in_img, in_img_bytes, pl, slot * 4 + w, poff[pl][slot]);
#pragma unroll
    for (int s = 0; s < PFP; s++) load_wtile_p(ring, wstream, s, lane, w);
    // accumulators start at the bias: register r of tile (c, j) is channel 64 (j>>2) + 16 g + 4 (j&3) + r
    v4i acc[4][8];
#pragma unroll
    for (int J = 0; J < 2; J++) {
        const v4i b4 = *(const v4i *)(bias + 64 * J + 16 * g);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            v4i v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = (int)(int8_t)((uint32_t)b4[jj] >> (8 * r));
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][4 * J + jj] = v;
        }
    }
    wait_vmcnt<0>();
    block_barrier();
    v4i wf[8], pa[4], pb[4];
    {   // fragments of pass 0: taps 0 / 1 of group 0, ring slots 0 / 1
        const uint8_t *p0 = ctx.lane_pix + (hi ? tap_off_p(1) : tap_off_p(0));
        const uint8_t *w0 = ctx.lane_wt + hi * TBP;
#pragma unroll
        for (int r = 0; r < 8; r++) wf[r] = *(const v4i *)(w0 + r * 16 * 32);
#pragma unroll
        for (int r = 0; r < 4; r++) pa[r] = *(const v4i *)(p0 + ((r / 2) * PX + (r % 2) * 16) * 32);
    }
    // VALU-initialised accumulators -> asm MFMA reading them as C: the wait states hipcc cannot know about
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" : "+v"(acc[c][j]));
    asm volatile("s_nop 3" ::: "memory");
    conv_passes_p<0, NPASS>(acc, wf, pa, pb, ctx, poff);
    wait_vmcnt<0>();
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" : "+v"(acc[c][j]));
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, out_img_bytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int gy = Y0 + 2 * w + c / 2, gx = X0 + (c % 2) * 16 + pos;
        const bool ok = gy < OH && gx < OW;
        const uint32_t off0 = tensor_offset(om, gy, gx, (uint32_t)(g >> 1)) + 16u * (g & 1);
#pragma unroll
        for (int J = 0; J < 2; J++) {
            v4i v;
#pragma unroll
            for (int d = 0; d < 4; d++)
                v[d] = (int)pack4_relu7(acc[c][4 * J + d][0], acc[c][4 * J + d][1], acc[c][4 * J + d][2], acc[c][4 * J + d][3],
                                        act_floor & ACT_FLOOR_MASK);
            if (act_floor & ACT_NT_STORE)
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 2);
            else
                __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 0);
        }
    }
}

hipError_t launch_conv128p(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images,
                           hipStream_t stream, int in_layout, int out_layout, bool relu)
{
    if (g.transposed || g.CIN != 128 || g.COUT != 128) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB || (size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;
    const int tiles_x = (g.OW + 31) / 32, tiles_y = (g.OH + TILE_Y - 1) / TILE_Y;
    const size_t lds = (size_t)4 * Geo<32>::ALLOC + (size_t)RINGP * TBP;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv128p), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv128p, dim3(xcd_grid_size(tiles_x * tiles_y * n_images)), dim3(256), lds, stream, in, out, w.d_w_mfma16,
                       w.d_bias, g.IW, g.IH, g.OW, g.OH, tiles_x, tiles_x * tiles_y, n_images, in_layout, out_layout,
                       (relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW) | (nt_store_wanted((size_t)g.OH * g.OW * g.COUT * n_images) ? ACT_NT_STORE : 0u));
    return hipGetLastError();
}

}  // namespace sicn
