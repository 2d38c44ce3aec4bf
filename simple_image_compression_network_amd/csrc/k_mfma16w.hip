// conv2d<> 128 -> 128 channels (layers 1 and 2) on v_mfma_i32_16x16x64_i8 with a 128 x 128 output tile PER WAVE and ONE
// wave per SIMD — the "wide" form of k_mfma16.hip (read that file first: same K walk, LDS patch planes, weight ring,
// layouts and fragment geometry).
//
// Why (tools/microbench/mfma_agpr.hip, profiles/r02_microbench_mfma_agpr.txt): k_mfma16 gives a wave 64 positions x 128
// channels (128 accumulator registers, 2 waves per SIMD) and re-reads 12 fragments from LDS per 32 MFMAs; LDS bandwidth is
// then the co-bottleneck (3.53 POP/s in a bare loop, 3.05 in the kernel).  Here a wave owns 128 positions x 128 channels:
// 256 accumulator registers, which only fit as AGPRs at one wave per SIMD, 16 fragment reads per 64 MFMAs (a third less
// LDS traffic per MAC: 3.77 POP/s in the bare loop).  Two things make that form work where the compiler-scheduled attempt
// of round 1 did not (2.7 POP/s):
//   * the accumulators are PINNED in AGPRs: the MFMA is an inline-asm statement with a tied "+a" operand (hipcc's own
//     allocation moved them through ~7 v_accvgpr copies per MFMA);
//   * with a single wave per SIMD nothing else hides the LDS round trip, so the fragments are double-buffered and the 16
//     reads of pass k+1 are issued in program order between the MFMAs of pass k (one ds_read_b128 per 4 MFMAs; the asm
//     statements are volatile, so the order written here is the order issued).  The weight ring therefore runs one pass
//     further ahead (PFW = 8 K steps) and the barrier that ends pass k publishes the tiles of pass k+2.
// Workgroup = 2 waves (128 threads), 8 x 32 positions as in k_mfma16 (wave w = rows 4w .. 4w+3), two workgroups per CU
// = one wave per SIMD, same 80 KB of LDS per workgroup.  The 50 passes of a tile are fully unrolled (the fragment buffers
// alternate and a 25-pass channel-group window is odd).
// Hazards the compiler cannot see inside the asm statements are covered by hand: s_nop after the accumulator
// initialisation and before the epilogue's v_accvgpr_read; an accumulator tile is touched once per pass (64 MFMAs apart).
#include "k_common.hpp"

namespace sicn {

constexpr int PFW = 8;     // weight tiles (K steps) requested ahead of the MFMAs that use them: 4 passes
constexpr int RINGW = 8;   // ring slots; slot of step s is free again once pass s/2 - 1 has read it
static_assert(RINGW == PFW, "the tile requested in pass k overwrites the slot pass k - 1 finished reading");

#define SICN_MFMA_ACC(ACC, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))

__host__ __device__ constexpr int refresh_start_w(int plane) { return plane == 3 ? 1 : plane == 0 ? 10 : plane == 1 ? 16 : 22; }
__host__ __device__ constexpr int refresh_plane_w(int t)
{
    for (int pl = 0; pl < 4; pl++)
        if (t >= refresh_start_w(pl) && t < refresh_start_w(pl) + 3) return pl;
    return -1;
}
__host__ __device__ constexpr int refresh_slot_w(int t) { return refresh_plane_w(t) < 0 ? 0 : t - refresh_start_w(refresh_plane_w(t)); }

struct PassGeo {   // compile-time description of pass P (steps 2P, 2P+1) of the 100-step walk of a 4-group layer
    int tA, tB, qA, qB;          // tap index inside the channel group, channel group
    uint32_t offA, offB;         // byte offset of the tap inside the patch (plane + position shift)
};
__host__ __device__ constexpr PassGeo pass_geo(int P)
{
    constexpr int PX = Geo<32>::PX, ALLOC = Geo<32>::ALLOC;
    const int sA = 2 * P, sB = 2 * P + 1;
    const Tap a = conv_tap(sA % 25), b = conv_tap(sB % 25);
    PassGeo g{sA % 25, sB % 25, sA / 25, sB / 25, 0u, 0u};
    g.offA = (uint32_t)(((a.ky & 1) * 2 + (a.kx & 1)) * ALLOC + ((a.ky >> 1) * PX + (a.kx >> 1)) * 32);
    g.offB = (uint32_t)(((b.ky & 1) * 2 + (b.kx & 1)) * ALLOC + ((b.ky >> 1) * PX + (b.kx >> 1)) * 32);
    return g;
}
__host__ __device__ constexpr int pass_requests(int P)   // LDS-DMA instructions a wave issues in pass P
{
    if (P < 0) return 0;
    return 4 + 2 * (refresh_plane_w((2 * P) % 25) >= 0) + 2 * (refresh_plane_w((2 * P + 1) % 25) >= 0);
}

struct ConvWCtx {
    uint8_t *patch, *ring;
    const int8_t *wstream;
    const uint8_t *in_img;
    int in_img_bytes;
    uint32_t lane_pix, lane_wt, qstride;
    int lane, w, hi;
};

constexpr int TBW = 128 * KSTEP;   // one weight tile: 128 rows x 32 B

__device__ __forceinline__ void load_wtile_w(const ConvWCtx &c, int tile)
{
    const int8_t *src = c.wstream + (size_t)tile * TBW + c.lane * 16;
    uint8_t *dst = c.ring + (tile % RINGW) * TBW;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int piece = 2 * r + c.w;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src + piece * 1024), LDS_PTR(dst + piece * 1024), 16, 0, 0);
    }
}

// pass P: DMA requests, the 64 MFMAs on the CURRENT fragments (pc, wc), the 16 reads of pass P+1 into (pn, wn), then the
// counted wait + barrier that publishes what pass P+2's reads will need
template <int P, int NPASS>
__device__ __forceinline__ void conv_pass_w(v4i (&acc)[8][8], const v4i (&pc)[8], const v4i (&wc)[8], v4i (&pn)[8], v4i (&wn)[8],
                                            const ConvWCtx &c, const uint32_t (&poff)[4][3][2])
{
    constexpr PassGeo G = pass_geo(P), N = pass_geo((P + 1) % NPASS);
    constexpr int PX = Geo<32>::PX, ALLOC = Geo<32>::ALLOC;
    // (1) plane refresh pieces scheduled for these two steps (2 pieces per wave and step), weight tiles PFW steps ahead
    constexpr int rpA = refresh_plane_w(G.tA), rpB = refresh_plane_w(G.tB);
#ifndef SICN_EXPW_NODMA   // timing experiments only (wrong results)
    if constexpr (rpA >= 0) {
        constexpr int slot = refresh_slot_w(G.tA);
#pragma unroll
        for (int i = 0; i < 2; i++)
            load_piece<ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpA, slot * 4 + 2 * c.w + i,
                              poff[rpA][slot][i] + (uint32_t)((rpA == 3) ? G.qA : G.qA + 1) * c.qstride);
    }
    if constexpr (rpB >= 0) {
        constexpr int slot = refresh_slot_w(G.tB);
#pragma unroll
        for (int i = 0; i < 2; i++)
            load_piece<ALLOC>(c.patch, c.in_img, c.in_img_bytes, rpB, slot * 4 + 2 * c.w + i,
                              poff[rpB][slot][i] + (uint32_t)((rpB == 3) ? G.qB : G.qB + 1) * c.qstride);
    }
    load_wtile_w(c, 2 * P + PFW);
    load_wtile_w(c, 2 * P + 1 + PFW);
#endif
    // (2) MFMAs of this pass, reads of the next one.  Fragment c of a wave: row c / 2 of its 4 rows, column tile c % 2.
    constexpr int stepNA = 2 * ((P + 1) % NPASS), stepNB = stepNA + 1;
    const uint32_t pixn = c.lane_pix + (c.hi ? N.offB : N.offA);
    const uint32_t wtn = c.lane_wt + (uint32_t)((c.hi ? (stepNB % RINGW) : (stepNA % RINGW)) * TBW);
    const uint8_t *pb = c.patch + pixn, *wb = c.ring + wtn;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            SICN_MFMA_ACC(acc[i][j], wc[j], pc[i]);
            if ((i & 3) == 3) {
                constexpr int dummy = 0;
                (void)dummy;
                const int r = j * 2 + (i >> 2);     // 0..15: the 8 pixel fragments first (needed by the first MFMAs), then the weights
                if (r < 8)
                    pn[r] = *(const v4i *)(pb + ((r >> 1) * PX + (r & 1) * 16) * 32);
                else
                    wn[r - 8] = *(const v4i *)(wb + (r - 8) * 16 * 32);
            }
        }
    }
    // (3) everything but the requests of this pass and the previous one has landed: the tiles / plane pieces pass P+2 reads
#ifndef SICN_EXPW_NODMA
    wait_vmcnt<pass_requests(P) + pass_requests(P - 1)>();
#endif
#ifndef SICN_EXPW_NOBAR
    block_barrier();
#endif
}

template <int P, int NPASS>
__device__ __forceinline__ void conv_passes_w(v4i (&acc)[8][8], v4i (&pa)[8], v4i (&wa)[8], v4i (&pb)[8], v4i (&wb)[8],
                                              const ConvWCtx &c, const uint32_t (&poff)[4][3][2])
{
    if constexpr ((P & 1) == 0)
        conv_pass_w<P, NPASS>(acc, pa, wa, pb, wb, c, poff);
    else
        conv_pass_w<P, NPASS>(acc, pb, wb, pa, wa, c, poff);
    if constexpr (P + 1 < NPASS) conv_passes_w<P + 1, NPASS>(acc, pa, wa, pb, wb, c, poff);
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv128w(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias,
    int IW, int IH, int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout, int out_layout, uint32_t act_floor)
{
    constexpr int NQ = 4, CIN = 128, COUT = 128, NPASS = 25 * NQ / 2;   // 50 passes of two K steps
    constexpr int PX = Geo<32>::PX, ALLOC = Geo<32>::ALLOC;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem, *ring = smem + 4 * ALLOC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, hi = g >> 1, half = g & 1;
    const int item = xcd_logical_index(n_tiles * n_images);
    if (item < 0) return;   // before any LDS-DMA is issued
    const int img = item / n_tiles, tile = item - img * n_tiles;
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int Y0 = tile_y * TILE_Y, X0 = tile_x * 32;

    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * out_img_bytes;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);

    // per-lane source offsets (channel group 0) of this wave's 2 pieces of every refresh slot of the 4 parity planes
    uint32_t poff[4][3][2];
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
#pragma unroll
        for (int slot = 0; slot < 3; slot++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const PieceSrc ps = piece_src<32>(im, slot * 4 + 2 * w + i, lane, Y0 - 1, X0 - 1, 2, pl >> 1, pl & 1, IW, IH);
                poff[pl][slot][i] = ps.ok ? ps.off : OOB;
            }
    const ConvWCtx ctx{patch, ring, wstream, in_img, in_img_bytes, (uint32_t)(((4 * w) * PX + pos) * 32 + half * 16),
                       (uint32_t)(pos * 32 + half * 16), im.grp, lane, w, hi};
    // ---- prologue: planes 0..2 of group 0 (plane 3 arrives in steps 1..3) + PFW weight tiles ------------------------------
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int slot = 0; slot < 3; slot++)
#pragma unroll
            for (int i = 0; i < 2; i++) load_piece<ALLOC>(patch, in_img, in_img_bytes, pl, slot * 4 + 2 * w + i, poff[pl][slot][i]);
#pragma unroll
    for (int s = 0; s < PFW; s++) load_wtile_w(ctx, s);

    // accumulators start at the bias: register r of tile (c, j) is channel 64 (j>>2) + 16 g + 4 (j&3) + r
    v4i acc[8][8];
#pragma unroll
    for (int J = 0; J < 2; J++) {
        const v4i b4 = *(const v4i *)(bias + 64 * J + 16 * g);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            v4i v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = (int)(int8_t)((uint32_t)b4[jj] >> (8 * r));
#pragma unroll
            for (int c = 0; c < 8; c++) acc[c][4 * J + jj] = v;
        }
    }
    wait_vmcnt<0>();
    block_barrier();

    // fragments of pass 0
    v4i pa[8], wa[8], pb[8], wb[8];
    {
        constexpr PassGeo G0 = pass_geo(0);
        const uint8_t *p0 = patch + ctx.lane_pix + (hi ? G0.offB : G0.offA);
        const uint8_t *w0 = ring + ctx.lane_wt + (uint32_t)((hi ? 1 : 0) * TBW);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            pa[r] = *(const v4i *)(p0 + ((r >> 1) * PX + (r & 1) * 16) * 32);
            wa[r] = *(const v4i *)(w0 + r * 16 * 32);
        }
    }
    asm volatile("s_nop 7" ::: "memory");   // v_accvgpr_write -> MFMA SrcC
    conv_passes_w<0, NPASS>(acc, pa, wa, pb, wb, ctx, poff);
    wait_vmcnt<0>();   // the padded tail of the prefetch must land before the LDS is handed to the next workgroup
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMAs -> v_accvgpr_read
#pragma unroll
    for (int c = 0; c < 8; c++)
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" : "+a"(acc[c][j]));   // the reads below stay behind the s_nop

    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, out_img_bytes, 0x00020000);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int gy = Y0 + 4 * w + (c >> 1), gx = X0 + (c & 1) * 16 + pos;
        const bool ok = gy < OH && gx < OW;
        const uint32_t off0 = tensor_offset(om, gy, gx, (uint32_t)(g >> 1)) + 16u * (g & 1);
#pragma unroll
        for (int J = 0; J < 2; J++) {
            v4i v;
#pragma unroll
            for (int d = 0; d < 4; d++)
                v[d] = (int)pack4_relu7(acc[c][4 * J + d][0], acc[c][4 * J + d][1], acc[c][4 * J + d][2], acc[c][4 * J + d][3], act_floor & ACT_FLOOR_MASK);
            __builtin_amdgcn_raw_buffer_store_b128(v, ro, ok ? off0 + (uint32_t)(2 * J) * om.grp : OOB, 0, 0);
        }
    }
}

bool conv128w_supported(const LayerGeom &g) { return !g.transposed && g.CIN == 128 && g.COUT == 128; }

hipError_t launch_conv128w(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images,
                           hipStream_t stream, int in_layout, int out_layout, bool relu)
{
    if (!conv128w_supported(g)) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB || (size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;
    const int tiles_x = (g.OW + 31) / 32, tiles_y = (g.OH + TILE_Y - 1) / TILE_Y;
    const size_t lds = (size_t)4 * Geo<32>::ALLOC + (size_t)RINGW * TBW;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv128w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv128w, dim3(xcd_grid_size(tiles_x * tiles_y * n_images)), dim3(128), lds, stream, in, out, w.d_w_mfma16,
                       w.d_bias, g.IW, g.IH, g.OW, g.OH, tiles_x, tiles_x * tiles_y, n_images, in_layout, out_layout,
                       relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW);
    return hipGetLastError();
}

}  // namespace sicn
