// Layer 7 (128 channels -> RGB, deconv522<>: conv_nonsquare_top.cpp:351-353, :71-195) fed with the PRE-ACTIVATION lanes of the
// layer before it: the GDN / IGDN that layer carries (include/sicn_gdn.h — extension beyond the reference, parity unpinned) is
// applied on the way into the LDS window, so the activated 2.1 GB tensor of the hyperprior's synthesis (8 x 4K) is neither written
// by k_gdn nor read back by k_l7 (round 4, the counterpart of k_l0g.hip).
//
// The arithmetic is k_l7's (k_rgb.hip): per input position the 4 output phases x 3 channels are 12 dot products over the 3 x 3
// neighbourhood, v_mfma_i32_16x16x64_i8 with the 16-row virtual weight matrix W' as A (the same image, sicn_weights.d_w_l7) and 16
// positions as B, read from an LDS window of [position][4 chunks x 16 B] x two 64-channel regions with k_l7's chunk swizzle.
// What differs is who fills the window and how the work is cut:
//   * a window position is not copied by LDS-DMA but COMPUTED: a wave loads the 128 raw bytes of 16 positions in the B-operand
//     layout of k_gdn (lane (pos, g): channels 64 J + 16 g .. + 15), runs gdn_item() (k_gdn_body.hpp) and writes the 2 x 16 bytes
//     per lane into region J, chunk g — the layout the consumer's MFMAs read, no transpose;
//   * that is ~3000 VALU cycles per item against ~300 matrix-pipe cycles per consumed tile, so the kernel is bound where k_gdn is
//     bound and the geometry follows the loader: strips of 62 columns + 1 halo column each side = 64 positions per window row, 4
//     rows per step = 16 items of 16 positions = exactly one per wave of a 16-wave workgroup (1024 threads, one per CU, four
//     waves per SIMD at <= 128 VGPRs — k_gdn's occupancy); every input position is activated 64 / 62 times over.  The same step
//     has 4 rows x 4 tiles of 16 positions to consume: one per wave;
//   * the window is k_l7's rolling ring: a step reads 6 rows while the 4 rows of the next step are written into the slots the
//     previous step freed — 10 rows x 64 positions x 128 B = 80 KB, one barrier per step, loader and consumer never touch the
//     same row between two barriers.  W' (18 KB) sits in LDS here (the 72 VGPRs k_l7 keeps it in are the gdn body's), beside
//     gamma and beta.  The raw bytes of an item are requested one step before it is worked on.
// MEASURED (r04, rocprof, 8 x 4K hyperprior step): 2.17 ms against 1.62 ms (k_gdn on the 2.1 GB tensor) + 0.49 ms (k_l7) = 2.11 ms for
// the two kernels it replaces — NO gain, so sicn_options.gdn_fuse has to ask for it (= 2); DESIGN.md section 11.  The activation
// is bound by VALU issue and k_l7 by its fragment reads from LDS and their address arithmetic, not by the HBM traffic the fusion
// removes, and in one kernel the two simply add (the consumer also reads W' from LDS here: 36 instead of 18 KB per tile).  Neither
// two workgroups per CU instead of one (so it is not the wait at the step barrier) nor separate consumer and loader phases (so
// it is not the two interfering in LDS) changes that: 2.17 / 2.17 / 2.23 ms, see L7G below.  Two s_waitcnt vmcnt(0) that hipcc
// had placed right behind the step's store (see request() and the loop) were worth 0.1 ms.
// Out-of-image positions load 0 and activate to 0 (x = 0 -> y = 0 for GDN and IGDN alike): the deconv's zero padding.
#include "k_gdn_body.hpp"
#include "sicn_gdn_internal.h"

namespace sicn {

// Geometry G: PITCH window positions per row (PITCH - 2 columns of outputs), WAVES waves per workgroup, ROWS input rows per step =
// ROWS * PITCH / 16 items of 16 positions and as many consumer tiles, IPW of each per wave.
//   SERIAL = false: the loader writes the rows of step t + 1 while step t is consumed (k_l7's rolling ring, 2 ROWS + 2 rows, one
//                   barrier per step);
//   SERIAL = true:  consumer phase, barrier, loader phase over the rows just freed, barrier: ROWS + 2 rows are enough, so a step can
//                   be four times as many rows in the same LDS.
// Built and measured on the 8 x 4K tensor: <64, 16, 4, false> (62-column strips, one 1024-thread workgroup per CU) and
// <32, 8, 4, false> (30 columns, two 512-thread workgroups per CU) 2.17 ms both; <32, 16, 16, true> (two items and two tiles per
// wave and phase) 2.23 ms.  Only the first is instantiated.
template <int PITCH_, int WAVES_, int ROWS_, bool SERIAL_>
struct L7G {
    static constexpr int PITCH = PITCH_, WAVES = WAVES_, ROWS = ROWS_, COLS = PITCH - 2, TPR = PITCH / 16;
    static constexpr bool SERIAL = SERIAL_;
    static constexpr int BLOCK_PIECES = ROWS * TPR;                   // items per step = tiles per step
    static constexpr int IPW = BLOCK_PIECES / WAVES;                  // of each per wave
    static constexpr int PROLOGUE_PIECES = (ROWS + 2) * TPR;          // the rows the first step reads
    static constexpr int RING_PIECES = ((SERIAL ? ROWS : 2 * ROWS) + 2) * TPR;
    static constexpr int RING_POS = RING_PIECES * 16, REGION = RING_PIECES * 1024;
    static constexpr int WBYTES = 18 * 16 * 64, GAMMA = 128 * 128, STAGE = 192;   // stage: 2 output rows x 32 pixels x 3 B per tile
    static constexpr int LDS = 2 * REGION + WBYTES + GAMMA + 512 + WAVES * IPW * STAGE;
    static constexpr int WGS_PER_CU = (160 * 1024) / LDS >= 2 ? 2 : 1;
    static_assert(IPW * WAVES == BLOCK_PIECES && IPW >= 1 && (TPR % IPW == 0 || IPW % TPR == 0), "whole items and tiles per wave");
    static_assert(LDS * WGS_PER_CU <= 160 * 1024 && WAVES * WGS_PER_CU <= 16, "LDS and <= 128 VGPRs");
};

template <bool INVERSE, class G>
__global__ __launch_bounds__(G::WAVES * 64, G::WGS_PER_CU) void k_l7g(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                                        const int8_t *__restrict__ w_l7, const int8_t *__restrict__ bias,
                                                                        const int8_t *__restrict__ gamma_img, const uint32_t *__restrict__ beta,
                                                                        int IW, int IH, int OW, int OH, int steps_y, int y_chunks,
                                                                        int tiles_x, int in_layout, float kc)
{
    constexpr int L7G_COLS = G::COLS, L7G_PITCH = G::PITCH, L7G_ROWS = G::ROWS, L7G_WAVES = G::WAVES, L7G_BLOCK_PIECES = G::BLOCK_PIECES,
                  L7G_PROLOGUE_PIECES = G::PROLOGUE_PIECES, L7G_RING_PIECES = G::RING_PIECES, L7G_RING_POS = G::RING_POS,
                  L7G_REGION = G::REGION, L7G_WBYTES = G::WBYTES, L7G_GAMMA = G::GAMMA, L7G_STAGE = G::STAGE, TPR = G::TPR, IPW = G::IPW;
    constexpr int CIN = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem;
    uint8_t *wl = patch + 2 * L7G_REGION;
    uint8_t *gl = wl + L7G_WBYTES;
    uint32_t *bl = (uint32_t *)(gl + L7G_GAMMA);
    uint8_t *stage = (uint8_t *)(bl + 128);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kg = lane >> 4;   // MFMA roles: A row / B column, K bytes 16 kg .. + 15; GDN item: position, chunk
    const int item = (int)blockIdx.x;          // strip (fastest), chunk, image
    const int bx = item % tiles_x, by = (item / tiles_x) % y_chunks, img = item / (tiles_x * y_chunks);
    const int X0 = bx * L7G_COLS;
    const int per = (steps_y + y_chunks - 1) / y_chunks;
    const int s_begin = by * per, s_end = min(steps_y, s_begin + per);
    if (s_begin >= s_end) return;  // before any LDS-DMA is issued

    // ---- prologue: W', gamma, beta -> LDS -------------------------------------------------------------------------------------
    for (int piece = w; piece < L7G_WBYTES / 1024; piece += L7G_WAVES)
        __builtin_amdgcn_global_load_lds(GLB_PTR(w_l7 + piece * 1024 + lane * 16), LDS_PTR(wl + piece * 1024), 16, 0, 0);
    for (int piece = w; piece < L7G_GAMMA / 1024; piece += L7G_WAVES)
        __builtin_amdgcn_global_load_lds(GLB_PTR(gamma_img + piece * 1024 + lane * 16), LDS_PTR(gl + piece * 1024), 16, 0, 0);
    if (tid < 128) bl[tid] = beta[tid];
    const int b0 = bias[0], b1 = bias[1], b2 = bias[2];
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), the builtin: hipcc then knows no LDS-DMA is pending (k_l0)

    const int in_img_bytes = IH * IW * CIN;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * OH * OW * 3;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in_img, 0, in_img_bytes, 0x00020000);
    const TensorMap tm = tensor_map(in_layout, CIN, IW, IH);
    // window row r <-> input row iy_top + r; window position q = PITCH r + tx <-> column X0 - 1 + tx; ring slot q mod RING_POS.
    // Piece p = positions 16 p .. + 15: the prologue computes the rows 0 .. ROWS + 1 the first step reads, step t adds block t = the
    // ROWS rows step t + 1 reads beyond the two it shares with step t (<64, 16, 4>: pieces 24 + 16 t .. + 15 = rows 6 + 4 t .. 9 + 4 t).
    const int iy_top = L7G_ROWS * s_begin - 1;
    const int iy_max = min(IH, L7G_ROWS * s_end + 1);   // one past the last row this chunk reads
    auto request = [&](int piece, v4i (&x)[2]) {
        const int q = piece * 16 + m, iy = iy_top + q / L7G_PITCH, ix = X0 - 1 + q % L7G_PITCH;
        // branch-free (& not &&, the offset computed for every lane): as a branch hipcc put the address arithmetic of the taken side
        // on the registers of the loads still in flight — an s_waitcnt vmcnt(0) at the top of every step, i.e. a wait for the
        // previous step's store
        const bool ok = (iy >= 0) & (iy < iy_max) & (ix >= 0) & (ix < IW);
        const uint32_t o0 = tensor_offset(tm, iy, ix, (uint32_t)(kg >> 1)) + 16u * (uint32_t)(kg & 1);
#pragma unroll
        for (int J = 0; J < 2; J++) {   // chunk k = 4 J + kg: channels 64 J + 16 kg .. + 15 = half kg&1 of the 32-channel group 2 J + (kg>>1)
            const uint32_t off = ok ? o0 + (uint32_t)(2 * J) * tm.grp : OOB;
            x[J] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
    };
    auto activate = [&](int piece, const v4i (&xf)[2]) {
        v4i y[2];
        gdn_item<2, INVERSE>(xf, gl, bl, kg, m, kc, y);
        const uint32_t P = (uint32_t)(piece % L7G_RING_PIECES) * 16u + (uint32_t)m;
#pragma unroll
        for (int J = 0; J < 2; J++)   // region J, chunk kg, stored at kg ^ ((P>>2)&3) (k_l7's swizzle)
            *(v4i *)(patch + J * L7G_REGION + P * 64u + ((((uint32_t)kg) ^ ((P >> 2) & 3u)) << 4)) = y[J];
    };
    const int n_steps = s_end - s_begin;
    v4i xn[2];
    block_barrier();                       // gamma and beta are there
    for (int piece = w; piece < L7G_PROLOGUE_PIECES; piece += L7G_WAVES) {   // once per chunk: not pipelined
        request(piece, xn);
        activate(piece, xn);
    }
    request(L7G_PROLOGUE_PIECES + w, xn);   // block 0's first item, worked on in step 0
    block_barrier();
    asm volatile("" : "+v"(xn[0]), "+v"(xn[1]));   // (as in the loop: no load of the prologue is pending in hipcc's books past this point)

    const int py = kg >> 1, px = kg & 1;
    const int cols_here = min(L7G_COLS, IW - X0);
    const bool fast_rows = ((OW * 3) & 3) == 0 && (cols_here & 1) == 0;   // whole dwords per output row of every tile
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, OH * OW * 3, 0x00020000);
    int base = 0;                          // (ROWS PITCH t) mod ring: ring slot of window row ROWS t, column 0
    // ---- loader: this wave's items of block t (the rows step t + 1 adds); the first one's bytes were requested a step ago -----------
    // Unconditional, also in the last step of a chunk (rows past iy_max load nothing and activate to 0 in slots nobody reads any
    // more): with conditions around it hipcc lost count of what is in flight and opened every step with s_waitcnt vmcnt(0) — a
    // wait for the store issued a moment earlier, a fifth of the step.
    auto loader = [&](int t) {
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            const v4i xf[2] = {xn[0], xn[1]};
            const int piece = L7G_PROLOGUE_PIECES + L7G_BLOCK_PIECES * t + w + L7G_WAVES * i;
            request(i + 1 < IPW ? piece + L7G_WAVES : L7G_PROLOGUE_PIECES + L7G_BLOCK_PIECES * (t + 1) + w, xn);
            activate(piece, xf);
            // the next item's bytes are taken out of hipcc's load bookkeeping HERE, an item's time after their request and not behind
            // a store: left to itself it rotates the registers at the bottom of the loop, behind the step's store, with a vmcnt(0)
            asm volatile("" : "+v"(xn[0]), "+v"(xn[1]));
        }
    };
    for (int t = 0; t < n_steps; t++) {
        if constexpr (!G::SERIAL) loader(t);
        // ---- consumer: tiles IPW w .. IPW w + IPW - 1 of the step (tile tau: row tau / TPR, columns 16 (tau % TPR) .. + 15) -----------
        v4i acc[IPW];
#pragma unroll
        for (int i = 0; i < IPW; i++) acc[i] = v4i{b0, b1, b2, 0};      // C row 4 kg + r = phase kg, channel r
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const v4i wf0 = *(const v4i *)(wl + ((tap * 2 + 0) * 16 + m) * 64 + kg * 16);
            const v4i wf1 = *(const v4i *)(wl + ((tap * 2 + 1) * 16 + m) * 64 + kg * 16);
#pragma unroll
            for (int i = 0; i < IPW; i++) {
                const int tau = w * IPW + i, wr = tau / TPR, tc = tau % TPR;
                uint32_t slot = (uint32_t)(base + (wr + tap / 3) * L7G_PITCH + 16 * tc + m + tap % 3);
                slot = min(slot, slot - (uint32_t)L7G_RING_POS);   // one wrap at most
                const uint32_t addr = (uint32_t)((kg & 1) * L7G_REGION) + slot * 64u + ((((uint32_t)(kg >> 1)) ^ ((slot >> 2) & 3u)) << 4);
                const v4i p0 = *(const v4i *)(patch + addr);
                const v4i p1 = *(const v4i *)(patch + (addr ^ 32u));   // channels + 32: chunk ^ 2
                acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf0, p0, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf1, p1, acc[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            const int tau = w * IPW + i, wr = tau / TPR, tc = tau % TPR;
            const int gy = L7G_ROWS * (s_begin + t) + wr;
            const int valid = max(0, min(16, cols_here - 16 * tc));            // columns of this tile inside the strip
            const int row_dw = 3 * valid / 2;                                   // 6 valid bytes / 4
            uint8_t *my_stage = stage + (w * IPW + i) * L7G_STAGE;
            const uint32_t v = pack4_relu7(acc[i][0], acc[i][1], acc[i][2], 0);
            if (fast_rows) {
                // stage [2 rows = py][32 pixels = 2 m + px][3] and write the rows as dwords
                if (m < valid) {
                    uint8_t *d = my_stage + py * 96 + (2 * m + px) * 3;
                    d[0] = (uint8_t)v;
                    d[1] = (uint8_t)(v >> 8);
                    d[2] = (uint8_t)(v >> 16);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private staging: no barrier needed
                const int row = lane >= row_dw ? 1 : 0, col = lane - row_dw * row;
                const bool ok = lane < 2 * row_dw && gy < IH;
                const uint32_t sv = *(const uint32_t *)(my_stage + (ok ? row * 96 + col * 4 : 0));
                const uint32_t off = ok ? (uint32_t)(((2 * gy + row) * OW + 2 * (X0 + 16 * tc)) * 3 + col * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(sv, ro, off, 0, 0);
            } else {
                const int gx = X0 + 16 * tc + m;
                if (gy < IH && m < valid) {
                    uint8_t *dst = out_img + ((size_t)(2 * gy + py) * OW + 2 * gx + px) * 3;
                    dst[0] = (uint8_t)v;
                    dst[1] = (uint8_t)(v >> 8);
                    dst[2] = (uint8_t)(v >> 16);
                }
            }
        }
        block_barrier();   // !SERIAL: block t is complete for every wave, the rows of step t are free; SERIAL: the latter
        if constexpr (G::SERIAL) {
            loader(t);     // over the ROWS rows just freed
            block_barrier();
        }
        base += L7G_ROWS * L7G_PITCH;
        base = base >= L7G_RING_POS ? base - L7G_RING_POS : base;
    }
}

hipError_t launch_l7_gdn(const LayerGeom &g, const sicn_weights &w, const sicn_gdn &gdn, const uint8_t *in, uint8_t *out, int n_images,
                         hipStream_t stream, int in_layout, const sicn_options &o, const ChipGeom &chip)
{
    if (g.CIN != 128 || g.COUT != 3 || gdn.channels != 128 || !w.d_w_l7 || !gdn.d_gamma_mfma) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB) return hipErrorInvalidValue;
    if ((size_t)g.OH * g.OW * 3 >= (size_t)OOB) return hipErrorInvalidValue;   // buffer-descriptor stores
    auto go = [&](auto kernel, auto geom) -> hipError_t {
        using G = decltype(geom);
        const int tiles_x = (g.IW + G::COLS - 1) / G::COLS, steps_y = (g.IH + G::ROWS - 1) / G::ROWS;
        const int y_chunks = l7g_chunks(tiles_x, n_images, steps_y, o.strip_chunks, G::WGS_PER_CU, chip);
        const long wgs = (long)tiles_x * y_chunks * n_images;
        if (wgs > 0x7fffffffL) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)wgs), dim3(G::WAVES * 64), G::LDS, stream, in, out, w.d_w_l7, w.d_bias, gdn.d_gamma_mfma,
                           gdn.d_beta_mfma, g.IW, g.IH, g.OW, g.OH, steps_y, y_chunks, tiles_x, in_layout, gdn.kc);
        return hipGetLastError();
    };
    using Wide = L7G<64, 16, 4, false>;
    static_assert(Wide::COLS == L7G_PLAN_COLS && Wide::ROWS == L7G_PLAN_ROWS && Wide::WGS_PER_CU == 1, "sicn_debug_plan mirrors this geometry");
    return gdn.inverse ? go(k_l7g<true, Wide>, Wide{}) : go(k_l7g<false, Wide>, Wide{});
}

}  // namespace sicn
