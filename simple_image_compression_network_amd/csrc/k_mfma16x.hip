// conv2d<> / deconv522<> 128 -> 128 (layers 1, 2 / 5, 6: 70 % of the step) — the WIDE PERSISTENT form (round 3).
//
// Where k_mfma16p.hip stands: 2 workgroups x 4 waves per CU, 64 positions x 128 channels per wave, 12 fragment
// reads per 32 MFMAs, every workgroup streams its own copy of the weights; the passes are issue-bound and a bare loop of that
// structure tops out at 3.44 POP/s (tools/microbench/wide_dma.hip, profiles/r03_microbench_wide_dma.txt).  The same
// microbenchmark says what the alternative keeps once it carries a real pass's load: ONE wave per SIMD with 128 positions x 128
// channels per wave (256 accumulators pinned in AGPRs, 16 fragment reads per 64 MFMAs), 4 LDS-DMA requests, a counted wait and a
// workgroup barrier per pass: 3.75 POP/s, +9 %.  Round 2's k_conv128w had that tile but lost it all in its per-tile prologue and
// epilogue, which nothing covers when a SIMD holds a single wave.  Here
//   * one workgroup of 4 waves per CU owns 16 x 32 positions (wave w = rows 4w .. 4w+3) and is PERSISTENT: it walks through its
//     share of the tile list, the plane refresh of a tile's last channel groups fetches the next tile's first ones, the weight
//     stream wraps, and between two tiles sit only the accumulator hand-over (read, ReLU, pack, store, bias) — no prologue;
//   * the tile list is dealt STATICALLY (XCD x works through the contiguous range [x * per, (x + 1) * per) of the list, its 32
//     workgroups interleaved): with one workgroup per CU all tiles take the same time, so there is no scheduler state at all;
//   * one weight ring per CU (not two): half the weight requests per MFMA; a 16 x 32 tile has 10 % less halo than two 8 x 32;
//   * every LDS read is an asm statement with an immediate offset (no address arithmetic in the loop) and hand-placed waits, the
//     MFMAs are asm statements with the accumulator tied to an AGPR: the order written is the order issued;
//   * the ring has 10 slots = 5 passes: a 50-pass tile is 0 mod 5, so every slot and every offset of the unrolled tile is a
//     compile-time constant, and 3 passes' requests may be in flight at a barrier.
// Conv: 4 parity planes of one channel group (18 x 34 positions x 32 B, 20 KiB each) + ring = 120 KB of LDS.
// Hazards hipcc cannot see inside asm are covered by hand and CHECKED: tools/isa_hazards.py runs over this file's ISA in build().
#include <algorithm>

#include "k_common.hpp"

namespace sicn {
namespace xw {

constexpr int TY = 16, TX = 32, PY = TY + 2, PX = TX + 2, PIX = PY * PX;   // 18 x 34 = 612 patch positions per plane / group
constexpr int PIECES = (PIX * KSTEP + 1023) / 1024;                       // 20 LDS-DMA pieces of 1 KiB
constexpr int SLOTS = PIECES / 4;                                         // 5 per wave
static_assert(PIECES % 4 == 0, "every wave issues the same number of pieces");
constexpr int PLANE = PIECES * 1024;                                      // 20480
constexpr int RING = 10, TB = 128 * KSTEP;                                // 10 slots x 4 KiB: one K step of 128 output channels
constexpr int FLIGHT = 3;                                                 // passes whose requests may be in flight at a barrier
constexpr int NSTORE = 16;                                                // output stores per wave and accumulator hand-over

typedef __attribute__((address_space(3))) uint8_t lds_u8;   // LDS pointers stay 32-bit (a generic pointer costs a null check per cast)

#define SICN_MFMA_A(ACC, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))

template <int OFF>
__device__ __forceinline__ v4i lds_read(uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    v4i v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

#ifdef SICN_STAMP   // diagnostic build only (tools/x_stamps.py): per workgroup {cycles, 100 MHz ticks, tiles, hand-over cycles, start, HW_ID, XCC_ID}
__device__ unsigned long long *g_sicn_stamp_x = nullptr;
extern "C" int sicn_debug_stamp_buffer_x(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sicn_stamp_x), &p, sizeof p); }
#endif

// ---- the conv's DMA schedule: one 25-pass window = two channel groups (50 K steps); a tile is two windows ------------------
// Plane pl of the group the window starts with is last READ (fragments are fetched one pass ahead of their MFMAs) in pass
// 3 / 6 / 9 / 11, of the window's second group in pass 15 / 18 / 21 / 23; it is re-filled from the pass after that.  What a
// pass reads was requested at least FLIGHT + 1 passes earlier.  KIND: which group a piece belongs to —
//   CUR0: the window's first group (the tail of plane 2 and plane 3, requested in the window's first passes)
//   CUR1: the window's second group
//   NEXT0: the first group of the NEXT window (in a tile's second window: of the workgroup's next tile)
enum { K_CUR0 = 0, K_CUR1 = 1, K_NEXT0 = 2 };
struct Refresh {
    int plane, slot, kind;
};
__host__ __device__ constexpr Refresh refresh_x(int P, int idx)   // the idx-th (0 / 1) piece of window pass P, plane < 0: none
{
    constexpr Refresh none{-1, 0, 0};
    switch (P) {
    case 0: return idx == 0 ? Refresh{2, 3, K_CUR0} : Refresh{3, 0, K_CUR0};
    case 1: return idx == 0 ? Refresh{2, 4, K_CUR0} : Refresh{3, 1, K_CUR0};
    case 2: return idx == 0 ? Refresh{3, 2, K_CUR0} : none;
    case 3: return idx == 0 ? Refresh{3, 3, K_CUR0} : none;
    case 4: return idx == 0 ? Refresh{3, 4, K_CUR0} : Refresh{0, 0, K_CUR1};
    case 5: return idx == 0 ? Refresh{0, 1, K_CUR1} : none;
    case 6: return idx == 0 ? Refresh{0, 2, K_CUR1} : Refresh{0, 3, K_CUR1};
    case 7: return idx == 0 ? Refresh{0, 4, K_CUR1} : Refresh{1, 0, K_CUR1};
    case 8: return idx == 0 ? Refresh{1, 1, K_CUR1} : none;
    case 9: return idx == 0 ? Refresh{1, 2, K_CUR1} : none;
    case 10: return idx == 0 ? Refresh{1, 3, K_CUR1} : Refresh{2, 0, K_CUR1};
    case 11: return idx == 0 ? Refresh{1, 4, K_CUR1} : Refresh{2, 1, K_CUR1};
    case 12: return idx == 0 ? Refresh{2, 2, K_CUR1} : Refresh{3, 0, K_CUR1};
    case 13: return idx == 0 ? Refresh{2, 3, K_CUR1} : Refresh{3, 1, K_CUR1};
    case 14: return idx == 0 ? Refresh{2, 4, K_CUR1} : Refresh{3, 2, K_CUR1};
    case 15: return idx == 0 ? Refresh{3, 3, K_CUR1} : Refresh{3, 4, K_CUR1};
    case 16: return idx == 0 ? Refresh{0, 0, K_NEXT0} : none;
    case 17: return idx == 0 ? Refresh{0, 1, K_NEXT0} : none;
    case 18: return idx == 0 ? Refresh{0, 2, K_NEXT0} : none;
    case 19: return idx == 0 ? Refresh{0, 3, K_NEXT0} : Refresh{1, 0, K_NEXT0};
    case 20: return idx == 0 ? Refresh{0, 4, K_NEXT0} : Refresh{1, 1, K_NEXT0};
    case 21: return idx == 0 ? Refresh{1, 2, K_NEXT0} : none;
    case 22: return idx == 0 ? Refresh{1, 3, K_NEXT0} : Refresh{2, 0, K_NEXT0};
    case 23: return idx == 0 ? Refresh{1, 4, K_NEXT0} : Refresh{2, 1, K_NEXT0};
    case 24: return idx == 0 ? Refresh{2, 2, K_NEXT0} : none;
    }
    return none;
}
__host__ __device__ constexpr int requests_x(int P)   // LDS-DMA instructions a wave issues in window pass P (any integer P)
{
    const int p = ((P % 25) + 25) % 25;
    return 2 + (refresh_x(p, 0).plane >= 0) + (refresh_x(p, 1).plane >= 0);
}
__host__ __device__ constexpr int in_flight_x(int P)   // requests of the last FLIGHT passes
{
    int s = 0;
    for (int i = 0; i < FLIGHT; i++) s += requests_x(P - i);
    return s;
}
// the schedule against the walk, checked at compile time: a piece of plane pl, kind k, requested in pass P must be requested
// after the plane's last read and FLIGHT + 1 passes before its first read
__host__ __device__ constexpr bool schedule_ok()
{
    int count[3][4] = {};
    for (int P = 0; P < 25; P++)
        for (int i = 0; i < 2; i++) {
            const Refresh r = refresh_x(P, i);
            if (r.plane < 0) continue;
            count[r.kind][r.plane] |= 1 << r.slot;
            // steps (window-local) at which the OLD content of the plane is last used and the NEW content first used
            const int first_tap[4] = {0, 9, 15, 21}, last_tap[4] = {8, 14, 20, 24};
            int last_use_step = 0, first_use_step = 0;   // relative to this window's step 0
            if (r.kind == K_CUR0) { last_use_step = last_tap[r.plane] - 25; first_use_step = first_tap[r.plane]; }
            else if (r.kind == K_CUR1) { last_use_step = last_tap[r.plane]; first_use_step = 25 + first_tap[r.plane]; }
            else { last_use_step = 25 + last_tap[r.plane]; first_use_step = 50 + first_tap[r.plane]; }
            // MFMAs of step s run in pass floor(s / 2) and their fragments are read one pass earlier
            auto fl = [](int s) { return s >= 0 ? s / 2 : -((-s + 1) / 2); };
            const int last_read = fl(last_use_step) - 1, first_read = fl(first_use_step) - 1;
            if (P <= last_read) return false;               // would overwrite fragments still to be read
            if (P > first_read - 1 - FLIGHT) return false;   // would not be covered by the counted wait before its first read
        }
    // every plane's 5 slots exactly once per group: CUR0 carries plane 2 slots 3, 4 and plane 3; NEXT0 the rest
    for (int pl = 0; pl < 4; pl++) {
        if (count[K_CUR1][pl] != 31) return false;
        if ((count[K_CUR0][pl] | count[K_NEXT0][pl]) != 31 || (count[K_CUR0][pl] & count[K_NEXT0][pl]) != 0) return false;
    }
    return true;
}
static_assert(schedule_ok(), "conv refresh schedule violates the read / landing windows");

__host__ __device__ constexpr uint32_t tap_off_x(int t)   // byte offset of tap t (plane-ordered walk) inside the planes
{
    const Tap a = conv_tap(t);
    return (uint32_t)(((a.ky & 1) * 2 + (a.kx & 1)) * PLANE + ((a.ky >> 1) * PX + (a.kx >> 1)) * 32);
}

struct TileX {
    int img, Y0, X0;
};

struct ConvXCtx {
    uint32_t lane_pix;      // LDS address of this lane's pixel fragment 0 at tap offset 0: ((4w) * PX + pos) * 32 + half * 16
    uint32_t lane_wt;       // LDS address of this lane's weight fragment 0 in ring slot 0 (+ one slot for the upper K half)
    lds_u8 *planes_w, *ring_w;       // planes / ring + w * 1024: where this wave's piece of a slot goes (LDS-DMA destination)
    uint32_t lane_w16;               // lane * 16 + w * 1024: this lane's bytes of this wave's piece of a weight tile
    const int8_t *wstream;
    const uint8_t *dma_img;          // the image the plane refresh reads: moves to the next tile's ahead of the tile
    int img_bytes;
    uint32_t grp;                    // byte stride between channel groups of the input
    int lane, w;
    bool hi;
};

// source offsets (channel group 0) of this wave's 5 pieces of each plane for the tile at (Y0, X0): piece k = slot * 4 + w
__device__ __forceinline__ void set_poff_x(uint32_t (&poff)[4][SLOTS], const TensorMap &im, const TileX &t, int w, int lane, int IW, int IH,
                                           bool valid)
{
#pragma unroll
    for (int slot = 0; slot < SLOTS; slot++) {
        const int p = (slot * 4 + w) * 32 + (lane >> 1);
        const int ty = p / PX, tx = p - ty * PX;
#pragma unroll
        for (int pl = 0; pl < 4; pl++) {
            const int iy = 2 * (t.Y0 - 1 + ty) + (pl >> 1), ix = 2 * (t.X0 - 1 + tx) + (pl & 1);
            const bool ok = valid && p < PIX && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
            poff[pl][slot] = ok ? tensor_offset(im, iy, ix, 0u) + (uint32_t)(lane & 1) * 16u : OOB;
        }
    }
}

// One pass (two K steps) of the conv: 64 MFMAs on the current fragments, the 16 reads of the next pass, this pass's requests.
//   T = pass of the tile (0 .. 49); the window pass is T % 25, the window T / 25
template <int T>
__device__ __forceinline__ void conv_pass_x(v4i (&acc)[8][8], const v4i (&pc)[8], const v4i (&wc)[8], v4i (&pn)[8], v4i (&wn)[8],
                                            const ConvXCtx &c, const uint32_t (&poff)[4][SLOTS], bool stores_in_flight)
{
    constexpr int P = T % 25, WIN = T / 25;
    // ---- fragments of the next pass (T + 1, wrapping into the next tile: same offsets, the ring does not care) --------------
    constexpr int TN = (T + 1) % 50;
    constexpr uint32_t offNA = tap_off_x((2 * TN) % 25), offNB = tap_off_x((2 * TN + 1) % 25);
    constexpr int slotN = (2 * TN) % RING;                    // the upper K half reads slot slotN + 1 (lane_wt carries that)
    const uint32_t pixn = c.lane_pix + (c.hi ? offNB : offNA);
    const uint32_t wtn = c.lane_wt;
    // ---- this pass's requests: weight tiles 5 passes ahead, the plane pieces of the schedule -------------------------------
    auto dma = [&](auto idx_tag) {
        constexpr int idx = decltype(idx_tag)::value;   // 0, 1: weight tiles; 2, 3: plane pieces
        if constexpr (idx < 2) {
            constexpr int step = (2 * T + 2 * (RING / 2) + idx) % 100, slot = (2 * T + idx) % RING;   // slot of step s is s % RING, and RING divides 2 * (RING / 2)
            __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)c.wstream, 0, 100 * TB, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, c.ring_w + slot * TB, 16, c.lane_w16, (uint32_t)(step * TB), 0, 0);
        } else {
            constexpr Refresh r = refresh_x(P, idx - 2);
#ifdef SICN_XW_NOPLANE   // timing experiment (wrong results): the passes without their plane requests
            constexpr bool plane_dma = false;
#else
            constexpr bool plane_dma = true;
#endif
            if constexpr (r.plane >= 0 && plane_dma) {
                // group of the piece: window WIN holds groups 2 WIN, 2 WIN + 1; NEXT0 of the second window = group 0 (of the next tile)
                constexpr int q = r.kind == K_CUR0 ? 2 * WIN : r.kind == K_CUR1 ? 2 * WIN + 1 : (2 * WIN + 2) % 4;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)c.dma_img, 0, c.img_bytes, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, c.planes_w + r.plane * PLANE + r.slot * 4096, 16,
                                                         poff[r.plane][r.slot] + (uint32_t)q * c.grp, 0, 0, 0);
            }
        }
    };
    int issued = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = j * 8 + i;
            SICN_MFMA_A(acc[i][j], wc[j], pc[i]);
            // the 16 reads of the next pass: one behind every third MFMA, the last one behind MFMA 47
            if (m % 3 == 2 && issued < 16) {
                switch (issued) {
#define SICN_R(R) case R: pn[R] = lds_read<((R >> 1) * PX + (R & 1) * 16) * 32>(pixn); break;
                    SICN_R(0) SICN_R(1) SICN_R(2) SICN_R(3) SICN_R(4) SICN_R(5) SICN_R(6) SICN_R(7)
#undef SICN_R
#define SICN_R(R) case 8 + R: wn[R] = lds_read<slotN * TB + R * 16 * 32>(wtn); break;
                    SICN_R(0) SICN_R(1) SICN_R(2) SICN_R(3) SICN_R(4) SICN_R(5) SICN_R(6) SICN_R(7)
#undef SICN_R
                }
                issued++;
            }
            if (m == 4) dma(std::integral_constant<int, 0>{});
            if (m == 20) dma(std::integral_constant<int, 1>{});
            if (m == 36) dma(std::integral_constant<int, 2>{});
            if (m == 52) dma(std::integral_constant<int, 3>{});
        }
    }
    // everything but the requests of the last FLIGHT passes has landed; right behind a hand-over the output stores are younger
    // than what the first FLIGHT - 1 waits are after: counted, not waited for
#if defined(SICN_XW_NOWAIT)   // timing experiment (wrong results): never wait for a request
    constexpr int VM = 40;
#elif defined(SICN_XW_NOPLANE)
    constexpr int VM = 2 * FLIGHT;
#else
    constexpr int VM = in_flight_x(P);
#endif
    if (T < FLIGHT - 1 && stores_in_flight)
        wait_vmcnt<VM + NSTORE>();
    else
        wait_vmcnt<VM>();
    block_barrier();   // lgkmcnt(0) (this wave's reads of the next pass are complete) + s_barrier
}

template <int T, int END>
__device__ __forceinline__ void conv_passes_x(v4i (&acc)[8][8], v4i (&pa)[8], v4i (&wa)[8], v4i (&pb)[8], v4i (&wb)[8], const ConvXCtx &c,
                                              const uint32_t (&poff)[4][SLOTS], bool stores_in_flight)
{
    if constexpr ((T & 1) == 0)
        conv_pass_x<T>(acc, pa, wa, pb, wb, c, poff, stores_in_flight);
    else
        conv_pass_x<T>(acc, pb, wb, pa, wa, c, poff, stores_in_flight);
    if constexpr (T + 1 < END) conv_passes_x<T + 1, END>(acc, pa, wa, pb, wb, c, poff, stores_in_flight);
}

constexpr int BIAS_LDS = 4 * PLANE + RING * TB;   // 128 bias bytes behind the ring
constexpr size_t CONV_LDS = BIAS_LDS + 128;
constexpr int SWITCH_T = 25 + 16;                 // from this pass of a tile on, every plane request belongs to the next tile

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_conv_x(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ wstream, const int8_t *__restrict__ bias, int IW,
    int IH, int OW, int OH, int tiles_x, int n_tiles, int n_images, int in_layout, int out_layout, uint32_t act_floor)
{
    constexpr int CIN = 128, COUT = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4, half = g & 1;
    const bool hi = (g >> 1) != 0;
    const int in_img_bytes = IH * IW * CIN, out_img_bytes = OH * OW * COUT;
    const TensorMap im = tensor_map(in_layout, CIN, IW, IH), om = tensor_map(out_layout, COUT, OW, OH);
    // this workgroup's tiles: XCD x = l % 8 owns [x * per, (x + 1) * per) of the (tile x, tile y, image) list; its gridDim.x / 8
    // workgroups take every (gridDim.x / 8)-th one — neighbours in the list run on one XCD at about the same time
    const int total = n_tiles * n_images, per = (total + N_XCD - 1) / N_XCD;
    const int xcd = (int)blockIdx.x % N_XCD, stride = (int)gridDim.x / N_XCD;
    int item = xcd * per + (int)blockIdx.x / N_XCD;
    const int item_end = min(total, (xcd + 1) * per);
    if (item >= item_end) return;   // before any barrier or request
    auto coord = [&](int it) {
        const int img = it / n_tiles, tile = it - img * n_tiles, ty = tile / tiles_x;
        return TileX{img, ty * TY, (tile - ty * tiles_x) * TX};
    };
    TileX tc = coord(item);
    uint32_t poff[4][SLOTS];
    set_poff_x(poff, im, tc, w, lane, IW, IH, true);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    ConvXCtx ctx{lds0 + (uint32_t)(((4 * w) * PX + pos) * 32 + half * 16),
                 lds0 + 4 * PLANE + (uint32_t)(pos * 32 + half * 16) + (hi ? (uint32_t)TB : 0u),
                 (lds_u8 *)LDS_PTR(smem) + w * 1024,
                 (lds_u8 *)LDS_PTR(smem) + 4 * PLANE + w * 1024,
                 (uint32_t)(lane * 16 + w * 1024),
                 wstream,
                 in + (size_t)tc.img * in_img_bytes,
                 in_img_bytes,
                 im.grp,
                 lane,
                 w,
                 hi};
    // ---- prologue of the FIRST tile only: what the previous window's NEXT0 requests would have brought (planes 0, 1 and
    // ---- slots 0 .. 2 of plane 2 of group 0) + the weight tiles of passes 0 .. 4 + the bias ---------------------------------
    {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)ctx.dma_img, 0, in_img_bytes, 0x00020000);
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int slot = 0; slot < (pl < 2 ? SLOTS : 3); slot++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, ctx.planes_w + pl * PLANE + slot * 4096, 16, poff[pl][slot], 0, 0, 0);
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wstream, 0, 100 * TB, 0x00020000);
#pragma unroll
        for (int s = 0; s < RING; s++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, ctx.ring_w + s * TB, 16, ctx.lane_w16, (uint32_t)(s * TB), 0, 0);
        if (tid < COUT / 4) ((uint32_t *)(smem + BIAS_LDS))[tid] = ((const uint32_t *)bias)[tid];
    }
    v4i acc[8][8];
    // accumulators start at the bias: register r of tile (c, j) is channel 64 (j >> 2) + 16 g + 4 (j & 3) + r
    auto init_acc = [&]() {
#pragma unroll
        for (int J = 0; J < 2; J++) {
            const v4i b4 = *(const v4i *)(smem + BIAS_LDS + 64 * J + 16 * g);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                v4i v;
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = (int)(int8_t)((uint32_t)b4[jj] >> (8 * r));
#pragma unroll
                for (int c = 0; c < 8; c++) acc[c][4 * J + jj] = v;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; c++)
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" : "+a"(acc[c][j]));
        asm volatile("s_nop 3" ::: "memory");   // v_accvgpr_write -> asm MFMA reading it as SrcC
    };
    wait_vmcnt<0>();
    __syncthreads();   // requests landed, bias visible
    init_acc();
    v4i pa[8], wa[8], pb[8], wb[8];
    {   // fragments of pass 0: taps 0 / 1 of group 0, ring slots 0 / 1
        const uint32_t p0 = ctx.lane_pix + (hi ? tap_off_x(1) : tap_off_x(0));
#define SICN_R(R) pa[R] = lds_read<((R >> 1) * PX + (R & 1) * 16) * 32>(p0); wa[R] = lds_read<R * 16 * 32>(ctx.lane_wt);
        SICN_R(0) SICN_R(1) SICN_R(2) SICN_R(3) SICN_R(4) SICN_R(5) SICN_R(6) SICN_R(7)
#undef SICN_R
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int tiles_done = __builtin_amdgcn_readfirstlane(0);
#ifdef SICN_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_hand = 0, st_p0 = 0;
#endif
#pragma unroll 1
    for (;;) {
        {   // keep hipcc from hoisting the tile's ~180 LDS-DMA destinations (M0 values) out of the tile loop: it did, and spilled
            // 115 SGPRs into VGPR lanes — a v_readlane + wait states in front of every request
            lds_u8 *pw = ctx.planes_w, *rw_ = ctx.ring_w;
            asm volatile("" : "+s"(pw), "+s"(rw_));
            ctx.planes_w = pw;
            ctx.ring_w = rw_;
        }
        conv_passes_x<0, SWITCH_T>(acc, pa, wa, pb, wb, ctx, poff, tiles_done > 0);
        // every plane request for THIS tile has been issued: the rest of the tile fetches the next tile's first group
        const int next = item + stride;
        const bool has_next = next < item_end;
        const TileX tn = coord(has_next ? next : item);
        {
            int lane_l;   // recomputed, not kept: the loop has no registers to spare
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_l));
            set_poff_x(poff, im, tn, w, lane_l, IW, IH, has_next);
        }
        ctx.dma_img = in + (size_t)tn.img * in_img_bytes;
        conv_passes_x<SWITCH_T, 50>(acc, pa, wa, pb, wb, ctx, poff, false);
#ifdef SICN_STAMP
        st_p0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- accumulator hand-over: read, ReLU, pack, store; then the bias again ---------------------------------------------
        {
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // the last MFMAs' results -> v_accvgpr_read
#pragma unroll
            for (int c = 0; c < 8; c++)
#pragma unroll
                for (int j = 0; j < 8; j++) asm volatile("" : "+a"(acc[c][j]));   // the reads below stay behind the s_nop
            int lane_e;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            const int pos_e = lane_e & 15, g_e = lane_e >> 4;
            uint8_t *out_img = out + (size_t)tc.img * out_img_bytes;
            __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, out_img_bytes, 0x00020000);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int gy = tc.Y0 + 4 * w + (c >> 1), gx = tc.X0 + (c & 1) * 16 + pos_e;
                const bool ok = gy < OH && gx < OW;
                const uint32_t off0 = tensor_offset(om, gy, gx, (uint32_t)(g_e >> 1)) + 16u * (g_e & 1);
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
        if (!has_next) break;
        item = next;
        tc = tn;
        tiles_done = __builtin_amdgcn_readfirstlane(tiles_done + 1);
        init_acc();
#ifdef SICN_STAMP
        st_hand += __builtin_amdgcn_s_memtime() - st_p0;
#endif
    }
#ifdef SICN_STAMP
    if (g_sicn_stamp_x && tid == 0) {
        unsigned long long *o = g_sicn_stamp_x + (size_t)blockIdx.x * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_t0;
        o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
        o[2] = (unsigned long long)(tiles_done + 1);
        o[3] = st_hand;
        o[4] = st_t0;
        o[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
        o[6] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
    }
#endif
    wait_vmcnt<0>();   // the wrapped tail of the prefetch (weights, out-of-range plane pieces) must land before the LDS is released
}

}  // namespace xw

// conv 128 -> 128 on 16 x 32 tiles by one persistent workgroup per CU
bool wide_supported(const LayerGeom &g) { return g.CIN == 128 && g.COUT == 128 && !g.transposed; }

hipError_t launch_wide(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images, hipStream_t stream,
                       int in_layout, int out_layout, bool relu, int grid_cap)
{
    using namespace xw;
    if (!wide_supported(g)) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB || (size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;
    const int MW = g.transposed ? g.IW : g.OW, MH = g.transposed ? g.IH : g.OH;
    const int tiles_x = (MW + TX - 1) / TX, tiles_y = (MH + TY - 1) / TY;
    const long total = (long)tiles_x * tiles_y * n_images;
    if (total <= 0 || total > 0x7fffffffL) return hipErrorInvalidValue;
    const uint32_t flags = (relu ? ACT_FLOOR_RELU : ACT_FLOOR_RAW) |
                           (nt_store_wanted((size_t)g.OH * g.OW * g.COUT * n_images) ? ACT_NT_STORE : 0u);
    const long cap = grid_cap > 0 ? std::max(N_XCD, grid_cap / N_XCD * N_XCD) : 256;              // one resident per CU
    const unsigned grid = (unsigned)std::min<long>(cap, (total + N_XCD - 1) / N_XCD * N_XCD);      // a multiple of 8
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_x), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_LDS);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv_x, dim3(grid), dim3(256), CONV_LDS, stream, in, out, w.d_w_mfma16, w.d_bias, g.IW, g.IH, g.OW, g.OH, tiles_x,
                       tiles_x * tiles_y, n_images, in_layout, out_layout, flags);
    return hipGetLastError();
}

}  // namespace sicn
