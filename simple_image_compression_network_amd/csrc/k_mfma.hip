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
//                    taps (inner); the patch of one channel group (4 planes) sits in LDS.
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
#include "k_common.hpp"

namespace sicn {

// ---- device helpers -------------------------------------------------------------------------
// (plain function templates on purpose: lambdas inside a __global__ template make hipcc's
//  host-side instantiation of the kernel fail with a silent substitution failure, ROCm 7.2)
template <int ROUNDS, int PIECES>
__device__ __forceinline__ void load_patch(uint8_t *patch, const uint8_t *in_img, int in_img_bytes,
                                           const uint32_t (&poff)[ROUNDS], int q, int w)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in_img, 0, in_img_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
        const int piece = r * 4 + w;
        if (piece < PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(patch + piece * 1024), 16,
                                                     poff[r] + (uint32_t)(q * 32), 0, 0, 0);
    }
}

template <int TB>
__device__ __forceinline__ void load_wtile(uint8_t *ring, const int8_t *wstream, int step, int lane, int w)
{
    constexpr int NPB = TB / 1024, WR = (NPB + 3) / 4;
    const int8_t *src = wstream + (size_t)step * TB + lane * 16;
    uint8_t *dst = ring + (step % RING) * TB;
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

template <int NTJ>
__device__ __forceinline__ void mma_step(v16i (&acc)[2][NTJ], const uint8_t *sub_patch, const uint8_t *wt,
                                         const uint32_t (&wrow)[NTJ], int p_lane, int kh, int oy, int ox)
{
    v4i wf[NTJ], pf[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int p = p_lane + (i + oy) * PATCH_X + ox;
        pf[i] = *(const v4i *)(sub_patch + p * 32 + ((((p >> 3) & 1) ^ kh) << 4));
    }
#pragma unroll
    for (int j = 0; j < NTJ; j++) wf[j] = *(const v4i *)(wt + wrow[j]);
#pragma unroll
    for (int j = 0; j < NTJ; j++)
#pragma unroll
        for (int i = 0; i < 2; i++)
            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
}

// one K step: top up the weight ring, consume tile `step`, publish tile step+1
template <int NTJ, int STEPS>
__device__ __forceinline__ void k_step(v16i (&acc)[2][NTJ], const uint8_t *patch, uint8_t *ring,
                                       const int8_t *wstream, const uint32_t (&wrow)[NTJ], int p_lane, int kh,
                                       int lane, int w, int step, int sub, int oy, int ox)
{
    constexpr int TB = NTJ * 32 * KSTEP, WR = (TB / 1024 + 3) / 4;
    if (step + PF < STEPS) load_wtile<TB>(ring, wstream, step + PF, lane, w);
    mma_step<NTJ>(acc, patch + sub * SUB_BYTES, ring + (step % RING) * TB, wrow, p_lane, kh, oy, ox);
    if (step + PF < STEPS)
        wait_vmcnt<(PF - 1) * WR>();
    else
        wait_vmcnt<0>();
    block_barrier();
}

// bias is already in the accumulator: truncate mod 256, relu7, 16 consecutive channels per lane
template <int NTJ>
__device__ __forceinline__ void store_tiles(const v16i (&acc)[2][NTJ], uint8_t *out_img, int OW, int MW, int MH,
                                            int Y0, int X0, int w, int m, int kh, bool deconv, int py, int px)
{
    constexpr int COUT = NTJ * 32;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int gy = Y0 + 2 * w + i, gx = X0 + m;
        if (gy < MH && gx < MW) {
            const int oy_ = deconv ? 2 * gy + py : gy, ox_ = deconv ? 2 * gx + px : gx;
            uint8_t *dst = out_img + ((size_t)oy_ * OW + ox_) * COUT + 16 * kh;
#pragma unroll
            for (int j = 0; j < NTJ; j++) {
                const v16i a = acc[i][j];
                uint4 v;
                v.x = pack4_relu7(a[0], a[1], a[2], a[3]);
                v.y = pack4_relu7(a[4], a[5], a[6], a[7]);
                v.z = pack4_relu7(a[8], a[9], a[10], a[11]);
                v.w = pack4_relu7(a[12], a[13], a[14], a[15]);
                *(uint4 *)(dst + j * 32) = v;
            }
        }
    }
}

template <int NQ, int NTJ, bool DECONV, int MINW>
__global__ __launch_bounds__(256, MINW) void k_mfma_t(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ wstream,
    const int8_t *__restrict__ bias, int IW, int IH, int OW, int OH, int MW, int MH, int tiles_x)
{
    constexpr int CIN = NQ * 32, COUT = NTJ * 32;
    constexpr int NSUB = DECONV ? NQ : 4;
    using PG = PatchGeom<NSUB>;
    constexpr int TB = COUT * KSTEP;  // weight tile bytes
    constexpr int STEPS = 25 * NQ;

    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem;
    uint8_t *ring = smem + PG::ALLOC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int img = blockIdx.z;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int Y0 = tile_y * TILE_Y, X0 = tile_x * TILE_X;

    // ---- per-lane source offsets of the patch pieces (bytes from the image base) --------------
    const int in_img_bytes = IH * IW * CIN;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
    uint8_t *out_img = out + (size_t)img * OH * OW * COUT;
    uint32_t poff[PG::ROUNDS];
#pragma unroll
    for (int r = 0; r < PG::ROUNDS; r++) {
        const int piece = r * 4 + w;
        const int gp = piece * 32 + (lane >> 1);
        const int sub = gp / PATCH_PIX, p = gp - sub * PATCH_PIX;
        const int ty = p / PATCH_X, tx = p - ty * PATCH_X;
        const int hlog = (lane & 1) ^ ((p >> 3) & 1);
        int iy, ix;
        if (DECONV) {
            iy = Y0 - 1 + ty;
            ix = X0 - 1 + tx;
        } else {
            iy = 2 * (Y0 - 1 + ty) + (sub >> 1);
            ix = 2 * (X0 - 1 + tx) + (sub & 1);
        }
        const bool ok = sub < NSUB && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
        poff[r] = ok ? (uint32_t)((iy * IW + ix) * CIN + (DECONV ? sub * 32 : 0) + hlog * 16) : OOB;
    }

    // ---- per-lane LDS read offsets -----------------------------------------------------------
    // weight rows: LDS row = j*32 + m, logical K half kh at physical half kh ^ ((row>>3)&1)
    uint32_t wrow[NTJ];
#pragma unroll
    for (int j = 0; j < NTJ; j++) wrow[j] = (uint32_t)((j * 32 + m) * 32 + ((kh ^ ((m >> 3) & 1)) << 4));
    const int p_lane = (2 * w) * PATCH_X + m;  // + (i + oy) * PATCH_X + ox per tile / tap

    v16i acc[2][NTJ];

    // ---- prologue ----------------------------------------------------------------------------
    load_patch<PG::ROUNDS, PG::PIECES>(patch, in_img, in_img_bytes, poff, 0, w);
#pragma unroll
    for (int s = 0; s < PF; s++) load_wtile<TB>(ring, wstream, s, lane, w);
    wait_vmcnt<0>();
    block_barrier();

    if (DECONV) {
        int step = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ph++) {
            const int py = ph >> 1, px = ph & 1;
            const int nkx = 3 - px, ntap = (3 - py) * nkx;
            init_acc<NTJ>(acc, bias, kh);
#pragma unroll 1
            for (int t = 0; t < ntap; t++) {
                const int iy = t / nkx, ix = t - iy * nkx;
#pragma unroll
                for (int q = 0; q < NQ; q++)
                    k_step<NTJ, STEPS>(acc, patch, ring, wstream, wrow, p_lane, kh, lane, w, step + q, q, iy + py,
                                       ix + px);
                step += NQ;
            }
            store_tiles<NTJ>(acc, out_img, OW, MW, MH, Y0, X0, w, m, kh, true, py, px);
        }
    } else {
        init_acc<NTJ>(acc, bias, kh);
#pragma unroll 1
        for (int q = 0; q < NQ; q++) {
            if (q > 0) {
                // every wave is past the barrier of the last step of group q-1: the patch is free
                load_patch<PG::ROUNDS, PG::PIECES>(patch, in_img, in_img_bytes, poff, q, w);
                wait_vmcnt<0>();
                block_barrier();
            }
#pragma unroll 1
            for (int ky = 0; ky < 5; ky++)
#pragma unroll
                for (int kx = 0; kx < 5; kx++)
                    k_step<NTJ, STEPS>(acc, patch, ring, wstream, wrow, p_lane, kh, lane, w, q * 25 + ky * 5 + kx,
                                       (ky & 1) * 2 + (kx & 1), ky >> 1, kx >> 1);
        }
        store_tiles<NTJ>(acc, out_img, OW, MW, MH, Y0, X0, w, m, kh, false, 0, 0);
    }
}


// Explicit instantiations: the host stubs of a __global__ template that is only named inside
// another template are not emitted by hipcc (ROCm 7.2) otherwise.
#define SICN_INST(NQ, NTJ, D)                                                                             \
    template __global__ void k_mfma_t<NQ, NTJ, D, (NTJ <= 4 ? 2 : 1)>(const uint8_t *__restrict__, uint8_t *__restrict__,       \
                                                const int8_t *__restrict__, const int8_t *__restrict__,   \
                                                int, int, int, int, int, int, int);
SICN_INST(4, 4, true)
SICN_INST(6, 4, true)
SICN_INST(4, 6, true)
SICN_INST(4, 4, false)
SICN_INST(6, 4, false)
SICN_INST(4, 6, false)
#undef SICN_INST

template <int NQ, int NTJ, bool DECONV>
static hipError_t launch_one(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                             int n_images, hipStream_t stream)
{
    constexpr int NSUB = DECONV ? NQ : 4;
    const int MW = DECONV ? g.IW : g.OW, MH = DECONV ? g.IH : g.OH;
    const int tiles_x = (MW + TILE_X - 1) / TILE_X, tiles_y = (MH + TILE_Y - 1) / TILE_Y;
    const size_t lds = PatchGeom<NSUB>::ALLOC + (size_t)RING * NTJ * 32 * KSTEP;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mfma_t<NQ, NTJ, DECONV, (NTJ <= 4 ? 2 : 1)>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)(tiles_x * tiles_y), 1, (unsigned)n_images);
    hipLaunchKernelGGL((k_mfma_t<NQ, NTJ, DECONV, (NTJ <= 4 ? 2 : 1)>), grid, dim3(256), lds, stream, in, out, w.d_w_mfma, w.d_bias,
                       g.IW, g.IH, g.OW, g.OH, MW, MH, tiles_x);
    return hipGetLastError();
}

bool mfma_supported(int cin, int cout)
{
    return (cin == 128 && (cout == 128 || cout == 192)) || (cin == 192 && cout == 128);
}

hipError_t launch_mfma(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                       int n_images, hipStream_t stream)
{
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB) return hipErrorInvalidValue;
    if (g.transposed) {
        if (g.CIN == 128 && g.COUT == 128) return launch_one<4, 4, true>(g, w, in, out, n_images, stream);
        if (g.CIN == 192 && g.COUT == 128) return launch_one<6, 4, true>(g, w, in, out, n_images, stream);
        if (g.CIN == 128 && g.COUT == 192) return launch_one<4, 6, true>(g, w, in, out, n_images, stream);
    } else {
        if (g.CIN == 128 && g.COUT == 128) return launch_one<4, 4, false>(g, w, in, out, n_images, stream);
        if (g.CIN == 192 && g.COUT == 128) return launch_one<6, 4, false>(g, w, in, out, n_images, stream);
        if (g.CIN == 128 && g.COUT == 192) return launch_one<4, 6, false>(g, w, in, out, n_images, stream);
    }
    return hipErrorInvalidValue;
}

// ---- host-side weight packing --------------------------------------------------------------
int mfma_stream_steps(int cin) { return 25 * (cin / 32); }
size_t mfma_stream_bytes(int cin, int cout) { return (size_t)mfma_stream_steps(cin) * cout * KSTEP; }

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
            for (int tap = 0; tap < 25; tap++) pack_tile(w_okc, cin, cout, tap, q, dst + (step++) * tb);
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
}

}  // namespace sicn
