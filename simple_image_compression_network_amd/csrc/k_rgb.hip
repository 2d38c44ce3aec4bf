// The two RGB-facing layers of the net — HBM-bound streaming kernels that still use int8 MFMA so
// the arithmetic stays far below the memory time:
//
//   k_l0 : conv2d<> with IFM_CH = 3 (layer 0, conv_nonsquare_top.cpp:282-286 / 299-301):
//          reads 3 B/pixel, writes OFM_CH B per output pixel (the bound: SURVEY.md §8d).
//   k_l7 : deconv522<> with OFM_CH = 3 (layer 7, conv_nonsquare_top.cpp:351-353):
//          reads IFM_CH B/pixel (the bound), writes 4 x 3 B per input position.
//
// Both follow the closed forms of SURVEY.md §8(a) a8/a9 (bit-exact, wrap mod 256, relu7).
#include "k_common.hpp"

namespace sicn {

// =============================================================================================
// Layer 0.  K = 75 is re-laid as 5 rows (ky) x 32 bytes: the 3-byte pixels are expanded to RGBX
// dwords in LDS, so the 5 taps of one kernel row are 20 contiguous bytes starting at an 8-byte
// aligned address (8*x - 8); the K step is padded to 8 pixel slots = 32 bytes with ZERO WEIGHTS for
// slots 5..7 and for the X byte (whatever data sits there is multiplied by 0).
// Same work split as k_mfma: 8 x 32 output tile, wave w = rows 2w, 2w+1, all output channels.
// =============================================================================================
constexpr int L0_ROWS = 2 * TILE_Y + 3;       // 19 input rows per tile
constexpr int L0_COLS = 2 * TILE_X + 6;       // 70 pixel slots per row (lane 31, kh=1 reads 62+4..69)
constexpr int L0_PITCH = L0_COLS * 4;         // 280 bytes, 8-byte aligned rows

template <int NTJ>
__global__ __launch_bounds__(256, 2) void k_l0(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                               const int8_t *__restrict__ w_l0,
                                               const int8_t *__restrict__ bias, int IW, int IH, int OW,
                                               int OH, int tiles_y, int y_chunks, int out_grouped)
{
    constexpr int COUT = NTJ * 32;
    constexpr int TB = COUT * KSTEP;
    constexpr int WBYTES = 5 * TB;
    constexpr int PATCH_BYTES = ((L0_ROWS * L0_PITCH + 15) / 16) * 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *wl = smem;
    uint32_t *patch = (uint32_t *)(smem + WBYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int img = blockIdx.z;
    const int X0 = blockIdx.x * TILE_X;
    const int ty_per = (tiles_y + y_chunks - 1) / y_chunks;
    const int ty_begin = blockIdx.y * ty_per, ty_end = min(tiles_y, ty_begin + ty_per);

    if (ty_begin >= ty_end) return;  // before any LDS-DMA is issued
    // weights: 5 tiles of [COUT][32 B] (row permutation + half swizzle as in k_mfma), linear copy
    for (int piece = w; piece < WBYTES / 1024; piece += 4)
        __builtin_amdgcn_global_load_lds(GLB_PTR(w_l0 + piece * 1024 + lane * 16), LDS_PTR(wl + piece * 1024),
                                         16, 0, 0);

    uint32_t wrow[NTJ];
#pragma unroll
    for (int j = 0; j < NTJ; j++) wrow[j] = (uint32_t)((j * 32 + m) * 32 + ((kh ^ ((m >> 3) & 1)) << 4));

    const uint8_t *im = in + (size_t)img * IH * IW * 3;
    for (int tile_y = ty_begin; tile_y < ty_end; tile_y++) {
        const int Y0 = tile_y * TILE_Y;
        // ---- fill the RGBX patch: slot (r, c) = input pixel (2*Y0 - 2 + r, 2*X0 - 2 + c) ----
        __syncthreads();  // previous tile's readers are done (also drains the weight DMA once)
        for (int s = tid; s < L0_ROWS * L0_COLS; s += 256) {
            const int r = s / L0_COLS, c = s - r * L0_COLS;
            const int iy = 2 * Y0 - 2 + r, ix = 2 * X0 - 2 + c;
            uint32_t v = 0;
            if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) {
                const uint8_t *p = im + ((size_t)iy * IW + ix) * 3;
                v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
            }
            patch[s] = v;
        }
        __syncthreads();

        v16i acc[2][NTJ];
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
#pragma unroll
        for (int ky = 0; ky < 5; ky++) {
            v4i wf[NTJ], pf[2];
#pragma unroll
            for (int j = 0; j < NTJ; j++) wf[j] = *(const v4i *)(wl + ky * TB + wrow[j]);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint8_t *src = (const uint8_t *)patch + (2 * (2 * w + i) + ky) * L0_PITCH + 8 * m + 16 * kh;
                const uint2 lo = *(const uint2 *)src, hi = *(const uint2 *)(src + 8);
                pf[i] = v4i{(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
            }
#pragma unroll
            for (int j = 0; j < NTJ; j++)
#pragma unroll
                for (int i = 0; i < 2; i++)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], pf[i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int gy = Y0 + 2 * w + i, gx = X0 + m;
            if (gy < OH && gx < OW) {
                uint8_t *out_img = out + (size_t)img * OH * OW * COUT;
                const size_t pix = (size_t)gy * OW + gx;
#pragma unroll
                for (int j = 0; j < NTJ; j++) {
                    const v16i a = acc[i][j];
                    uint4 v;
                    v.x = pack4_relu7(a[0], a[1], a[2], a[3]);
                    v.y = pack4_relu7(a[4], a[5], a[6], a[7]);
                    v.z = pack4_relu7(a[8], a[9], a[10], a[11]);
                    v.w = pack4_relu7(a[12], a[13], a[14], a[15]);
                    uint8_t *dst = out_grouped ? out_img + ((size_t)j * OW * OH + pix) * 32 + 16 * kh
                                               : out_img + pix * COUT + j * 32 + 16 * kh;
                    *(uint4 *)dst = v;
                }
            }
        }
    }
}

size_t l0_bytes(int cout) { return (size_t)5 * cout * KSTEP; }

void pack_l0(const int8_t *w_okc, int cout, int8_t *dst)
{
    // w_okc: [cout][75], k = (ky*5+kx)*3 + c
    for (int ky = 0; ky < 5; ky++)
        for (int row = 0; row < cout; row++) {
            const int j = row >> 5, rho = row & 31;
            const int ch = j * 32 + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
            int8_t logical[32];
            for (int s = 0; s < 8; s++)
                for (int c = 0; c < 4; c++)
                    logical[s * 4 + c] = (s < 5 && c < 3) ? w_okc[(size_t)ch * 75 + (ky * 5 + s) * 3 + c] : 0;
            const int g = (row >> 3) & 1;
            int8_t *t = dst + ((size_t)ky * cout + row) * 32;
            for (int h = 0; h < 2; h++)
                for (int b = 0; b < 16; b++) t[((h ^ g) << 4) + b] = logical[h * 16 + b];
        }
}

hipError_t launch_l0(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                     int n_images, hipStream_t stream, int out_grouped)
{
    const int tiles_x = (g.OW + TILE_X - 1) / TILE_X, tiles_y = (g.OH + TILE_Y - 1) / TILE_Y;
    int y_chunks = (4096 + tiles_x * n_images - 1) / (tiles_x * n_images);
    y_chunks = y_chunks < 1 ? 1 : (y_chunks > tiles_y ? tiles_y : y_chunks);
    dim3 grid((unsigned)tiles_x, (unsigned)y_chunks, (unsigned)n_images);
    const int patch_bytes = ((L0_ROWS * L0_PITCH + 15) / 16) * 16;
    if (g.COUT == 128) {
        const size_t lds = 5 * 128 * KSTEP + patch_bytes;
        hipLaunchKernelGGL(k_l0<4>, grid, dim3(256), lds, stream, in, out, w.d_w_l0, w.d_bias, g.IW, g.IH,
                           g.OW, g.OH, tiles_y, y_chunks, out_grouped);
    } else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// =============================================================================================
// Layer 7.  Per input position (y,x) the 4 output phases x 3 channels are 12 dot products over the
// 3x3 input neighbourhood: row (4*phase + c) of a 16-row virtual weight matrix W' holds
// W[c][2dy-py][2dx-px][:] at neighbourhood tap (dy,dx) (zero where that tap does not belong to the
// phase).  v_mfma_i32_16x16x64_i8: A = W' (16 rows x 64 K bytes), B = 16 positions, 18 K steps
// (9 taps x two 64-channel halves).  Output tile: lane (position l&15, phase l>>4) holds the 3
// channel bytes of out[2y+py][2x+px].
// The input patch uses the sub-patch format of k_mfma's deconv path (4 groups of 32 channels).
// =============================================================================================
__global__ __launch_bounds__(256, 2) void k_l7(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                               const int8_t *__restrict__ w_l7,
                                               const int8_t *__restrict__ bias, int IW, int IH, int OW,
                                               int OH, int tiles_x, int in_grouped)
{
    constexpr int NQ = 4, CIN = 128;
    constexpr int WBYTES = 18 * 16 * 64;  // 18 K steps x 16 rows x 64 bytes
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *patch = smem;
    uint8_t *wl = smem + NQ * SUB_ALLOC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.z;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int Y0 = tile_y * TILE_Y, X0 = tile_x * TILE_X;

    const int in_img_bytes = IH * IW * CIN;
    const uint8_t *in_img = in + (size_t)img * in_img_bytes;
#pragma unroll
    for (int sub = 0; sub < NQ; sub++)
#pragma unroll
        for (int slot = 0; slot < 3; slot++)
            load_piece(patch, in_img, in_img_bytes, sub, slot * 4 + w,
                       piece_src_offset(slot * 4 + w, lane, Y0 - 1, X0 - 1, 1, 0, 0, IW, IH, in_grouped != 0,
                                        (uint32_t)sub, CIN));
    for (int piece = w; piece < WBYTES / 1024; piece += 4)
        __builtin_amdgcn_global_load_lds(GLB_PTR(w_l7 + piece * 1024 + lane * 16), LDS_PTR(wl + piece * 1024),
                                         16, 0, 0);
    wait_vmcnt<0>();
    block_barrier();

    // lane roles in v_mfma_i32_16x16x64_i8: A row / B column = lane & 15, K bytes 16*(lane>>4)..+15
    const int m = lane & 15, kg = lane >> 4;
    // wave w owns rows 2w, 2w+1; each row = two 16-position column tiles
    v4i acc[2][2];
    {
        // C rows 4*kg + r  ->  phase kg, channel r (r == 3 is a dummy row)
        const int b0 = bias[0], b1 = bias[1], b2 = bias[2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int c = 0; c < 2; c++) acc[i][c] = v4i{b0, b1, b2, 0};
    }
#pragma unroll 1
    for (int tap = 0; tap < 9; tap++) {
        const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const v4i wf = *(const v4i *)(wl + ((tap * 2 + half) * 16 + m) * 64 + kg * 16);
            const int sub = 2 * half + (kg >> 1);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int p = (2 * w + i + dy) * PATCH_X + 16 * c + m + dx;
                    const v4i pf = *(const v4i *)(patch + sub * SUB_ALLOC + p * 32 + ((((p >> 3) & 1) ^ (kg & 1)) << 4));
                    acc[i][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, pf, acc[i][c], 0, 0, 0);
                }
        }
    }
    const int py = kg >> 1, px = kg & 1;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int gy = Y0 + 2 * w + i, gx = X0 + 16 * c + m;
            if (gy < IH && gx < IW) {
                const uint32_t v = pack4_relu7(acc[i][c][0], acc[i][c][1], acc[i][c][2], 0);
                uint8_t *dst = out + (((size_t)img * OH + 2 * gy + py) * OW + 2 * gx + px) * 3;
                dst[0] = (uint8_t)v;
                dst[1] = (uint8_t)(v >> 8);
                dst[2] = (uint8_t)(v >> 16);
            }
        }
}

size_t l7_bytes(int cin) { return (size_t)18 * 16 * 64 * (cin / 128); }

void pack_l7(const int8_t *w_okc, int cin, int8_t *dst)
{
    // w_okc: [3][25*cin]; dst: [tap 9][half 2][row 16][64 B], row = 4*phase + c
    for (int tap = 0; tap < 9; tap++)
        for (int half = 0; half < 2; half++)
            for (int row = 0; row < 16; row++) {
                const int ph = row >> 2, c = row & 3, py = ph >> 1, px = ph & 1;
                const int dy = tap / 3, dx = tap % 3, ky = 2 * dy - py, kx = 2 * dx - px;
                int8_t *t = dst + (((size_t)tap * 2 + half) * 16 + row) * 64;
                for (int b = 0; b < 64; b++)
                    t[b] = (c < 3 && ky >= 0 && ky < 5 && kx >= 0 && kx < 5)
                               ? w_okc[(size_t)c * 25 * cin + (ky * 5 + kx) * cin + half * 64 + b]
                               : 0;
            }
}

hipError_t launch_l7(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                     int n_images, hipStream_t stream, int in_grouped)
{
    if (g.CIN != 128 || g.COUT != 3) return hipErrorInvalidValue;
    if ((size_t)g.IH * g.IW * g.CIN >= (size_t)OOB) return hipErrorInvalidValue;
    const int tiles_x = (g.IW + TILE_X - 1) / TILE_X, tiles_y = (g.IH + TILE_Y - 1) / TILE_Y;
    const size_t lds = 4 * SUB_ALLOC + 18 * 16 * 64;
    hipError_t e = hipFuncSetAttribute((const void *)k_l7, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_l7, dim3((unsigned)(tiles_x * tiles_y), 1, (unsigned)n_images), dim3(256), lds, stream,
                       in, out, w.d_w_l7, w.d_bias, g.IW, g.IH, g.OW, g.OH, tiles_x, in_grouped);
    return hipGetLastError();
}

}  // namespace sicn
