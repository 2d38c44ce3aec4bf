// Layer 0 with its GDN / IGDN in ONE kernel (round 4; extension beyond the reference, parity unpinned like the GDN itself:
// include/sicn_gdn.h).  The layer is conv2d<> with IFM_CH = 3 up to and including the bias add (conv_nonsquare_top.cpp:198-280,
// :272), the activation replaces the sign-bit ReLU of :273-275.
//
// Unfused, the hyperprior's analysis writes layer 0's 2.1 GB of pre-activation lanes (k_l0, 0.43 ms on 8 x 4K), and k_gdn reads
// them back and rewrites them in place (1.42 ms, VALU-bound on its integer roots).  Here a wave computes the 16 positions x 128
// channels of a k_gdn item itself, from the RGBX patch k_l0 keeps in LDS, and hands them to the same gdn_item() body in registers:
//
//   * v_mfma_i32_16x16x64_i8 with the WEIGHTS as the A operand and the pixels as B: K = 64 is two kernel rows of 32 bytes
//     (5 taps x RGBX = 20 bytes, zero weights for the rest), three MFMAs per 16 channels x 16 positions;
//   * the weight rows are permuted exactly like k_gdn's gamma rows — tile j, row 4 g + r = channel 64 (j>>2) + 16 g + 4 (j&3) + r —
//     so the four accumulators of tile j in lane (pos, g) are bytes 0..3 of dword j&3 of that lane's 16-byte B-operand chunk
//     J = j>>2 of the GDN product: the low bytes are packed (2 v_perm + 1 v_or per dword) and the tensor never exists in HBM
//     before the activation;
//   * the kernel stays VALU-bound on the roots (k_gdn_body.hpp); the layer's 24 MFMAs per item run beside the GDN's 32 in the
//     matrix pipe, and what the fusion adds to the VALU is the bias / packing / patch expansion: about 10 %.
//
// Work split: 512 threads = 8 waves, tile = 8 output rows x 32 columns as in k_l0, wave w = row w, two items (columns 0..15 and
// 16..31) per tile; two workgroups per CU = four waves per SIMD at <= 128 VGPRs, the occupancy k_gdn runs at.  A workgroup walks a
// vertical run of <= L0G_CHUNK tiles; the raw pixels of the run, the weights, gamma, beta and the bias come into LDS once in its
// prologue (LDS-DMA), so the tile loop has no global load.
#include "k_gdn_body.hpp"
#include "k_l0_common.hpp"
#include "sicn_gdn_internal.h"

namespace sicn {

#ifndef SICN_L0G_CHUNK
#define SICN_L0G_CHUNK 8
#endif
constexpr int L0G_CHUNK = SICN_L0G_CHUNK;                      // tiles per run (8 is the most two workgroups' LDS holds)
constexpr int L0G_RAW_ROWS = 2 * L0_TY * L0G_CHUNK + 3;        // 131 (round 5: runs of 8 tiles instead of 4: 1.31 -> 1.27 ms on 8 x 4K, profiles/r05_l0g_parts.txt)
constexpr int L0G_RAW_BYTES = (L0G_RAW_ROWS * L0_RAW_DW + 63) / 64 * 256;   // whole request instructions (64 lanes x 4 B)
static_assert(L0G_RAW_BYTES >= L0G_RAW_ROWS * L0_RAW_DW * 4 + 12, "the last quad over-reads 3 dwords");
constexpr int L0G_KSTEPS = 3;                                  // kernel rows (0,1), (2,3), (4, -)
constexpr int L0G_WBYTES = L0G_KSTEPS * 8 * 4 * 16 * 16;       // [ks][tile j][g][rho][16 B] = 24 KB
constexpr int L0G_GAMMA = 128 * 128;
// LDS: weights | gamma | beta (u32 x 128) | bias (i32 x 128) | two patches | raw rows.  The zero-weight half of the last K step
// reads one patch row past a patch: from patch 0 into patch 1, from patch 1 into the raw rows — inside the allocation either way.
constexpr int L0G_LDS = L0G_WBYTES + L0G_GAMMA + 512 + 512 + 2 * L0_PATCH + L0G_RAW_BYTES;
static_assert(L0G_RAW_BYTES >= L0_PITCH, "the over-read row lies inside the raw buffer");
static_assert(2 * L0G_LDS <= 160 * 1024, "two workgroups per CU");

size_t l0g_bytes() { return (size_t)L0G_WBYTES; }

// w_okc: [128][75], k = (ky*5+kx)*3 + c  ->  the A-operand image: byte b of (ks, j, g, rho) = K index 16 g + b of step ks
// (kernel row 2 ks + (K>>5), patch byte K & 31 = 4 kx + c) of channel 64 (j>>2) + 16 (rho>>2) + 4 (j&3) + (rho&3)
void pack_l0g(const int8_t *w_okc, int8_t *dst)
{
    for (int ks = 0; ks < L0G_KSTEPS; ks++)
        for (int j = 0; j < 8; j++)
            for (int g = 0; g < 4; g++)
                for (int rho = 0; rho < 16; rho++) {
                    const int ch = 64 * (j >> 2) + 16 * (rho >> 2) + 4 * (j & 3) + (rho & 3);
                    int8_t *o = dst + ((((size_t)ks * 8 + j) * 4 + g) * 16 + rho) * 16;
                    for (int b = 0; b < 16; b++) {
                        const int kk = 16 * g + b, ky = 2 * ks + (kk >> 5), kx = (kk & 31) >> 2, c = kk & 3;
                        o[b] = (ky < 5 && kx < 5 && c < 3) ? w_okc[(size_t)ch * 75 + (ky * 5 + kx) * 3 + c] : (int8_t)0;
                    }
                }
}

template <bool INVERSE>
__global__ __launch_bounds__(512, 2) void k_l0g(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const int8_t *__restrict__ w_l0g,
                                                const int8_t *__restrict__ bias, const int8_t *__restrict__ gamma_img,
                                                const uint32_t *__restrict__ beta, int IW, int IH, int OW, int OH, int tiles_y, int ty_per,
                                                int out_layout, float kc)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *wl = smem;
    uint8_t *gl = wl + L0G_WBYTES;
    uint32_t *bl = (uint32_t *)(gl + L0G_GAMMA);
    int *bias_lds = (int *)(bl + 128);
    uint8_t *patch0 = (uint8_t *)(bias_lds + 128);
    uint8_t *raw = patch0 + 2 * L0_PATCH;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pos = lane & 15, g = lane >> 4;
    const int img = blockIdx.z;
    const int X0 = blockIdx.x * TILE_X;
    const int ty_begin = blockIdx.y * ty_per, ty_end = min(tiles_y, ty_begin + ty_per);   // ty_per <= L0G_CHUNK
    if (ty_begin >= ty_end) return;  // before any LDS-DMA is issued

    const int im_bytes = IH * IW * 3;
    const int img_byte0 = img * im_bytes;                       // launch_l0_gdn guarantees the tensor is < 2 GiB
    const int tensor_bytes4 = ((int)gridDim.z * im_bytes + 3) & ~3;
    uint8_t *out_img = out + (size_t)img * OH * OW * 128;
    const TensorMap om = tensor_map(out_layout, 128, OW, OH);

    // ---- prologue: everything this workgroup will ever read (the raw rows exactly as in k_l0) ----------------------------------
    {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, tensor_bytes4, 0x00020000);
        const int rows = 2 * L0_TY * (ty_end - ty_begin) + 3;
        const int n_instr = (rows * L0_RAW_DW + 63) / 64;
        for (int k = w; k < n_instr; k += 8) {
            const int idx = 64 * k + lane;
            const int r = idx / L0_RAW_DW, c = idx - r * L0_RAW_DW;
            const int iy = 2 * L0_TY * ty_begin - 2 + r;
            const int o = ((img_byte0 + (iy * IW + 2 * X0 - 2) * 3) & ~3) + 4 * c;   // negative only left of the very first pixel
            const bool ok = r < rows && iy >= 0 && iy < IH && o >= 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(raw + k * 256), 4, ok ? (uint32_t)o : OOB, 0, 0, 0);
        }
        // the weight image and the gamma image are adjacent in LDS: one linear copy of 1 KB pieces each
        for (int piece = w; piece < L0G_WBYTES / 1024; piece += 8)
            __builtin_amdgcn_global_load_lds(GLB_PTR(w_l0g + piece * 1024 + lane * 16), LDS_PTR(wl + piece * 1024), 16, 0, 0);
        for (int piece = w; piece < L0G_GAMMA / 1024; piece += 8)
            __builtin_amdgcn_global_load_lds(GLB_PTR(gamma_img + piece * 1024 + lane * 16), LDS_PTR(gl + piece * 1024), 16, 0, 0);
        if (tid < 128) {
            bl[tid] = beta[tid];
            bias_lds[tid] = (int)bias[tid];
        }
    }
    // the builtin (not inline asm): hipcc then KNOWS no LDS-DMA is pending and puts no vmcnt(0) of its own in front of the LDS
    // accesses of the loop (k_l0)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    l0_expand(raw, patch0, img_byte0, tid, ty_begin * L0_TY, X0, IW, IH);
    __syncthreads();

    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out_img, 0, OH * OW * 128, 0x00020000);
    int buf = 0;
    for (int tile_y = ty_begin; tile_y < ty_end; tile_y++, buf ^= 1) {
        const int Y0 = tile_y * L0_TY;
        const uint8_t *patch = patch0 + buf * L0_PATCH;
        if (tile_y + 1 < ty_end) {   // pixels of the next tile -> the patch nobody reads in this iteration
            const uint8_t *rsrc = raw + (tile_y + 1 - ty_begin) * (2 * L0_TY * L0_RAW_DW * 4);
            l0_expand(rsrc, patch0 + (buf ^ 1) * L0_PATCH, img_byte0, tid, Y0 + L0_TY, X0, IW, IH);
        }

        // ---- layer 0 for the wave's row, both items: acc[it][j][r] = channel 64 (j>>2) + 16 g + 4 (j&3) + r of column 16 it + pos
        v4i acc[2][8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const v4i b4 = *(const v4i *)(bias_lds + 64 * (j >> 2) + 16 * g + 4 * (j & 3));
            acc[0][j] = b4;
            acc[1][j] = b4;
        }
#ifndef SICN_EXP_L0G_NOL0   // timing experiment (wrong results): the kernel without layer 0's fragment reads and MFMAs
#pragma unroll
        for (int ks = 0; ks < L0G_KSTEPS; ks++) {
            v4i pf[2];
#pragma unroll
            for (int it = 0; it < 2; it++) {   // B operand: K bytes 16 g .. + 15 = half g&1 of kernel row 2 ks + (g>>1); 8-byte aligned
                const uint8_t *src = patch + (2 * w + 2 * ks + (g >> 1)) * L0_PITCH + 8 * (16 * it + pos) + 16 * (g & 1);
                const uint2 lo = *(const uint2 *)src, hi = *(const uint2 *)(src + 8);
                pf[it] = v4i{(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const v4i wf = *(const v4i *)(wl + (((ks * 8 + j) * 4 + g) * 16 + pos) * 16);
                acc[0][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, pf[0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf, pf[1], acc[1][j], 0, 0, 0);
            }
        }
#endif
        // the pre-activation lane = the low byte of every accumulator (what k_l0 stores with the RAW floor)
        v4i xf[2][2];
#pragma unroll
        for (int it = 0; it < 2; it++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const v4i a = acc[it][j];
                xf[it][j >> 2][j & 3] = (int)(__builtin_amdgcn_perm((uint32_t)a[1], (uint32_t)a[0], 0x0c0c0400u) |
                                              __builtin_amdgcn_perm((uint32_t)a[3], (uint32_t)a[2], 0x04000c0cu));
            }
        const int gy = Y0 + w;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            v4i y[2];
#ifdef SICN_EXP_L0G_NOGDN   // timing experiment (wrong results): the kernel without the activation
            y[0] = xf[it][0];
            y[1] = xf[it][1];
#else
            gdn_item<2, INVERSE>(xf[it], gl, bl, g, pos, kc, y);
#endif
            const int gx = X0 + 16 * it + pos;
            const bool ok = gy < OH && gx < OW;
#pragma unroll
            for (int J = 0; J < 2; J++) {   // chunk k = 4 J + g: channels 64 J + 16 g .. + 15 = half g&1 of the 32-channel group 2 J + (g>>1)
                const uint32_t off = ok ? tensor_offset(om, gy, gx, (uint32_t)(2 * J + (g >> 1))) + 16u * (uint32_t)(g & 1) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(y[J], ro, off, 0, 0);
            }
        }
        block_barrier();  // next patch complete, this patch free — a RAW barrier (lgkmcnt only): __syncthreads() would also wait for vmcnt(0), i.e. for this tile's
                          // stores to complete, once per tile (round 5: the skeleton of this kernel — no MFMAs, no activation — took 0.55 ms for what k_l0 does in 0.43)
    }
}

hipError_t launch_l0_gdn(const LayerGeom &g, const sicn_weights &w, const sicn_gdn &gdn, const uint8_t *in, uint8_t *out, int n_images,
                         hipStream_t stream, int out_layout, const sicn_options &o, const ChipGeom &chip)
{
    if (g.COUT != 128 || gdn.channels != 128 || !w.d_w_l0g || !gdn.d_gamma_mfma) return hipErrorInvalidValue;
    const int tiles_x = (g.OW + TILE_X - 1) / TILE_X, tiles_y = (g.OH + L0_TY - 1) / L0_TY;
    const L0Cut cut = l0_chunks(tiles_x, tiles_y, n_images, L0G_CHUNK, o.strip_chunks, chip);
    if ((size_t)g.IH * g.IW * 3 * (size_t)n_images + 4 >= (size_t)OOB) return hipErrorInvalidValue;
    if ((size_t)g.OH * g.OW * g.COUT >= (size_t)OOB) return hipErrorInvalidValue;   // buffer-descriptor stores
    if (cut.ty_per > L0G_CHUNK) return hipErrorInvalidValue;
    dim3 grid((unsigned)tiles_x, (unsigned)cut.y_chunks, (unsigned)n_images);
    auto go = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, L0G_LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, grid, dim3(512), L0G_LDS, stream, in, out, w.d_w_l0g, w.d_bias, gdn.d_gamma_mfma, gdn.d_beta_mfma, g.IW, g.IH,
                           g.OW, g.OH, tiles_y, cut.ty_per, out_layout, gdn.kc);
        return hipGetLastError();
    };
    return gdn.inverse ? go(k_l0g<true>) : go(k_l0g<false>);
}

}  // namespace sicn
