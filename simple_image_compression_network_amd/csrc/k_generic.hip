// Shape-agnostic conv2d<> / deconv522<> (any IFM_CH / OFM_CH / image size): one thread per output
// byte, direct evaluation of the closed forms of SURVEY.md §8(a) a8/a9.  It serves layer shapes the
// specialised kernels do not cover (and the force-generic testing switch); it is a HIP kernel like
// the others — there is no CPU fallback anywhere in this library.
//
//   conv   (conv_nonsquare_top.cpp:198-280): out[y][x][o] = relu7((sum in[2y+ky-2][2x+kx-2][c] W + b) mod 256)
//   deconv (conv_nonsquare_top.cpp:71-195) : out[y][x][o] = relu7((sum Up[y+ky-2][x+kx-2][c] W + b) mod 256),
//                                            Up[2i][2j] = in[i][j], zero elsewhere (kernel not flipped)
#include "sicn_internal.h"

namespace sicn {

__global__ __launch_bounds__(256) void k_generic(const uint8_t *__restrict__ in, uint8_t *__restrict__ out,
                                                 const int8_t *__restrict__ w_okc,
                                                 const int8_t *__restrict__ bias, int IW, int IH, int C,
                                                 int OW, int OH, int N, int transposed, int relu)
{
    const size_t total = (size_t)OH * OW * N;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int img = blockIdx.y;
    const int o = (int)(idx % N);
    const size_t pix = idx / N;
    const int x = (int)(pix % OW), y = (int)(pix / OW);
    const uint8_t *im = in + (size_t)img * IH * IW * C;
    const int8_t *wo = w_okc + (size_t)o * 25 * C;
    int acc = bias[o];
    for (int ky = 0; ky < 5; ky++)
        for (int kx = 0; kx < 5; kx++) {
            int iy, ix;
            if (transposed) {
                const int py = y + ky - 2, px = x + kx - 2;
                if ((py & 1) || (px & 1)) continue;
                iy = py >> 1;
                ix = px >> 1;
            } else {
                iy = 2 * y + ky - 2;
                ix = 2 * x + kx - 2;
            }
            if (iy < 0 || iy >= IH || ix < 0 || ix >= IW) continue;
            const uint8_t *s = im + ((size_t)iy * IW + ix) * C;
            const int8_t *wk = wo + (ky * 5 + kx) * C;
            for (int c = 0; c < C; c++) acc += (int)s[c] * (int)wk[c];
        }
    const int v = acc & 0xFF;
    out[(size_t)img * total + idx] = (uint8_t)((relu && (v & 0x80)) ? 0 : v);
}

hipError_t launch_generic(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                          int n_images, hipStream_t stream, bool relu)
{
    const size_t total = (size_t)g.OH * g.OW * g.COUT;
    const size_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_generic, dim3((unsigned)blocks, (unsigned)n_images), dim3(256), 0, stream, in, out,
                       w.d_w_okc, w.d_bias, g.IW, g.IH, g.CIN, g.OW, g.OH, g.COUT, g.transposed, relu ? 1 : 0);
    return hipGetLastError();
}

// ---- crop of a batch of NHWC tensors: dst[n][h][w][c] = src[n][y < h][x < w][c] of [n][hs][ws][c], ONE launch ------------------
// (a deconv522 doubles a size that a conv2d rounded up, so a decoded map can be one row / column larger than the tensor it models:
// the hyperprior's scale map.  torch's strided copy_ did this as one D2D memcpy per image: 16 launches of ~50 us per 8 x 4K step.)
// zero `words` 64-bit words (the tile-deal area of a forward pass: a kernel node rather than a memset node, see sicn_net_forward)
__global__ __launch_bounds__(256) void k_zero_words(unsigned long long *p, int words)
{
    for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < words; i += (int)gridDim.x * 256) p[i] = 0ull;
}
hipError_t launch_zero_words(unsigned long long *p, size_t words, hipStream_t stream)
{
    if (!words) return hipSuccess;
    const unsigned blocks = (unsigned)((words + 255) / 256 < 64 ? (words + 255) / 256 : 64);
    hipLaunchKernelGGL(k_zero_words, dim3(blocks), dim3(256), 0, stream, p, (int)words);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_crop_nhwc(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int n, int hs, int ws,
                                                   int h, int w, int c, int vec)
{
    const size_t row_dst = (size_t)w * c, row_src = (size_t)ws * c;
    if (vec) {   // rows of both tensors are multiples of 16 bytes and both bases are 16-byte aligned
        const size_t q_row = row_dst / 16, total = (size_t)n * h * q_row;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t r = i / q_row, q = i - r * q_row, img = r / h, y = r - img * h;
            reinterpret_cast<uint4 *>(dst)[i] = *reinterpret_cast<const uint4 *>(src + ((img * hs + y) * row_src) + q * 16);
        }
    } else {
        const size_t total = (size_t)n * h * row_dst;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
            const size_t r = i / row_dst, b = i - r * row_dst, img = r / h, y = r - img * h;
            dst[i] = src[(img * hs + y) * row_src + b];
        }
    }
}

hipError_t launch_crop_nhwc(const uint8_t *src, uint8_t *dst, int n, int hs, int ws, int h, int w, int c, hipStream_t stream)
{
    const size_t row_dst = (size_t)w * c, row_src = (size_t)ws * c;
    const int vec = row_dst % 16 == 0 && row_src % 16 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const size_t items = (size_t)n * h * (vec ? row_dst / 16 : row_dst);
    if (items == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((items + 255) / 256, 8192);
    hipLaunchKernelGGL(k_crop_nhwc, dim3(blocks), dim3(256), 0, stream, src, dst, n, hs, ws, h, w, c, vec);
    return hipGetLastError();
}

}  // namespace sicn
