// ORACLE PIN (test infrastructure, NOT product code): a harness translation unit around the
// reference's own golden convolution.  It #includes /root/reference/conv.hpp UNMODIFIED (-I on the
// command line; nothing of the reference is copied into this repository) and instantiates
//     conv_nonsquare<MAX_IMAGE, IFMDim_x, IFMDim_y, OFMDim_x, OFMDim_y, IFMCh, OFMCh, 5, 5, S, S, TI, TO, TW>
// (conv.hpp:91-123) — the model the reference's own testbench trusts (conv3_nonsquare_tb.cpp:608-612,
// 728-732, 871-874) — with native fixed-width types in place of the Vivado containers:
//     TI = uint8_t  for ap_uint<IN_BIT = 8>   (tb:581 input_padding)
//     TO = int8_t   for ap_int<OUT_BIT = 8>   (tb:536 output; `tmp += a*b` wraps mod 2^8: gcc defines the
//                                              int -> int8_t narrowing as modular, like ap_int<8>)
//     TW = int8_t   for ap_int<W_BIT = 4>     (sign-extended nibbles; the product is formed in int either way)
// conv.hpp has no Vivado types and needs only <iostream> (it prints progress to std::cout, conv.hpp:119-121;
// OFMDim_y >= 8 is required because that line divides by OFMDim_y / 8).
//
// The bias / ReLU tail below restates conv3_nonsquare_tb.cpp:616-627 (`output += BIAS; if (output < 0)
// output = 0` on ap_int<8>).  Padding (tb:581-600) and the zero-stuffed deconv map (tb:700-718) are built by
// the caller, tests/golden/make_ref_conv_vectors.py, which cites those lines.
//
// Array index order is the reference's: img[n][x][y][c], weights[o][kx][ky][c], out[n][x][y][o]  (x FIRST).
//
// Built by oracle/Makefile into oracle/_ref/libsicn_refconv.so (git-ignored); only
// tests/golden/make_ref_conv_vectors.py (run in the build container, where /root/reference exists)
// loads it.  The fixtures it writes travel; this library and the reference do not have to.
#include <cstdint>
#include <iostream>
#include <sstream>

#include "conv.hpp"   // /root/reference/conv.hpp, unmodified

namespace {

template <int IX, int IY, int OX, int OY, int CI, int CO, int S>
void run_case(const uint8_t* img, const int8_t* w, const int8_t* bias, int8_t* out) {
    typedef uint8_t const (*img_t)[IX][IY][CI];
    typedef int8_t const (*w_t)[5][5][CI];
    typedef int8_t (*out_t)[OX][OY][CO];
    conv_nonsquare<1, IX, IY, OX, OY, CI, CO, 5, 5, S, S, uint8_t, int8_t, int8_t>(
        reinterpret_cast<img_t>(img), reinterpret_cast<w_t>(w), reinterpret_cast<out_t>(out));
    if (bias) {   // conv3_nonsquare_tb.cpp:616-627
        out_t o = reinterpret_cast<out_t>(out);
        for (int y = 0; y < OY; y++)
            for (int x = 0; x < OX; x++)
                for (int ch = 0; ch < CO; ch++) {
                    o[0][x][y][ch] += bias[ch];
                    if (o[0][x][y][ch] < 0) o[0][x][y][ch] = 0;
                }
    }
}

struct CaseDims { int id, ix, iy, ox, oy, ci, co, s; };

}  // namespace

// One line per instantiated shape: id, padded input x/y, output x/y, Cin, Cout, stride.
// conv (stride 2): padded = W+4 / H+4, out = ceil(W/2) / ceil(H/2).   deconv (stride 1): padded = 2W+4 / 2H+4, out = 2W / 2H.
#define SICN_REF_CASES(X)                                                                            \
    /* small seeded-random cases: Cin/Cout in {3,128,192}, odd sizes, OFMDim_y >= 8 */               \
    X(0, 28, 20, 12, 8, 3, 128, 2)      /* conv   24x16  3->128 */                                   \
    X(1, 25, 21, 11, 9, 3, 128, 2)      /* conv   21x17  3->128 (odd) */                             \
    X(2, 24, 20, 10, 8, 128, 128, 2)    /* conv   20x16  128->128 */                                 \
    X(3, 23, 21, 10, 9, 128, 192, 2)    /* conv   19x17  128->192 (odd) */                           \
    X(4, 16, 12, 12, 8, 192, 128, 1)    /* deconv 6x4    192->128 */                                 \
    X(5, 14, 14, 10, 10, 128, 128, 1)   /* deconv 5x5    128->128 */                                 \
    X(6, 22, 16, 18, 12, 128, 3, 1)     /* deconv 9x6    128->3 */                                   \
    /* the eight layers of eight_layers_net at 256x256 (BASELINE configs[1] geometry) */             \
    X(10, 260, 260, 128, 128, 3, 128, 2)                                                             \
    X(11, 132, 132, 64, 64, 128, 128, 2)                                                             \
    X(12, 68, 68, 32, 32, 128, 128, 2)                                                               \
    X(13, 36, 36, 16, 16, 128, 192, 2)                                                               \
    X(14, 36, 36, 32, 32, 192, 128, 1)                                                               \
    X(15, 68, 68, 64, 64, 128, 128, 1)                                                               \
    X(16, 132, 132, 128, 128, 128, 128, 1)                                                           \
    X(17, 260, 260, 256, 256, 128, 3, 1)                                                             \
    /* the eight layers at the reference's own 768x512 (config_nonsquare.h) */                       \
    X(20, 772, 516, 384, 256, 3, 128, 2)                                                             \
    X(21, 388, 260, 192, 128, 128, 128, 2)                                                           \
    X(22, 196, 132, 96, 64, 128, 128, 2)                                                             \
    X(23, 100, 68, 48, 32, 128, 192, 2)                                                              \
    X(24, 100, 68, 96, 64, 192, 128, 1)                                                              \
    X(25, 196, 132, 192, 128, 128, 128, 1)                                                           \
    X(26, 388, 260, 384, 256, 128, 128, 1)                                                           \
    X(27, 772, 516, 768, 512, 128, 3, 1)

extern "C" {

// Dimensions of case `id` (so the caller can size and check its buffers). Returns 0, or -1 for an unknown id.
int sicn_refconv_dims(int id, int dims[7]) {
#define X(ID, IX, IY, OX, OY, CI, CO, S) \
    if (id == ID) { int d[7] = {IX, IY, OX, OY, CI, CO, S}; for (int i = 0; i < 7; i++) dims[i] = d[i]; return 0; }
    SICN_REF_CASES(X)
#undef X
    return -1;
}

// img  uint8 [IX][IY][CI]   (already padded / zero-stuffed by the caller)
// w    int8  [CO][5][5][CI] indexed [o][kx][ky][c]
// bias int8  [CO] or NULL (raw wrapped sums)
// out  int8  [OX][OY][CO]
int sicn_refconv_run(int id, const uint8_t* img, const int8_t* w, const int8_t* bias, int8_t* out) {
    std::ostringstream sink;                       // conv.hpp:119-121 prints progress lines
    std::streambuf* old = std::cout.rdbuf(sink.rdbuf());
    int rc = -1;
#define X(ID, IX, IY, OX, OY, CI, CO, S) \
    if (id == ID) { run_case<IX, IY, OX, OY, CI, CO, S>(img, w, bias, out); rc = 0; }
    SICN_REF_CASES(X)
#undef X
    std::cout.rdbuf(old);
    return rc;
}

}  // extern "C"
