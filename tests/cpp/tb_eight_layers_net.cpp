// C++ testbench written the way the reference's own test is (conv3_nonsquare_tb.cpp:781-1132,
// test_eight_layers_net + main): build the all-ones input stream, run eight_layers_net through the
// drop-in veneer (include/sicn_hls.hpp -> libsicn.so -> HIP kernels), compute the golden result
// with the testbench's naive model (tb weight unpack + conv_nonsquare + bias/ReLU, restated in the
// ORACLE, oracle/sicn_oracle.c: sicn_or_naive_conv2d / sicn_or_naive_deconv2d), compare every
// output byte, print "Image # n passed the testing." and return 0, else count errors and return 1.
// Also checks conv2d_layer0 and deconv2d_layer4 (the reference's commented-out unit tests,
// tb:115-285 and tb:300-509) against the same golden chain.
//
// At the reference's OWN size (768 x 512, config_nonsquare.h:5-7, all-ones stimulus, tb:781-821) the naive golden model would
// take minutes, so `direct` selects the oracle's OpenMP closed form (sicn_or_layer_direct: the same bytes, held to the naive
// model and to the reference-compiled vectors by tests/test_oracle_golden.py) as the golden chain.
//
// usage: tb_eight_layers_net <param_weights.bin> [width height] [seed] [direct]   (seed > 0: random image)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "sicn_hls.hpp"

extern "C" {
// oracle/sicn_oracle.c (test infrastructure)
typedef sicn_layer_desc sicn_or_layer_desc;
int sicn_or_naive_conv2d(const sicn_or_layer_desc *, const uint64_t *, const int8_t *, const uint8_t *, uint8_t *);
int sicn_or_naive_deconv2d(const sicn_or_layer_desc *, const uint64_t *, const int8_t *, const uint8_t *, uint8_t *);
int sicn_or_layer_direct(const sicn_or_layer_desc *, const uint64_t *, const int8_t *, const uint8_t *, uint8_t *, int threads);
}

using namespace sicn_hls;

#define MAX_IMAGES 1  // tb:65

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s param_weights.bin [w h] [seed]\n", argv[0]); return 2; }
    const int W = argc > 3 ? std::atoi(argv[2]) : 768, H = argc > 3 ? std::atoi(argv[3]) : 512;
    const unsigned seed = argc > 4 ? (unsigned)std::atoi(argv[4]) : 0;
    const bool direct = argc > 5 && std::string(argv[5]) == "direct";
    Context::load_params(argv[1]);
    Context::set_image_size(W, H);
    std::printf("Input image size is %d X %d X 3\n", W, H);

    // initialize the input image (tb:786-816): every channel = 1, or seeded random bytes
    std::vector<uint8_t> image((size_t)H * W * 3);
    std::mt19937 rng(seed);
    for (auto &v : image) v = seed ? (uint8_t)(rng() & 0xFF) : 1;
    stream<3> input_stream("input_stream");
    for (int oy = 0; oy < H; oy++)
        for (int ox = 0; ox < W; ox++) input_stream.write(&image[((size_t)oy * W + ox) * 3]);

    stream<3> output_stream("output_stream");
    std::printf("Hardware computation begin.  \n");
    eight_layers_net(input_stream, output_stream, MAX_IMAGES);
    std::printf("Hardware computation complete.  \n");

    // golden chain, layer by layer, each layer fed by the golden output of the previous (tb:900-1056)
    std::vector<std::vector<uint8_t>> golden(8);
    const uint8_t *cur = image.data();
    for (int l = 0; l < 8; l++) {
        const sicn_layer_desc d = net_desc(l, W, H);
        const LayerParams &p = Context::get().layer(l);
        golden[l].resize((size_t)d.OFM_ROW * d.OFM_COL * d.OFM_CH);
        std::printf("layer%d verification computation begin. \n", l);
        int rc = direct ? sicn_or_layer_direct(&d, p.weights.m_weights.data(), p.bias.data(), cur, golden[l].data(), 16)
                 : d.transposed ? sicn_or_naive_deconv2d(&d, p.weights.m_weights.data(), p.bias.data(), cur, golden[l].data())
                                : sicn_or_naive_conv2d(&d, p.weights.m_weights.data(), p.bias.data(), cur, golden[l].data());
        if (rc) { std::printf("golden model failed rc=%d\n", rc); return 2; }
        std::printf("layer%d verification computation complete. \n", l);
        cur = golden[l].data();
    }
    const sicn_layer_desc d7 = net_desc(7, W, H);
    std::printf("Output image size is %d X %d X %d\n", d7.OFM_ROW, d7.OFM_COL, d7.OFM_CH);

    int err_counter = 0, err_perimage = 0;
    for (unsigned n_image = 0; n_image < MAX_IMAGES; n_image++) {  // tb:1070-1104
        for (int oy = 0; oy < d7.OFM_COL; oy++)
            for (int ox = 0; ox < d7.OFM_ROW; ox++) {
                uint8_t outElem[3];
                output_stream.read(outElem);
                for (int channel = 0; channel < 3; channel++) {
                    const int EXP = (int8_t)golden[7][((size_t)oy * d7.OFM_ROW + ox) * 3 + channel];
                    const int out_chan = (int8_t)outElem[channel];
                    if (EXP != out_chan) {
                        if (err_counter < 10)
                            std::printf("ERROR: Expected[%d][%d][%d]=%d actual %d\n", oy, ox, channel, EXP, out_chan);
                        err_counter++;
                        err_perimage++;
                    }
                }
            }
        if (err_perimage == 0)
            std::printf("Image # %u passed the testing.\n", n_image);
        else {
            err_perimage = 0;
            std::printf("Image # %u failed the testing.\n", n_image);
        }
    }
    if (!output_stream.empty()) { std::printf("ERROR: output stream holds extra words\n"); err_counter++; }

    // conv2d_layer0 (tb:115-285)
    {
        stream<3> in0;
        for (size_t i = 0; i < (size_t)H * W; i++) in0.write(&image[i * 3]);
        stream<128> out0;
        conv2d_layer0(in0, out0, 1);
        std::vector<uint8_t> got(golden[0].size());
        out0.drain(got.data(), got.size() / 128);
        int e = 0;
        for (size_t i = 0; i < got.size(); i++) e += got[i] != golden[0][i];
        std::printf("conv2d_layer0: %s (%d byte errors)\n", e ? "failed" : "passed", e);
        err_counter += e;
    }
    // deconv2d_layer4 (tb:300-509), fed with the golden latent
    {
        const sicn_layer_desc d4 = net_desc(4, W, H);
        stream<192> in4;
        for (size_t i = 0; i < (size_t)d4.IFM_ROW * d4.IFM_COL; i++) in4.write(&golden[3][i * 192]);
        stream<128> out4;
        deconv2d_layer4(in4, out4, 1);
        std::vector<uint8_t> got(golden[4].size());
        out4.drain(got.data(), got.size() / 128);
        int e = 0;
        for (size_t i = 0; i < got.size(); i++) e += got[i] != golden[4][i];
        std::printf("deconv2d_layer4: %s (%d byte errors)\n", e ? "failed" : "passed", e);
        err_counter += e;
    }
    return err_counter == 0 ? 0 : 1;
}
