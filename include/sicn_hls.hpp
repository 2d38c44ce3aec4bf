// sicn_hls.hpp — header-only C++ veneer that re-creates the reference's entry points
// (conv_nonsquare_top.cpp) on top of the C ABI of sicn.h, for a host program written like the
// reference's testbench (conv3_nonsquare_tb.cpp):
//
//   reference                                                      here (namespace sicn_hls)
//   hls::stream<ap_uint<C*8>>                                       stream<C>   (FIFO of C-byte words)
//   FixedPointWeights<SIMD, ap_int<4>, PE, TILES>  weights.hpp:110  FixedPointWeights (run-time fold)
//   namespace PARAM { weights_layerN, bias_layerN } memdata_nonsquare.h   ParamSet::load(file)
//   conv2d<...>(weights, bias, in, out, numReps)    top:198-280     conv2d(desc, weights, bias, in, out, numReps)
//   deconv522<...>(weights, bias, in, out, numReps) top:71-195      deconv522(desc, ...)
//   conv2d_layer0(in, out, numReps)                 top:282-286     conv2d_layer0(in, out, numReps)
//   deconv2d_layer4(in, out, numReps)               top:288-291     deconv2d_layer4(in, out, numReps)
//   eight_layers_net(in, out, numReps)              top:295-357     eight_layers_net(in, out, numReps)
//
// Same argument order and meaning.  The callee drains `in` (H*W words per image) and appends
// H'*W' words per image to `out`, like the HLS dataflow does; streams live in HOST memory, so
// these wrappers copy through PCIe — they exist for drop-in testbenches, not for throughput (the
// throughput path keeps tensors on the device: sicn.h).  Errors throw std::runtime_error (the
// reference's CASSERT_DATAFLOW would exit(-1), bnn-library.h:55).
//
// The image size is a template constant in the reference (config_nonsquare.h); here it is
// `Context::set_image_size(width, height)`, default 768 x 512 as in config_nonsquare.h:5-7.
#ifndef SICN_HLS_HPP
#define SICN_HLS_HPP

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "sicn.h"

namespace sicn_hls {

inline void check(int rc, const char *what)
{
    if (rc != SICN_OK) throw std::runtime_error(std::string(what) + ": " + sicn_strerror(rc));
}
inline void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// hls::stream<ap_uint<BYTES*8>>: channel c of a word is byte c (bits [8c, 8c+8), tb:807-808).
template <int BYTES>
class stream {
    std::deque<uint8_t> q_;

public:
    explicit stream(const char * = "") {}
    void write(const uint8_t *word) { q_.insert(q_.end(), word, word + BYTES); }
    void read(uint8_t *word)
    {
        if (q_.size() < (size_t)BYTES) throw std::runtime_error("read from empty stream");
        for (int i = 0; i < BYTES; i++) { word[i] = q_.front(); q_.pop_front(); }
    }
    bool empty() const { return q_.empty(); }
    size_t size() const { return q_.size() / BYTES; }  // words
    // bulk forms used by the wrappers
    void drain(uint8_t *dst, size_t words)
    {
        if (size() < words) throw std::runtime_error("stream under-run");
        for (size_t i = 0; i < words * BYTES; i++) { dst[i] = q_.front(); q_.pop_front(); }
    }
    void append(const uint8_t *src, size_t words) { q_.insert(q_.end(), src, src + words * BYTES); }
};

struct FixedPointWeights {  // weights.hpp:110-150, m_weights[PE][TILES] as 64-bit words
    int SIMD = 0, PE = 0, TILES = 0;
    std::vector<uint64_t> m_weights;
};

struct LayerParams {
    FixedPointWeights weights;
    std::vector<int8_t> bias;  // FixedPointWeights<1, ap_int<8>, 1, OFM_CH>
};

struct ParamSet {  // namespace PARAM of memdata_nonsquare.h, read from param_weights.bin (tools/npz_to_bin.py)
    std::vector<LayerParams> layers;
    static ParamSet load(const std::string &path)
    {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        auto rd = [&](void *p, size_t n) {
            if (std::fread(p, 1, n, f) != n) { std::fclose(f); throw std::runtime_error("short read: " + path); }
        };
        char magic[8];
        rd(magic, 8);
        if (std::memcmp(magic, "SICNPAR1", 8)) { std::fclose(f); throw std::runtime_error("bad magic: " + path); }
        uint32_t n = 0;
        rd(&n, 4);
        ParamSet ps;
        for (uint32_t l = 0; l < n; l++) {
            uint32_t h[4];
            rd(h, 16);
            LayerParams lp;
            lp.weights.SIMD = (int)h[0];
            lp.weights.PE = (int)h[1];
            lp.weights.TILES = (int)h[2];
            lp.weights.m_weights.resize((size_t)h[1] * h[2]);
            rd(lp.weights.m_weights.data(), lp.weights.m_weights.size() * 8);
            lp.bias.resize(h[3]);
            rd(lp.bias.data(), h[3]);
            char pad[8];
            rd(pad, (8 - h[3] % 8) % 8);
            ps.layers.push_back(std::move(lp));
        }
        std::fclose(f);
        return ps;
    }
};

inline sicn_layer_desc make_desc(int cin, int cout, int simd, int pe, int width, int height, int transposed)
{
    sicn_layer_desc d{};
    d.K = 5; d.S = 2; d.P = 2;
    d.IFM_CH = cin; d.IFM_ROW = width; d.IFM_COL = height;
    d.OFM_CH = cout;
    d.OFM_ROW = transposed ? 2 * width : (width + 1) / 2;
    d.OFM_COL = transposed ? 2 * height : (height + 1) / 2;
    d.SIMD = simd; d.PE = pe; d.IN_BIT = 8; d.OUT_BIT = 8; d.W_BIT = 4;
    d.W_TILES = (cout / pe) * (25 * cin / simd);
    d.transposed = transposed;
    return d;
}

// One layer through the device: drain numReps images from `in`, run, append to `out`.
template <int CIN, int COUT>
void run_layer(const sicn_layer_desc &d, const FixedPointWeights &w, const std::vector<int8_t> &bias,
               stream<CIN> &in, stream<COUT> &out, unsigned numReps)
{
    if (d.IFM_CH != CIN || d.OFM_CH != COUT) throw std::runtime_error("stream width does not match the layer");
    const size_t in_words = (size_t)d.IFM_ROW * d.IFM_COL * numReps, out_words = (size_t)d.OFM_ROW * d.OFM_COL * numReps;
    std::vector<uint8_t> hin(in_words * CIN), hout(out_words * COUT);
    in.drain(hin.data(), in_words);
    sicn_weights *dw = nullptr;
    check(sicn_weights_from_finn_tiles(&d, w.m_weights.data(), 8, bias.data(), &dw), "sicn_weights_from_finn_tiles");
    uint8_t *din = nullptr, *dout = nullptr;
    hip_check(hipMalloc((void **)&din, hin.size()), "hipMalloc");
    hip_check(hipMalloc((void **)&dout, hout.size()), "hipMalloc");
    hip_check(hipMemcpy(din, hin.data(), hin.size(), hipMemcpyHostToDevice), "hipMemcpy");
    const int rc = d.transposed ? sicn_deconv522(&d, dw, din, dout, (int)numReps, nullptr)
                                : sicn_conv2d(&d, dw, din, dout, (int)numReps, nullptr);
    hipError_t e = hipMemcpy(hout.data(), dout, hout.size(), hipMemcpyDeviceToHost);
    (void)hipFree(din);
    (void)hipFree(dout);
    sicn_weights_free(dw);
    check(rc, d.transposed ? "sicn_deconv522" : "sicn_conv2d");
    hip_check(e, "hipMemcpy");
    out.append(hout.data(), out_words);
}

template <int CIN, int COUT>
void conv2d(const sicn_layer_desc &d, const FixedPointWeights &weights, const std::vector<int8_t> &bias,
            stream<CIN> &in, stream<COUT> &out, unsigned numReps)
{
    if (d.transposed) throw std::runtime_error("conv2d called with a deconv descriptor");
    run_layer<CIN, COUT>(d, weights, bias, in, out, numReps);
}
template <int CIN, int COUT>
void deconv522(const sicn_layer_desc &d, const FixedPointWeights &weights, const std::vector<int8_t> &bias,
               stream<CIN> &in, stream<COUT> &out, unsigned numReps)
{
    if (!d.transposed) throw std::runtime_error("deconv522 called with a conv descriptor");
    run_layer<CIN, COUT>(d, weights, bias, in, out, numReps);
}

// Process-wide context standing in for the reference's compile-time configuration + PARAM tables.
struct Context {
    int width = 768, height = 512;  // config_nonsquare.h:5-7
    ParamSet params;
    bool loaded = false;
    static Context &get()
    {
        static Context c;
        return c;
    }
    static void set_image_size(int w, int h) { get().width = w; get().height = h; }
    static void load_params(const std::string &path) { get().params = ParamSet::load(path); get().loaded = true; }
    const LayerParams &layer(int n) const
    {
        if (!loaded || n >= (int)params.layers.size()) throw std::runtime_error("PARAM tables not loaded (Context::load_params)");
        return params.layers[n];
    }
};

// config_nonsquare.h channel / fold table
inline sicn_layer_desc net_desc(int layer, int width, int height)
{
    static const int tab[8][5] = {{3, 128, 3, 8, 0},    {128, 128, 8, 16, 0}, {128, 128, 8, 16, 0}, {128, 192, 8, 24, 0},
                                  {192, 128, 12, 16, 1}, {128, 128, 8, 16, 1}, {128, 128, 8, 16, 1}, {128, 3, 8, 3, 1}};
    int w = width, h = height;
    for (int l = 0; l < layer; l++) {
        if (tab[l][4]) { w *= 2; h *= 2; } else { w = (w + 1) / 2; h = (h + 1) / 2; }
    }
    return make_desc(tab[layer][0], tab[layer][1], tab[layer][2], tab[layer][3], w, h, tab[layer][4]);
}

inline void conv2d_layer0(stream<3> &in, stream<128> &out, unsigned numReps)
{
    Context &c = Context::get();
    conv2d<3, 128>(net_desc(0, c.width, c.height), c.layer(0).weights, c.layer(0).bias, in, out, numReps);
}

// input = the latent of a width x height image, i.e. (width/16) x (height/16) x 192
inline void deconv2d_layer4(stream<192> &in, stream<128> &out, unsigned numReps)
{
    Context &c = Context::get();
    deconv522<192, 128>(net_desc(4, c.width, c.height), c.layer(4).weights, c.layer(4).bias, in, out, numReps);
}

inline void eight_layers_net(stream<3> &in, stream<3> &out, unsigned numReps)
{
    Context &c = Context::get();
    sicn_layer_desc descs[8];
    sicn_weights *w[8] = {};
    sicn_net *net = nullptr;
    uint8_t *din = nullptr, *dout = nullptr, *ws = nullptr;
    std::vector<uint8_t> hin, hout;
    auto cleanup = [&]() {
        if (din) (void)hipFree(din);
        if (dout) (void)hipFree(dout);
        if (ws) (void)hipFree(ws);
        if (net) sicn_net_free(net);
        for (auto *p : w) sicn_weights_free(p);
    };
    try {
        for (int l = 0; l < 8; l++) {
            descs[l] = net_desc(l, c.width, c.height);
            check(sicn_weights_from_finn_tiles(&descs[l], c.layer(l).weights.m_weights.data(), 8, c.layer(l).bias.data(), &w[l]),
                  "sicn_weights_from_finn_tiles");
        }
        check(sicn_net_create(descs, w, 8, &net), "sicn_net_create");
        const size_t in_words = (size_t)c.width * c.height * numReps;
        const size_t out_words = (size_t)descs[7].OFM_ROW * descs[7].OFM_COL * numReps;
        hin.resize(in_words * 3);
        hout.resize(out_words * 3);
        in.drain(hin.data(), in_words);
        const size_t wsb = sicn_net_workspace_bytes(net, (int)numReps);
        hip_check(hipMalloc((void **)&din, hin.size()), "hipMalloc");
        hip_check(hipMalloc((void **)&dout, hout.size()), "hipMalloc");
        hip_check(hipMalloc((void **)&ws, wsb), "hipMalloc");
        hip_check(hipMemcpy(din, hin.data(), hin.size(), hipMemcpyHostToDevice), "hipMemcpy");
        check(sicn_eight_layers_net(net, din, dout, nullptr, (int)numReps, ws, wsb, nullptr), "sicn_eight_layers_net");
        hip_check(hipMemcpy(hout.data(), dout, hout.size(), hipMemcpyDeviceToHost), "hipMemcpy");
        out.append(hout.data(), out_words);
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

}  // namespace sicn_hls
#endif  // SICN_HLS_HPP
