// libsicn.so — C ABI (include/sicn.h): descriptor validation, weight ingestion from the
// reference's FixedPointWeights tile format, kernel dispatch, layer chains with caller-provided
// workspace, per-layer hipEvent timing.  Host code only; kernels live in k_*.hip.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <random>
#include <vector>

#include "sicn_gdn_internal.h"
#include "sicn_internal.h"

using namespace sicn;

// ---- options: no mutable process-wide state. The environment is read exactly once (thread-safe static
// ---- initialisation at first use, in practice at load) into an immutable default; everything else
// ---- travels by value in sicn_options.
namespace sicn {
static int env_int(const char *name)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : 0;
}
const sicn_options &default_options()
{
    static const sicn_options o = [] {
        sicn_options d;
        std::memset(&d, 0, sizeof d);
        d.struct_bytes = (int32_t)sizeof(sicn_options);
        d.force_generic = env_int("SICN_FORCE_GENERIC") != 0;
        d.mfma_shape = env_int("SICN_MFMA_SHAPE");
        d.tile_x = env_int("SICN_TILE_X");
        d.strip_chunks = env_int("SICN_STRIP_CHUNKS");
        d.no_phase_layout = env_int("SICN_NO_PHASE_LAYOUT");
        d.split_n = env_int("SICN_SPLIT_N");
        d.wave_tile = env_int("SICN_WAVE_TILE");
        d.prefetch = env_int("SICN_PREFETCH");
        d.split_k = env_int("SICN_SPLIT_K");
        d.l7_loader = env_int("SICN_L7_LOADER");
        d.l0_form = env_int("SICN_L0_FORM");
        d.gdn_fuse = env_int("SICN_GDN_FUSE");
        // the same range checks a caller's struct gets (resolve_options): an out-of-range variable is ignored, loudly, once
        auto bad = [](const char *name, int32_t &v) {
            fprintf(stderr, "libsicn: ignoring out-of-range %s=%d\n", name, (int)v);
            v = 0;
        };
        if (d.mfma_shape != 0 && d.mfma_shape != 16 && d.mfma_shape != 32) bad("SICN_MFMA_SHAPE", d.mfma_shape);
#ifndef SICN_ALT_KERNELS
        if (d.mfma_shape == 32) bad("SICN_MFMA_SHAPE", d.mfma_shape);
#endif
        if (d.tile_x != 0 && d.tile_x != 16 && d.tile_x != 32) bad("SICN_TILE_X", d.tile_x);
        if (d.strip_chunks < 0) bad("SICN_STRIP_CHUNKS", d.strip_chunks);
        if (d.no_phase_layout < 0 || d.no_phase_layout > 2) bad("SICN_NO_PHASE_LAYOUT", d.no_phase_layout);
        if (d.split_n < 0 || d.split_n > 4) bad("SICN_SPLIT_N", d.split_n);
        if (d.wave_tile != 0 && d.wave_tile != 64 && d.wave_tile != 128) bad("SICN_WAVE_TILE", d.wave_tile);
        if (d.prefetch < 0 || d.prefetch > 3) bad("SICN_PREFETCH", d.prefetch);
        if (d.split_k < 0 || d.split_k > 4) bad("SICN_SPLIT_K", d.split_k);
        if (d.l7_loader < 0 || d.l7_loader > 2) bad("SICN_L7_LOADER", d.l7_loader);
        if (d.l0_form < 0 || d.l0_form > 2) bad("SICN_L0_FORM", d.l0_form);
        if (d.gdn_fuse < 0 || d.gdn_fuse > 2) bad("SICN_GDN_FUSE", d.gdn_fuse);
#ifndef SICN_ALT_KERNELS   // forms that measured a loss live in the ALT build only
        if (d.split_k > 1) bad("SICN_SPLIT_K", d.split_k);
        if (d.l7_loader == 2) bad("SICN_L7_LOADER", d.l7_loader);
        if (d.l0_form == 2) bad("SICN_L0_FORM", d.l0_form);
        if (d.gdn_fuse == 2) bad("SICN_GDN_FUSE", d.gdn_fuse);
#endif
        return d;
    }();
    return o;
}
// Geometry of the current device, read once per device ordinal from hipDeviceProp_t (immutable afterwards).  The code objects in
// this library are gfx950 only: any other architecture is SICN_ENODEV here, with one line on stderr, instead of a failed code
// object load at the first launch (the reference's error path is exit(-1), bnn-library.h:55).  SICN_N_CU (read at load) overrides
// the CU count — experiments only: e.g. what the launch planning of a 128-CU partition does on a whole chip.
int chip_geom(ChipGeom *out)
{
    constexpr int MAX_DEV = 64;
    struct Entry { std::once_flag once; int rc = SICN_ENODEV; ChipGeom geom{1, 1}; };
    static Entry table[MAX_DEV];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return SICN_ENODEV;
    Entry &e = table[dev];
    std::call_once(e.once, [&] {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            fprintf(stderr, "libsicn: device %d is %s; this library holds gfx950 (MI355X) kernels only\n", dev, prop.gcnArchName);
            return;
        }
        static const int forced_cu = env_int("SICN_N_CU");
        e.geom = chip_from_cus(forced_cu > 0 ? forced_cu : prop.multiProcessorCount);
        e.rc = SICN_OK;
    });
    if (e.rc == SICN_OK && out) *out = e.geom;
    return e.rc;
}

const DebugEnv &debug_env()
{
    static const DebugEnv d{env_int("SICN_MFMA_VARIANT"), env_int("SICN_DEBUG_KERNEL"), env_int("SICN_DEBUG_EXTRA_LDS"), env_int("SICN_NO_DEAL")};
    return d;
}
}  // namespace sicn

// a caller's options -> a validated copy (NULL = defaults; a shorter struct from an older build is zero-extended)
static int resolve_options(const sicn_options *in, sicn_options *out)
{
    *out = default_options();
    if (!in) return SICN_OK;
    if (in->struct_bytes < 8 || in->struct_bytes > (int32_t)sizeof(sicn_options)) return SICN_EINVAL;
    sicn_options o;
    std::memset(&o, 0, sizeof o);
    std::memcpy(&o, in, (size_t)in->struct_bytes);
    o.struct_bytes = (int32_t)sizeof(sicn_options);
    if (o.mfma_shape != 0 && o.mfma_shape != 16 && o.mfma_shape != 32) return SICN_EINVAL;
#ifndef SICN_ALT_KERNELS
    if (o.mfma_shape == 32) return SICN_EINVAL;   // the 32x32x32 family lives in the ALT build only (libsicn_alt.so)
#endif
    if (o.tile_x != 0 && o.tile_x != 16 && o.tile_x != 32) return SICN_EINVAL;
    if (o.strip_chunks < 0 || o.no_phase_layout < 0 || o.no_phase_layout > 2) return SICN_EINVAL;
    if (o.split_n < 0 || o.split_n > 4) return SICN_EINVAL;
    if (o.wave_tile != 0 && o.wave_tile != 64 && o.wave_tile != 128) return SICN_EINVAL;
    if (o.prefetch < 0 || o.prefetch > 3 || o.persistent_grid < 0) return SICN_EINVAL;
    if (o.split_k < 0 || o.split_k > 4 || o.l7_loader < 0 || o.l7_loader > 2 || o.l0_form < 0 || o.l0_form > 2 || o.gdn_fuse < 0 ||
        o.gdn_fuse > 2)
        return SICN_EINVAL;
#ifndef SICN_ALT_KERNELS
    // k_l0p, k_l7s, k_l7g and the K split measured a loss against the defaults: they are built into
    // libsicn_alt.so only, where their parity tests run (tests/alt_kernels_check.py)
    if (o.split_k > 1 || o.l7_loader == 2 || o.l0_form == 2 || o.gdn_fuse == 2) return SICN_EINVAL;
#endif
    *out = o;
    return SICN_OK;
}

extern "C" void sicn_options_init(sicn_options *opt)
{
    if (opt) *opt = default_options();
}

// 0.2: sicn_options.split_k (a reserved slot), sicn_net_workspace_bytes grows by the K-split scratch, sicn_codec_info is 44 bytes
// (stream_symbols), SICN_ENODEV for a device that is not gfx950
// 0.3: GDN / IGDN specification version 2 (include/sicn_gdn.h): the activation's BYTES change, sicn_gdn_selftest_roots_narrow is gone
extern "C" int sicn_version(void) { return 1000 * 0 + 3; }
extern "C" int sicn_gdn_spec_version(void) { return 2; }

extern "C" int sicn_has_alt_kernels(void)
{
#ifdef SICN_ALT_KERNELS
    return 1;
#else
    return 0;
#endif
}

extern "C" const char *sicn_strerror(int code)
{
    switch (code) {
    case SICN_OK: return "ok";
    case SICN_EINVAL: return "invalid descriptor or argument";
    case SICN_ENOMEM: return "out of memory";
    case SICN_ENODEV: return "no gfx950 device, or HIP runtime error";
    case SICN_ENOSPC: return "workspace too small";
    default: return "unknown error";
    }
}

// Shape preconditions of the reference, checked up front instead of CASSERT_DATAFLOW/exit(-1):
// slidingwindow.h:1259 (IFM_CH % SIMD), mvau.hpp:101-105 (folds), conv_nonsquare_top.cpp:94-95 and
// :246-259 (output dims), config_nonsquare.h (K=5, S=2, P=2, 8/8/4 bit).
static int validate_weights_fields(const sicn_layer_desc *d)
{
    if (!d) return SICN_EINVAL;
    if (d->IFM_CH <= 0 || d->OFM_CH <= 0 || d->SIMD <= 0 || d->PE <= 0) return SICN_EINVAL;
    if (d->SIMD > 16) return SICN_EINVAL;  // SIMD*4 bits must fit a 64-bit word
    if (d->IFM_CH % d->SIMD || d->OFM_CH % d->PE) return SICN_EINVAL;
    if ((long long)d->W_TILES != (long long)(d->OFM_CH / d->PE) * (25LL * d->IFM_CH / d->SIMD)) return SICN_EINVAL;
    if (d->transposed != 0 && d->transposed != 1) return SICN_EINVAL;
    return SICN_OK;
}

extern "C" int sicn_validate_desc(const sicn_layer_desc *d)
{
    int rc = validate_weights_fields(d);
    if (rc) return rc;
    if (d->K != 5 || d->S != 2 || d->P != 2) return SICN_EINVAL;
    if (d->IN_BIT != 8 || d->OUT_BIT != 8 || d->W_BIT != 4) return SICN_EINVAL;
    if (d->IFM_ROW <= 0 || d->IFM_COL <= 0) return SICN_EINVAL;
    if (d->IFM_ROW > (1 << 20) || d->IFM_COL > (1 << 20)) return SICN_EINVAL;
    if (d->transposed) {
        if (d->OFM_ROW != 2 * d->IFM_ROW || d->OFM_COL != 2 * d->IFM_COL) return SICN_EINVAL;
    } else if (d->OFM_ROW != (d->IFM_ROW + 1) / 2 || d->OFM_COL != (d->IFM_COL + 1) / 2)
        return SICN_EINVAL;
    return SICN_OK;
}

namespace sicn {
KernelKind pick_kernel(const sicn_layer_desc &d, const sicn_options &o)
{
    if (o.force_generic) return KK_GENERIC;
    if (!d.transposed && d.IFM_CH == 3 && d.OFM_CH == 128) return KK_L0_RGB;
    if (d.transposed && d.IFM_CH == 128 && d.OFM_CH == 3) return KK_L7_RGB;
    if (mfma_supported(d.IFM_CH, d.OFM_CH, d.transposed)) return d.transposed ? KK_MFMA_DECONV : KK_MFMA_CONV;
    return KK_GENERIC;
}
}  // namespace sicn

extern "C" const char *sicn_kernel_for(const sicn_layer_desc *d)
{
    if (sicn_validate_desc(d)) return "invalid";
    switch (pick_kernel(*d, default_options())) {
    case KK_L0_RGB: return "l0_rgb";
    case KK_L7_RGB: return "l7_rgb";
    case KK_MFMA_CONV: return "mfma_conv";
    case KK_MFMA_DECONV: return "mfma_deconv";
    default: return "generic";
    }
}

// ---- weights ----------------------------------------------------------------------------------
static bool upload(const void *host, size_t bytes, int8_t **dev)
{
    if (hipMalloc((void **)dev, bytes) != hipSuccess) { *dev = nullptr; return false; }
    if (hipMemcpy(*dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
    return true;
}

extern "C" void sicn_weights_free(sicn_weights *w)
{
    if (!w) return;
    if (w->d_w_okc) (void)hipFree(w->d_w_okc);
    if (w->d_bias) (void)hipFree(w->d_bias);
    if (w->d_w_mfma) (void)hipFree(w->d_w_mfma);
    if (w->d_w_mfma16) (void)hipFree(w->d_w_mfma16);
    if (w->d_w_mfma16x) (void)hipFree(w->d_w_mfma16x);
    if (w->d_w_l0) (void)hipFree(w->d_w_l0);
    if (w->d_w_l0g) (void)hipFree(w->d_w_l0g);
    if (w->d_w_l7) (void)hipFree(w->d_w_l7);
    delete w;
}

extern "C" int sicn_weights_from_finn_tiles(const sicn_layer_desc *d, const void *m_weights, int word_bytes,
                                            const int8_t *bias, sicn_weights **out)
{
    if (!out) return SICN_EINVAL;
    *out = nullptr;
    int rc = validate_weights_fields(d);
    if (rc || !m_weights || !bias) return SICN_EINVAL;
    if (chip_geom(nullptr) != SICN_OK) return SICN_ENODEV;   // no device, or not a gfx950 one: refuse before anything is uploaded
    if (word_bytes != 1 && word_bytes != 2 && word_bytes != 4 && word_bytes != 8) return SICN_EINVAL;
    if (d->SIMD * 4 > word_bytes * 8) return SICN_EINVAL;
    const int cin = d->IFM_CH, cout = d->OFM_CH, simd = d->SIMD, pe_n = d->PE, tiles = d->W_TILES;
    const int kk = 25 * cin, sf_n = kk / simd, nf_n = cout / pe_n;

    // FixedPointWeights (weights.hpp:110-150): W[o = nf*PE + pe][k = sf*SIMD + s] = sign-extended
    // nibble s of m_weights[pe][nf*SF + sf]; k = (ky*5 + kx)*IFM_CH + c (slidingwindow.h:1304-1325,
    // cross-checked by conv3_nonsquare_tb.cpp:546-571).
    std::vector<int8_t> w_okc;
    try {
        w_okc.resize((size_t)cout * kk);
    } catch (const std::bad_alloc &) { return SICN_ENOMEM; }
    const uint8_t *raw = (const uint8_t *)m_weights;
    for (int pe = 0; pe < pe_n; pe++)
        for (int nf = 0; nf < nf_n; nf++)
            for (int sf = 0; sf < sf_n; sf++) {
                const size_t idx = (size_t)pe * tiles + (size_t)nf * sf_n + sf;
                uint64_t word = 0;
                for (int b = 0; b < word_bytes; b++) word |= (uint64_t)raw[idx * word_bytes + b] << (8 * b);
                for (int s = 0; s < simd; s++) {
                    int v = (int)((word >> (4 * s)) & 15u);
                    if (v > 7) v -= 16;
                    w_okc[(size_t)(nf * pe_n + pe) * kk + sf * simd + s] = (int8_t)v;
                }
            }

    sicn_weights *w = new (std::nothrow) sicn_weights();
    if (!w) return SICN_ENOMEM;
    *w = sicn_weights{};
    w->cin = cin;
    w->cout = cout;
    w->transposed = d->transposed;
    bool ok = upload(w_okc.data(), w_okc.size(), &w->d_w_okc);
    {   // bias, padded to a multiple of 16 bytes so kernels may read it with 16-byte loads
        std::vector<int8_t> b(((size_t)cout + 15) / 16 * 16 + 16, 0);
        for (int i = 0; i < cout; i++) b[i] = bias[i];
        ok = ok && upload(b.data(), b.size(), &w->d_bias);
    }
    try {
        if (ok && mfma_supported(cin, cout, d->transposed)) {
#ifdef SICN_ALT_KERNELS
            if (mfma32_supported(cin, cout, d->transposed)) {   // second implementation (sicn_options.mfma_shape = 32), ALT build only
                std::vector<int8_t> s(mfma_stream_bytes(cin, cout));
                pack_mfma_stream(w_okc.data(), cin, cout, d->transposed, s.data());
                w->mfma_steps = mfma_stream_steps(cin);
                ok = upload(s.data(), s.size(), &w->d_w_mfma);
            }
#endif
            if (ok) {
                std::vector<int8_t> s16(mfma16_stream_bytes(cin, cout));
                pack_mfma16_stream(w_okc.data(), cin, cout, d->transposed, s16.data());
                ok = upload(s16.data(), s16.size(), &w->d_w_mfma16);
            }
            if (ok && d->transposed && mfma16x_deconv_stream_bytes(cin, cout)) {   // the wide persistent deconv walks the taps in its own order
                std::vector<int8_t> sx(mfma16x_deconv_stream_bytes(cin, cout));
                pack_mfma16x_deconv_stream(w_okc.data(), cin, cout, sx.data());
                ok = upload(sx.data(), sx.size(), &w->d_w_mfma16x);
            }
        }
        if (ok && !d->transposed && cin == 3 && cout % 32 == 0) {
            std::vector<int8_t> s(l0_bytes(cout));
            pack_l0(w_okc.data(), cout, s.data());
            ok = upload(s.data(), s.size(), &w->d_w_l0);
            if (ok && cout == 128) {   // the image of the kernel that applies a GDN before its store
                std::vector<int8_t> sg(l0g_bytes());
                pack_l0g(w_okc.data(), sg.data());
                ok = upload(sg.data(), sg.size(), &w->d_w_l0g);
            }
        }
        if (ok && d->transposed && cin == 128 && cout == 3) {
            std::vector<int8_t> s(l7_bytes(cin));
            pack_l7(w_okc.data(), cin, s.data());
            ok = upload(s.data(), s.size(), &w->d_w_l7);
        }
    } catch (const std::bad_alloc &) { ok = false; }
    if (!ok) {
        sicn_weights_free(w);
        return SICN_ENOMEM;
    }
    *out = w;
    return SICN_OK;
}

// ---- single layers ------------------------------------------------------------------------------
// Layout of the tensor between layer `p` (producer) and layer `c` (consumer) of a chain: the best
// one both kernels implement (k_common.hpp).  0 = NHWC, 1 = GROUP, 2 = PHASE.
// the kernel family a layer runs on: the RGB deconv kernel has no pre-ReLU output, so a layer-7-shaped layer with a
// GDN goes to the shape-agnostic kernel
static KernelKind layer_kernel(const sicn_layer_desc &d, const sicn_options &o, bool has_gdn)
{
    const KernelKind k = pick_kernel(d, o);
    return (has_gdn && k == KK_L7_RGB) ? KK_GENERIC : k;
}

static int link_layout(const sicn_layer_desc &p, const sicn_layer_desc &c, const sicn_options &o, bool p_gdn, bool c_gdn)
{
    const KernelKind kp = layer_kernel(p, o, p_gdn), kc = layer_kernel(c, o, c_gdn);
    const bool w_group = kp == KK_MFMA_CONV || kp == KK_MFMA_DECONV || kp == KK_L0_RGB;
    const bool w_phase = kp == KK_MFMA_DECONV;   // its outputs come one pixel parity at a time
    const bool r_group = kc == KK_MFMA_CONV || kc == KK_MFMA_DECONV || kc == KK_L7_RGB;
    const bool r_phase = kc == KK_MFMA_DECONV || kc == KK_L7_RGB;
    // experiments: no_phase_layout 1 = never, 2 = not towards the RGB layer
    const bool phase_ok = o.no_phase_layout == 0 || (o.no_phase_layout == 2 && kc != KK_L7_RGB);
    if (w_phase && r_phase && phase_ok) return 2;
    if (w_group && r_group) return 1;
    return 0;
}

// gdn != nullptr: the layer stores its lanes BEFORE the sign-bit ReLU and the GDN / IGDN kernel then rewrites them in
// place, in whatever layout the layer wrote (include/sicn_gdn.h; extension beyond the reference).  defer_gdn: the rewrite is
// left to the NEXT layer of a chain, which gets this layer's activation as its in_gdn (sicn_net_forward decides).
static int run_layer(const sicn_layer_desc *d, const sicn_weights *w, const uint8_t *in, uint8_t *out,
                     int n_images, hipStream_t stream, int want_transposed, const sicn_options &o, int in_layout = 0,
                     int out_layout = 0, const sicn_gdn *gdn = nullptr, const KSplitScratch *ks = nullptr, bool defer_gdn = false,
                     const sicn_gdn *in_gdn = nullptr)
{
    int rc = sicn_validate_desc(d);
    if (rc) return rc;
    if (!w || !in || !out || n_images < 0) return SICN_EINVAL;
    if (want_transposed >= 0 && d->transposed != want_transposed) return SICN_EINVAL;
    if (w->cin != d->IFM_CH || w->cout != d->OFM_CH || w->transposed != d->transposed) return SICN_EINVAL;
    if (n_images == 0) return SICN_OK;
    if (n_images > 65535) return SICN_EINVAL;
    if (gdn && gdn->channels != d->OFM_CH) return SICN_EINVAL;
    const LayerGeom g = geom_of(*d);
    const bool relu = gdn == nullptr;
    if (in_gdn && layer_kernel(*d, o, gdn != nullptr) != KK_L7_RGB) return SICN_EINVAL;   // only that kernel takes one — checked BEFORE anything is enqueued
    ChipGeom chip;
    if ((rc = chip_geom(&chip)) != SICN_OK) return rc;   // no device, or not a gfx950 one
    hipError_t e;
    switch (layer_kernel(*d, o, gdn != nullptr)) {
    case KK_L0_RGB:
        // layer 0 + activation in one kernel where it exists (128 channels): the pre-activation tensor never reaches HBM
        if (gdn && o.gdn_fuse != 1 && w->d_w_l0g && gdn->d_gamma_mfma && gdn->channels == 128) {
            e = launch_l0_gdn(g, *w, *gdn, in, out, n_images, stream, out_layout, o, chip);
            gdn = nullptr;
            break;
        }
        e = launch_l0(g, *w, in, out, n_images, stream, out_layout, o, chip, relu);
        break;
    case KK_L7_RGB:
        // in_gdn: the input holds the previous layer's pre-activation lanes, its activation is applied on the way in (k_l7g.hip)
#ifdef SICN_ALT_KERNELS
        if (in_gdn) {
            e = launch_l7_gdn(g, *w, *in_gdn, in, out, n_images, stream, in_layout, o, chip);
            break;
        }
#else
        if (in_gdn) return SICN_EINVAL;   // k_l7g: ALT build only (gdn_fuse = 2 is rejected, so no chain ever defers an activation)
#endif
        e = launch_l7(g, *w, in, out, n_images, stream, in_layout, o, chip);
        break;
    case KK_MFMA_CONV:
    case KK_MFMA_DECONV:
        // MFMA shape: 16x16x64 by default (higher sustained clock, k_mfma16.hip); mfma_shape = 32 selects
        // the 32x32x32 kernels of k_mfma.hip (a second implementation kept under test; reference shapes, ReLU only)
#ifdef SICN_ALT_KERNELS
        if (o.mfma_shape == 32 && relu && mfma32_supported(d->IFM_CH, d->OFM_CH, d->transposed)) {
            e = launch_mfma(g, *w, in, out, n_images, stream, in_layout, out_layout);
            break;
        }
#endif
        e = launch_mfma16(g, *w, in, out, n_images, stream, in_layout, out_layout, o, chip, relu, ks);
        break;
    default: e = launch_generic(g, *w, in, out, n_images, stream, relu); break;
    }
    if (e == hipSuccess && gdn && !defer_gdn) e = launch_gdn(*gdn, out, out_layout, d->OFM_ROW, d->OFM_COL, n_images, stream);
    if (e == hipErrorInvalidValue) return SICN_EINVAL;
    return e == hipSuccess ? SICN_OK : SICN_ENODEV;
}

extern "C" int sicn_conv2d(const sicn_layer_desc *d, const sicn_weights *w, const uint8_t *in, uint8_t *out,
                           int n_images, void *hip_stream)
{
    return run_layer(d, w, in, out, n_images, (hipStream_t)hip_stream, 0, default_options());
}

extern "C" int sicn_conv2d_opt(const sicn_layer_desc *d, const sicn_weights *w, const uint8_t *in, uint8_t *out,
                               int n_images, const sicn_options *opt, void *hip_stream)
{
    sicn_options o;
    int rc = resolve_options(opt, &o);
    return rc ? rc : run_layer(d, w, in, out, n_images, (hipStream_t)hip_stream, 0, o);
}

extern "C" int sicn_deconv522(const sicn_layer_desc *d, const sicn_weights *w, const uint8_t *in,
                              uint8_t *out, int n_images, void *hip_stream)
{
    return run_layer(d, w, in, out, n_images, (hipStream_t)hip_stream, 1, default_options());
}

extern "C" int sicn_deconv522_opt(const sicn_layer_desc *d, const sicn_weights *w, const uint8_t *in, uint8_t *out,
                                  int n_images, const sicn_options *opt, void *hip_stream)
{
    sicn_options o;
    int rc = resolve_options(opt, &o);
    return rc ? rc : run_layer(d, w, in, out, n_images, (hipStream_t)hip_stream, 1, o);
}

// ---- layer chains ---------------------------------------------------------------------------------
struct sicn_net {
    std::vector<sicn_layer_desc> descs;
    std::vector<const sicn_weights *> weights;
    std::vector<const sicn_gdn *> gdn;         // per layer, nullptr = the reference's ReLU
    sicn_options opt;                          // fixed at creation
    unsigned long long ks_tag;                 // K-split arrival words of this net carry this random tag (k_mfma16p.hip); low byte 0
    // profiling: the only state a launch changes.  One flat ring of event pairs; a forward call reserves the
    // slots of its layers with one atomic fetch_add, so calls on several streams / threads never share a slot.
    static constexpr int EV_RING = 8192;
    mutable std::atomic<bool> profile{false};
    mutable std::atomic<int> ev_next{0};       // slots handed out since the last reset (may run past EV_RING)
    std::vector<hipEvent_t> ev_begin, ev_end;  // [EV_RING], created by sicn_net_profile
    mutable std::unique_ptr<std::atomic<int>[]> ev_layer;   // [EV_RING] layer recorded in the slot (release on write, acquire on read:
                                                            // sicn_net_layer_ms may run beside forward calls on other threads)
};

static size_t out_bytes(const sicn_layer_desc &d) { return (size_t)d.OFM_COL * d.OFM_ROW * d.OFM_CH; }
static size_t in_bytes(const sicn_layer_desc &d) { return (size_t)d.IFM_COL * d.IFM_ROW * d.IFM_CH; }
static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

extern "C" int sicn_net_create(const sicn_layer_desc *descs, sicn_weights *const *weights, int n_layers,
                               sicn_net **out)
{
    return sicn_net_create_opt(descs, weights, n_layers, nullptr, out);
}

extern "C" int sicn_net_create_opt(const sicn_layer_desc *descs, sicn_weights *const *weights, int n_layers,
                                   const sicn_options *opt, sicn_net **out)
{
    return sicn_net_create_gdn(descs, weights, nullptr, n_layers, opt, out);
}

extern "C" int sicn_net_create_gdn(const sicn_layer_desc *descs, sicn_weights *const *weights, const sicn_gdn *const *gdn,
                                   int n_layers, const sicn_options *opt, sicn_net **out)
{
    if (!out) return SICN_EINVAL;
    *out = nullptr;
    if (!descs || !weights || n_layers <= 0 || n_layers > 64) return SICN_EINVAL;
    sicn_options o;
    if (int rc = resolve_options(opt, &o)) return rc;
    for (int i = 0; i < n_layers; i++) {
        int rc = sicn_validate_desc(&descs[i]);
        if (rc) return rc;
        const sicn_weights *w = weights[i];
        if (!w || w->cin != descs[i].IFM_CH || w->cout != descs[i].OFM_CH || w->transposed != descs[i].transposed)
            return SICN_EINVAL;
        if (i > 0 && (descs[i].IFM_CH != descs[i - 1].OFM_CH || descs[i].IFM_ROW != descs[i - 1].OFM_ROW ||
                      descs[i].IFM_COL != descs[i - 1].OFM_COL))
            return SICN_EINVAL;
        if (gdn && gdn[i] && gdn[i]->channels != descs[i].OFM_CH) return SICN_EINVAL;
    }
    sicn_net *net = new (std::nothrow) sicn_net();
    if (!net) return SICN_ENOMEM;
    try {
        net->descs.assign(descs, descs + n_layers);
        net->weights.assign(weights, weights + n_layers);
        net->gdn.assign((size_t)n_layers, nullptr);
        if (gdn) net->gdn.assign(gdn, gdn + n_layers);
        net->opt = o;
        std::random_device rd;
        do {
            net->ks_tag = (((unsigned long long)rd() << 32) | (unsigned long long)rd()) & ~0xffull;
        } while (net->ks_tag == 0);
    } catch (const std::exception &) {
        delete net;
        return SICN_ENOMEM;
    }
    *out = net;
    return SICN_OK;
}

extern "C" void sicn_net_free(sicn_net *net)
{
    if (!net) return;
    for (hipEvent_t e : net->ev_begin) (void)hipEventDestroy(e);
    for (hipEvent_t e : net->ev_end) (void)hipEventDestroy(e);
    delete net;
}

// Two ping-pong buffers, each large enough for the largest intermediate activation.
static size_t pingpong_slot_bytes(const sicn_net *net, int n_images)
{
    size_t mx = 0;
    for (size_t i = 0; i + 1 < net->descs.size(); i++) mx = mx > out_bytes(net->descs[i]) ? mx : out_bytes(net->descs[i]);
    return align256(mx * (size_t)n_images);
}

// K-split scratch behind the ping-pong buffers (k_mfma16p.hip): KSPLIT_MAX partial tensors of the largest output any layer of the
// chain would compute K-split on this chip, and one arrival word per workgroup of the largest such grid.  Small by construction:
// the split is only taken by grids of at most half a workgroup per CU.
struct KsNeed { size_t partial, words; };
static KsNeed ksplit_need(const sicn_net *net, int n_images, const ChipGeom &chip)
{
    KsNeed n{0, 0};
    for (size_t l = 0; l < net->descs.size(); l++) {
        const sicn_layer_desc &d = net->descs[l];
        const KernelKind k = pick_kernel(d, net->opt);
        if (k != KK_MFMA_CONV && k != KK_MFMA_DECONV) continue;
        const MfmaPlan p = plan_mfma(geom_of(d), n_images, net->opt, chip);
        if (p.split_k <= 1) continue;
        const size_t ob = align256(out_bytes(d) * (size_t)n_images), words = (size_t)p.grid_x * p.grid_y;
        n.partial = n.partial > ob ? n.partial : ob;
        n.words = n.words > words ? n.words : words;
    }
    return n;
}
static ChipGeom chip_or_default()
{
    ChipGeom c;
    return chip_geom(&c) == SICN_OK ? c : chip_from_cus(256);   // sizes can be asked for without a device: the whole MI355X
}

// round 5: the wide persistent kernels' tile deal (k_mfma16x.hip: DealX) — DEAL_WORDS zeroed words per layer, behind everything else
static size_t deal_bytes(const sicn_net *net) { return align256(net->descs.size() * (size_t)DEAL_WORDS * sizeof(unsigned long long)); }

extern "C" size_t sicn_net_workspace_bytes(const sicn_net *net, int n_images)
{
    if (!net || n_images <= 0) return 0;
    const KsNeed ks = ksplit_need(net, n_images, chip_or_default());
    return 2 * pingpong_slot_bytes(net, n_images) + KSPLIT_MAX * ks.partial + align256(ks.words * sizeof(unsigned long long)) + deal_bytes(net);
}

// Layer l's GDN / IGDN is applied by layer l + 1's kernel (k_l7g.hip) instead of k_gdn (sicn_options.gdn_fuse = 2): l + 1 is part of this call, takes the RGB
// deconv kernel, has no activation of its own, and layer l's output is an intermediate nobody else reads.
static bool gdn_moves_to_next(const sicn_net *net, int l, int last, int tap_layer)
{
    if (l >= last || l == tap_layer || !net->gdn[l] || net->gdn[l + 1]) return false;
    if (net->opt.gdn_fuse != 2) return false;   // measured: no gain over k_gdn + k_l7 (k_l7g.hip), so only on request
    const sicn_gdn *g = net->gdn[l];
    return layer_kernel(net->descs[l + 1], net->opt, false) == KK_L7_RGB && g->channels == 128 && g->d_gamma_mfma != nullptr &&
           net->weights[l + 1]->d_w_l7 != nullptr;
}

extern "C" int sicn_net_forward(const sicn_net *net, int first, int last, const uint8_t *in, uint8_t *out,
                                int tap_layer, uint8_t *tap_out, int n_images, void *workspace,
                                size_t workspace_bytes, void *hip_stream)
{
    if (!net || !in || !out || n_images < 0) return SICN_EINVAL;
    const int n_layers = (int)net->descs.size();
    if (first < 0 || last >= n_layers || first > last) return SICN_EINVAL;
    if (tap_layer >= 0 && (tap_layer < first || tap_layer > last || !tap_out)) return SICN_EINVAL;
    if (n_images == 0) return SICN_OK;
    const size_t slot = pingpong_slot_bytes(net, n_images);
    if (last > first && (!workspace || workspace_bytes < 2 * slot)) return SICN_ENOSPC;
    hipStream_t stream = (hipStream_t)hip_stream;
    uint8_t *pp[2] = {(uint8_t *)workspace, (uint8_t *)workspace + slot};
    // K-split scratch: behind the ping-pong buffers when the workspace is of the size sicn_net_workspace_bytes asks for.  A smaller
    // (older-sized) or absent workspace only switches the automatic split off; a forced one (options.split_k > 1) is SICN_ENOSPC.
    KSplitScratch ks_store{nullptr, 0, nullptr, 0, net->ks_tag, nullptr};
    const KSplitScratch *ks = nullptr;
    unsigned long long *deal_base = nullptr;   // the tile-deal words of this call's layers, zeroed below (a smaller / absent workspace: static deal)
    {
        ChipGeom chip;
        if (int rc = chip_geom(&chip)) return rc;
        const KsNeed need = ksplit_need(net, n_images, chip);
        const size_t ks_total = 2 * slot + KSPLIT_MAX * need.partial + align256(need.words * sizeof(unsigned long long));
        if (need.partial) {
            if (workspace && workspace_bytes >= ks_total) {
                ks_store.partials = (uint8_t *)workspace + 2 * slot;
                ks_store.partial_stride = need.partial;
                ks_store.flags = (unsigned long long *)((uint8_t *)workspace + 2 * slot + KSPLIT_MAX * need.partial);
                ks_store.n_flags = need.words;
                ks = &ks_store;
            } else if (net->opt.split_k > 1)
                return SICN_ENOSPC;
        }
        // does a layer of this call deal tiles dynamically?  (small inputs never do: no zeroing launch in front of them)
        bool any_deal = false;
        for (int l = first; l <= last && !any_deal; l++) {
            const KernelKind k = pick_kernel(net->descs[l], net->opt);
            any_deal = (k == KK_MFMA_CONV || k == KK_MFMA_DECONV) && plan_mfma(geom_of(net->descs[l]), n_images, net->opt, chip).deal;
        }
        if (any_deal && workspace && workspace_bytes >= ks_total + deal_bytes(net) && !debug_env().no_deal) {
            deal_base = (unsigned long long *)((uint8_t *)workspace + ks_total);
            // zeroed once per forward pass, by a kernel; every layer of the call gets its own DEAL_WORDS
            if (launch_zero_words(deal_base, deal_bytes(net) / sizeof(unsigned long long), stream) != hipSuccess) return SICN_ENODEV;
            ks = &ks_store;
        }
    }
    const uint8_t *cur = in;
    int cur_layout = 0;  // the chain's input is always NHWC
    // profiling: reserve this call's event slots (one per layer) in one atomic step
    int slot0 = -1;
    if (net->profile.load(std::memory_order_acquire)) {
        const int need = last - first + 1;
        const int at = net->ev_next.fetch_add(need, std::memory_order_relaxed);
        if (at + need <= sicn_net::EV_RING) slot0 = at;   // ring full: this call is not timed
    }
    const sicn_gdn *deferred = nullptr;    // the previous layer's activation, when it is this layer's to apply
    for (int l = first; l <= last; l++) {
        // a tapped layer (the latent) is written straight into the caller's buffer and the next layer reads it there: no copy
        // (round 3 copied it device-to-device behind the layer: 4.8 us per forward pass on small inputs, 15 us on 8 x 4K)
        uint8_t *dst = (l == last) ? out : (l == tap_layer ? tap_out : pp[(l - first) & 1]);
        // intermediates nobody outside sees travel in the grouped layout when both neighbours can
        const int out_layout = (l < last && l != tap_layer)
                                   ? link_layout(net->descs[l], net->descs[l + 1], net->opt, net->gdn[l] != nullptr, net->gdn[l + 1] != nullptr)
                                   : 0;
        const int slot = slot0 >= 0 ? slot0 + (l - first) : -1;
        if (slot >= 0) {
            net->ev_layer[slot].store(-1, std::memory_order_relaxed);   // becomes l once both events are recorded
            if (hipEventRecord(net->ev_begin[slot], stream) != hipSuccess) return SICN_ENODEV;
        }
        // the activation of layer l moves into layer l + 1's kernel where that kernel exists (128 channels -> RGB, k_l7g) and nobody
        // else sees layer l's output: the activated tensor is then never written
        const bool defer = gdn_moves_to_next(net, l, last, tap_layer);
        ks_store.deal = deal_base ? deal_base + (size_t)l * DEAL_WORDS : nullptr;
        int rc = run_layer(&net->descs[l], net->weights[l], cur, dst, n_images, stream, -1, net->opt, cur_layout, out_layout,
                           net->gdn[l], ks, defer, deferred);
        deferred = defer ? net->gdn[l] : nullptr;
        if (rc) return rc;
        if (slot >= 0) {
            if (hipEventRecord(net->ev_end[slot], stream) != hipSuccess) return SICN_ENODEV;
            net->ev_layer[slot].store(l, std::memory_order_release);
        }
        if (l == tap_layer && tap_out != dst) {
            if (hipMemcpyAsync(tap_out, dst, out_bytes(net->descs[l]) * (size_t)n_images, hipMemcpyDeviceToDevice,
                               stream) != hipSuccess)
                return SICN_ENODEV;
        }
        cur = dst;
        cur_layout = out_layout;
    }
    (void)in_bytes;
    return SICN_OK;
}

extern "C" int sicn_eight_layers_net(const sicn_net *net, const uint8_t *in, uint8_t *out, uint8_t *latent,
                                     int n_images, void *workspace, size_t workspace_bytes, void *hip_stream)
{
    if (!net) return SICN_EINVAL;
    const int n = (int)net->descs.size();
    const int tap = latent ? 3 : -1;
    if (latent && n < 4) return SICN_EINVAL;
    return sicn_net_forward(net, 0, n - 1, in, out, tap, latent, n_images, workspace, workspace_bytes, hip_stream);
}

extern "C" int sicn_net_profile(sicn_net *net, int enable)
{
    if (!net) return SICN_EINVAL;
    if (enable && net->ev_begin.empty()) {
        const size_t n = (size_t)sicn_net::EV_RING;
        std::vector<hipEvent_t> a, b;
        try {
            a.reserve(n);
            b.reserve(n);
            net->ev_layer.reset(new std::atomic<int>[n]);
            for (size_t i = 0; i < n; i++) net->ev_layer[i].store(-1, std::memory_order_relaxed);
        } catch (const std::bad_alloc &) { return SICN_ENOMEM; }
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++) {
            hipEvent_t e;
            if ((ok = hipEventCreate(&e) == hipSuccess)) a.push_back(e);
            if (ok && (ok = hipEventCreate(&e) == hipSuccess)) b.push_back(e);
        }
        if (!ok) {   // all or nothing: a half-made ring would be indexed out of range later
            for (hipEvent_t e : a) (void)hipEventDestroy(e);
            for (hipEvent_t e : b) (void)hipEventDestroy(e);
            return SICN_ENODEV;
        }
        net->ev_begin.swap(a);
        net->ev_end.swap(b);
    }
    net->profile.store(enable != 0, std::memory_order_release);
    return SICN_OK;
}

extern "C" int sicn_net_layer_ms(sicn_net *net, int reset, float *ms_sum, int *launches)
{
    if (!net || !ms_sum || !launches) return SICN_EINVAL;
    const size_t n_layers = net->descs.size();
    for (size_t l = 0; l < n_layers; l++) {
        ms_sum[l] = 0.f;
        launches[l] = 0;
    }
    int used = net->ev_next.load(std::memory_order_acquire);
    if (used > sicn_net::EV_RING) used = sicn_net::EV_RING;
    if (net->ev_begin.empty()) used = 0;
    for (int i = 0; i < used; i++) {
        const int l = net->ev_layer[i].load(std::memory_order_acquire);
        if (l < 0 || (size_t)l >= n_layers) continue;   // reserved by a call that failed before recording
        if (hipEventSynchronize(net->ev_end[i]) != hipSuccess) return SICN_ENODEV;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, net->ev_begin[i], net->ev_end[i]) != hipSuccess) return SICN_ENODEV;
        ms_sum[l] += ms;
        launches[l]++;
    }
    if (reset) {
        for (int i = 0; i < used; i++) net->ev_layer[i].store(-1, std::memory_order_relaxed);
        net->ev_next.store(0, std::memory_order_release);
    }
    return SICN_OK;
}

// ---- crop of NHWC batches (one launch; see k_crop_nhwc) -----------------------------------------------------------------------------
extern "C" int sicn_crop_nhwc(const uint8_t *src, uint8_t *dst, int n_images, int src_h, int src_w, int h, int w, int channels,
                              void *hip_stream)
{
    if (n_images < 0 || h < 0 || w < 0 || channels <= 0 || h > src_h || w > src_w) return SICN_EINVAL;
    if ((size_t)n_images * h * w && (!src || !dst)) return SICN_EINVAL;
    if (chip_geom(nullptr) != SICN_OK) return SICN_ENODEV;
    const hipError_t e = launch_crop_nhwc(src, dst, n_images, src_h, src_w, h, w, channels, (hipStream_t)hip_stream);
    return e == hipSuccess ? SICN_OK : SICN_ENODEV;
}

// ---- launch planning, inspectable without a GPU (tests/test_abi_load.py): what would this layer launch on a chip of n_cu CUs? ----
extern "C" int sicn_debug_plan(const sicn_layer_desc *d, int n_images, const sicn_options *opt, int n_cu, int32_t out[12])
{
    sicn_options o;
    if (int rc = resolve_options(opt, &o)) return rc;
    if (int rc = sicn_validate_desc(d)) return rc;
    if (!out || n_images <= 0 || n_cu <= 0) return SICN_EINVAL;
    const ChipGeom chip = chip_from_cus(n_cu);
    const LayerGeom g = geom_of(*d);
    for (int i = 0; i < 12; i++) out[i] = 0;
    out[0] = chip.n_cu;
    out[1] = chip.n_xcd;
    const KernelKind k = pick_kernel(*d, o);
    out[2] = (int)k;
    if (k == KK_MFMA_CONV || k == KK_MFMA_DECONV) {
        const MfmaPlan p = plan_mfma(g, n_images, o, chip);
        out[3] = p.family; out[4] = p.tile_x; out[5] = p.split_n; out[6] = p.split_k;
        out[7] = (int)p.grid_x; out[8] = (int)p.grid_y; out[9] = (int)p.grid_z;
        out[10] = p.deal;
    } else if (k == KK_L7_RGB && o.gdn_fuse == 2) {   // as the layer runs behind a layer with an activation in a chain (k_l7g)
        const int tiles_x = (g.IW + L7G_PLAN_COLS - 1) / L7G_PLAN_COLS, steps_y = (g.IH + L7G_PLAN_ROWS - 1) / L7G_PLAN_ROWS;
        const int yc = l7g_chunks(tiles_x, n_images, steps_y, o.strip_chunks, 1, chip);
        out[7] = tiles_x * yc * n_images; out[8] = 1; out[9] = 1; out[10] = yc; out[11] = (steps_y + yc - 1) / yc;
    } else if (k == KK_L7_RGB) {
        const int tiles_x = (g.IW + 31) / 32, steps_y = (g.IH + 3) / 4;
        const int yc = l7_chunks(tiles_x, n_images, steps_y, o.strip_chunks, chip);
        out[7] = (int)xcd_grid_size((long)tiles_x * yc * n_images, chip.n_xcd); out[8] = 1; out[9] = 1; out[10] = yc;
    } else if (k == KK_L0_RGB) {
        const int tiles_x = (g.OW + 31) / 32, tiles_y = (g.OH + 7) / 8;
        const L0Cut c = l0_chunks(tiles_x, tiles_y, n_images, 9, o.strip_chunks, chip);
        out[7] = tiles_x; out[8] = c.y_chunks; out[9] = n_images; out[10] = c.y_chunks; out[11] = c.ty_per;
    }
    return SICN_OK;
}
// host mirror of the kernels' XCD-aware work list: the item of workgroup `block`, or -1 (a bijection for every XCD count)
extern "C" long long sicn_debug_xcd_item(long long block, long long n_items, int n_xcd)
{
    return n_xcd > 0 ? (long long)xcd_item_of((long)block, (long)n_items, n_xcd) : -1;
}

// ---- GDN / IGDN activation objects (include/sicn_gdn.h; extension beyond the reference) ------------------------
extern "C" void sicn_gdn_free(sicn_gdn *g)
{
    if (!g) return;
    if (g->d_beta) (void)hipFree(g->d_beta);
    if (g->d_beta_mfma) (void)hipFree(g->d_beta_mfma);
    if (g->d_gamma) (void)hipFree(g->d_gamma);
    if (g->d_gamma_mfma) (void)hipFree(g->d_gamma_mfma);
    delete g;
}

extern "C" long long sicn_gdn_selftest_roots(int inverse, uint32_t n_begin, unsigned long long count)
{
    if ((inverse != 0 && inverse != 1) || (unsigned long long)n_begin + count > (1ull << 31)) return SICN_EINVAL;
    unsigned long long bad = 0;
    if (sicn::gdn_selftest_roots(inverse, n_begin, count, &bad) != hipSuccess) return SICN_ENODEV;
    return (long long)bad;
}

extern "C" int sicn_gdn_create(int channels, int inverse, int shift, const uint32_t *beta, const uint8_t *gamma, sicn_gdn **out)
{
    if (!out) return SICN_EINVAL;
    *out = nullptr;
    if (channels <= 0 || channels > 1024 || (inverse != 0 && inverse != 1) || shift < 1 || shift > 24 || !beta || !gamma)
        return SICN_EINVAL;
    for (int i = 0; i < channels; i++) {
        if (beta[i] < 1 || beta[i] > 65535) return SICN_EINVAL;
        for (int j = 0; j < channels; j++)
            if (gamma[(size_t)i * channels + j] > 127) return SICN_EINVAL;
    }
    sicn_gdn *g = new (std::nothrow) sicn_gdn();
    if (!g) return SICN_ENOMEM;
    // kc = 2^s (1 + b 2^-16), s = 16 - shift / 8 - shift, b = 5 / 33 (oracle/sicn_gdn_oracle.c): 17 significant bits, exact in binary32
    *g = sicn_gdn{channels, inverse, shift, std::ldexp((float)(65536 + (inverse ? 33 : 5)), (inverse ? 8 : 16) - shift - 16), nullptr, nullptr, nullptr, nullptr};
    bool ok = upload(beta, (size_t)channels * 4, (int8_t **)&g->d_beta) &&
              upload(gamma, (size_t)channels * channels, &g->d_gamma);
    if (ok && (channels == 128 || channels == 192)) {
        try {
            std::vector<int8_t> img((size_t)channels * channels);
            pack_gdn_gamma(gamma, channels, img.data());
            std::vector<uint32_t> beta_mfma((size_t)channels);   // the low digit of x^2 goes through the MFMA as lo - 128
            for (int i = 0; i < channels; i++) {
                uint32_t row = 0;
                for (int j = 0; j < channels; j++) row += gamma[(size_t)i * channels + j];
                beta_mfma[(size_t)i] = beta[i] + 128u * row;
            }
            ok = upload(img.data(), img.size(), &g->d_gamma_mfma) && upload(beta_mfma.data(), beta_mfma.size() * 4, (int8_t **)&g->d_beta_mfma);
        } catch (const std::bad_alloc &) { ok = false; }
    }
    if (!ok) {
        sicn_gdn_free(g);
        return SICN_ENOMEM;
    }
    *out = g;
    return SICN_OK;
}

extern "C" int sicn_gdn_apply(const sicn_gdn *g, uint8_t *lanes, long long n_positions, void *hip_stream)
{
    if (!g || n_positions < 0 || (n_positions && !lanes)) return SICN_EINVAL;
    hipStream_t stream = (hipStream_t)hip_stream;
    // chunks of < 2 GiB (the kernel addresses a chunk through one buffer descriptor with 31-bit offsets)
    const long long chunk = ((1LL << 31) - 4096) / g->channels;
    for (long long p = 0; p < n_positions; p += chunk) {
        const long long n = n_positions - p < chunk ? n_positions - p : chunk;
        hipError_t e = launch_gdn(*g, lanes + (size_t)p * g->channels, 0 /* NHWC */, (int)n, 1, 1, stream);
        if (e == hipErrorInvalidValue) return SICN_EINVAL;
        if (e != hipSuccess) return SICN_ENODEV;
    }
    return SICN_OK;
}

extern "C" int sicn_conv2d_gdn(const sicn_layer_desc *d, const sicn_weights *w, const sicn_gdn *gdn, const uint8_t *in,
                               uint8_t *out, int n_images, const sicn_options *opt, void *hip_stream)
{
    sicn_options o;
    int rc = resolve_options(opt, &o);
    return rc ? rc : run_layer(d, w, in, out, n_images, (hipStream_t)hip_stream, 0, o, 0, 0, gdn);
}

extern "C" int sicn_deconv522_gdn(const sicn_layer_desc *d, const sicn_weights *w, const sicn_gdn *gdn, const uint8_t *in,
                                  uint8_t *out, int n_images, const sicn_options *opt, void *hip_stream)
{
    sicn_options o;
    int rc = resolve_options(opt, &o);
    return rc ? rc : run_layer(d, w, in, out, n_images, (hipStream_t)hip_stream, 1, o, 0, 0, gdn);
}
