// Internal declarations shared by the libsicn.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sicn.h"
#include "sicn_plan.h"

namespace sicn {

// Geometry of the implicit-GEMM kernels (k_mfma.hip).  One workgroup owns an "M tile" of
// TILE_Y x TILE_X positions of the M grid (conv: output pixels; deconv: input pixels) and all
// output channels.  LDS holds SUB sub-patches of (TILE_Y+2) x (TILE_X+2) positions x 32 bytes.
constexpr int TILE_Y = 8;
constexpr int TILE_X = 32;
constexpr int PATCH_Y = TILE_Y + 2;
constexpr int PATCH_X = TILE_X + 2;
constexpr int PATCH_PIX = PATCH_Y * PATCH_X;  // 340
constexpr int KSTEP = 32;                      // bytes of K per MFMA (v_mfma_i32_32x32x32_i8)

enum KernelKind : int { KK_GENERIC = 0, KK_MFMA_CONV, KK_MFMA_DECONV, KK_L0_RGB, KK_L7_RGB };

struct LayerGeom {
    int IW, IH, CIN, OW, OH, COUT, transposed;
};

inline LayerGeom geom_of(const sicn_layer_desc &d)
{
    return LayerGeom{d.IFM_ROW, d.IFM_COL, d.IFM_CH, d.OFM_ROW, d.OFM_COL, d.OFM_CH, d.transposed};
}

}  // namespace sicn

// One layer's device-resident parameters, in every layout a kernel family wants.
struct sicn_weights {
    int cin, cout, transposed;
    int8_t *d_w_okc;      // [cout][25*cin] int8, k = (ky*5+kx)*cin + c        (generic kernel)
    int8_t *d_bias;       // [cout] int8
    // implicit-GEMM tile stream (k_mfma.hip), or nullptr when the shape is not served by it:
    // n_steps tiles of [cout rows][32 B] in consumption order, rows permuted (sigma) and the two
    // 16-byte halves of a row swapped where the LDS swizzle wants it.
    int8_t *d_w_mfma;
    int mfma_steps;
    int8_t *d_w_mfma16;    // the same tile sequence laid out for v_mfma_i32_16x16x64_i8 (k_mfma16.hip)
    int8_t *d_w_mfma16x;   // deconv 128 -> 128: the tiles in the order the wide persistent kernel walks them (k_mfma16x.hip), or nullptr
    int8_t *d_bias_sigma;  // [cout] bias in sigma order == natural order (kept for clarity)
    // layer-0 (RGB -> cout) and layer-7 (cin -> RGB) layouts, or nullptr
    int8_t *d_w_l0;
    int8_t *d_w_l0g;       // layer 0, 128 channels: the A-operand image of the kernel that applies a GDN before its store (k_l0g.hip)
    int8_t *d_w_l7;
};

namespace sicn {

KernelKind pick_kernel(const sicn_layer_desc &d, const sicn_options &o);

// Scratch of a K-split launch (k_mfma16p.hip): `slices` partial tensors of the layer's output size each and the flag words of
// the tiles, inside the caller's workspace; nonce: see sicn_abi.hip (ksplit_nonce)
struct KSplitScratch {
    uint8_t *partials;          // [slices][n_images * out_bytes]
    size_t partial_stride;      // bytes between two slices' tensors
    unsigned long long *flags;  // [workgroups of the unsplit grid][KSPLIT_MAX]
    size_t n_flags;             // capacity in flag words
    unsigned long long nonce;
    unsigned long long *deal;   // round 5: DEAL_WORDS zeroed words for the wide persistent kernels' tile deal (k_mfma16x.hip: DealX), or nullptr
};
// the tile deal of the wide persistent kernels: 16 ticket counters (one per XCD) + one mailbox per workgroup
constexpr int DEAL_MAX_WORKGROUPS = WIDE_DEAL_MAX_WORKGROUPS;
constexpr int DEAL_WORDS = WIDE_DEAL_MAX_XCDS + DEAL_MAX_WORKGROUPS;
hipError_t launch_zero_words(unsigned long long *p, size_t words, hipStream_t stream);   // k_generic.hip
constexpr int KSPLIT_MAX = 3;   // 192 input channels = 3 channel-group pairs

// what plan_mfma (k_mfma16.hip) decides for a conv / deconv layer of the 128 / 192-channel shapes
struct MfmaPlan {
    int family;      // 0: k_mfma16_t, 1: the software-pipelined kernels (k_mfma16p.hip), 2: the wide persistent kernels (k_mfma16x.hip)
    int tile_x;      // 16 | 32 (families 0, 1)
    int split_n;     // output-channel slices (workgroups of blockIdx.y), 1 = none
    int split_k;     // K slices (blockIdx.z), 1 = none
    unsigned grid_x, grid_y, grid_z;
    int deal;        // family 2: 1 where the tile deal has its dynamic part (sicn_plan.h wide_deal_pays), given a workspace with room for it
};
MfmaPlan plan_mfma(const LayerGeom &g, int n_images, const sicn_options &o, const ChipGeom &chip);
// does `ks` hold the partial tensors and flag words the plan's K split needs?
inline bool ksplit_scratch_fits(const LayerGeom &g, int n_images, const MfmaPlan &p, const KSplitScratch *ks)
{
    if (!ks || !ks->partials || !ks->flags) return false;
    const size_t out_bytes = (size_t)g.OH * g.OW * g.COUT * (size_t)n_images;
    return ks->partial_stride >= out_bytes && ks->n_flags >= (size_t)p.grid_x * p.grid_y;
}
// library defaults: zeros overridden by the SICN_* environment as it was at load time (read once)
const sicn_options &default_options();
// build-time experiment switches of k_mfma.hip (SICN_MFMA_VARIANT, SICN_DEBUG_KERNEL, SICN_DEBUG_EXTRA_LDS), read once
struct DebugEnv { int mfma_variant, debug_kernel, extra_lds, no_deal; };   // no_deal (SICN_NO_DEAL=1, read at load): the static tile deal even with a workspace (A/B)
const DebugEnv &debug_env();

// Geometry of the CURRENT device (hipGetDevice), read once per device from hipDeviceProp_t: SICN_OK, or SICN_ENODEV when there
// is no device or it is not a gfx950 part (the code objects of this library exist for gfx950 only)
int chip_geom(ChipGeom *out);

// Launchers: enqueue on `stream`, return hipError_t of the launch.  `chip`: sizes every grid (sicn_plan.h).
// relu = false: store the lane BEFORE the sign-bit ReLU (input of the GDN extension, include/sicn_gdn.h)
hipError_t launch_generic(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                          int n_images, hipStream_t stream, bool relu = true);
// k_generic.hip: dst[n][h][w][c] = the top-left h x w corner of src[n][hs][ws][c], one launch
hipError_t launch_crop_nhwc(const uint8_t *src, uint8_t *dst, int n, int hs, int ws, int h, int w, int c, hipStream_t stream);
// in_layout / out_layout: LAYOUT_NHWC (the ABI layout) / LAYOUT_GROUP / LAYOUT_PHASE (k_common.hpp)
hipError_t launch_mfma(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                       int n_images, hipStream_t stream, int in_layout, int out_layout);
hipError_t launch_mfma16(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                         int n_images, hipStream_t stream, int in_layout, int out_layout, const sicn_options &o, const ChipGeom &chip,
                         bool relu = true, const KSplitScratch *ks = nullptr);
// k_mfma16x.hip: the wide persistent form — one workgroup of 4 waves per CU walks through 16 x 32-position tiles, 128 x 128 outputs
// per wave (accumulators in AGPRs, one wave per SIMD); grid_cap > 0 limits the number of workgroups (tests)
bool wide_supported(const LayerGeom &g);
size_t mfma16x_deconv_stream_bytes(int cin, int cout);   // 0 where the wide deconv does not exist
void pack_mfma16x_deconv_stream(const int8_t *w_okc, int cin, int cout, int8_t *dst);
hipError_t launch_wide(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images, hipStream_t stream,
                       int in_layout, int out_layout, bool relu, int grid_cap, const ChipGeom &chip, unsigned long long *deal = nullptr);
// k_mfma16p.hip: the software-pipelined conv / deconv kernels (tile_x = 16 | 32)
bool pipelined_supported(const LayerGeom &g, int tile_x);
hipError_t launch_pipelined(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out, int n_images,
                            hipStream_t stream, int in_layout, int out_layout, bool relu, int tile_x, bool split_channels,
                            const ChipGeom &chip, int split_k = 1, const KSplitScratch *ks = nullptr);
hipError_t launch_l0(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                     int n_images, hipStream_t stream, int out_layout, const sicn_options &o, const ChipGeom &chip, bool relu = true);
hipError_t launch_l7(const LayerGeom &g, const sicn_weights &w, const uint8_t *in, uint8_t *out,
                     int n_images, hipStream_t stream, int in_layout, const sicn_options &o, const ChipGeom &chip);

// Host-side weight packers (pure CPU, unit-testable without a GPU).
// w_okc: [cout][25*cin].  Returns bytes written into `dst` (size from *_bytes()).
size_t mfma_stream_bytes(int cin, int cout);
int mfma_stream_steps(int cin);
void pack_mfma_stream(const int8_t *w_okc, int cin, int cout, int transposed, int8_t *dst);
bool mfma_supported(int cin, int cout, int transposed);
bool mfma32_supported(int cin, int cout, int transposed);
size_t mfma16_stream_bytes(int cin, int cout);
void pack_mfma16_stream(const int8_t *w_okc, int cin, int cout, int transposed, int8_t *dst);

size_t l0_bytes(int cout);
void pack_l0(const int8_t *w_okc, int cout, int8_t *dst);
size_t l0g_bytes();
void pack_l0g(const int8_t *w_okc, int8_t *dst);   // cout = 128
size_t l7_bytes(int cin);
void pack_l7(const int8_t *w_okc, int cin, int8_t *dst);

}  // namespace sicn
