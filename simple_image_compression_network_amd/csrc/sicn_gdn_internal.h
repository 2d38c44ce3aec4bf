// Internal: the GDN / IGDN activation object (include/sicn_gdn.h) and its launchers (k_gdn.hip).
#pragma once
#include "../../include/sicn_gdn.h"
#include "sicn_internal.h"

struct sicn_gdn {
    int channels, inverse, shift;
    float kc;              // 2^(16 - shift) (1 + 5 2^-16) (GDN) or 2^(8 - shift) (1 + 33 2^-16) (IGDN): the root's scale and rounding bias
    uint32_t *d_beta;      // [C] natural channel order
    uint32_t *d_beta_mfma; // [C] beta + 128 * (row sum of gamma): the MFMA kernels carry the low digit of x^2 as lo - 128; nullptr as d_gamma_mfma
    int8_t *d_gamma;       // [C][C] natural order (generic kernel)
    int8_t *d_gamma_mfma;  // permuted LDS image of k_gdn, or nullptr when C is not 128 / 192
};

namespace sicn {
void pack_gdn_gamma(const uint8_t *gamma, int C, int8_t *dst);
// in place over n_images images of W x H positions in the given internal layout (k_common.hpp LAYOUT_*)
hipError_t launch_gdn(const sicn_gdn &g, uint8_t *data, int layout, int W, int H, int n_images, hipStream_t stream);
// layer 0 (RGB -> 128 channels) and its GDN / IGDN in one kernel (k_l0g.hip): needs w.d_w_l0g and g.d_gamma_mfma
hipError_t launch_l0_gdn(const LayerGeom &g, const sicn_weights &w, const sicn_gdn &gdn, const uint8_t *in, uint8_t *out, int n_images,
                         hipStream_t stream, int out_layout, const sicn_options &o, const ChipGeom &chip);
// layer 7 (128 channels -> RGB) reading the pre-activation lanes of the layer before it and applying that layer's GDN / IGDN on the
// way into its LDS window (k_l7g.hip): needs w.d_w_l7 and g.d_gamma_mfma
hipError_t launch_l7_gdn(const LayerGeom &g, const sicn_weights &w, const sicn_gdn &gdn, const uint8_t *in, uint8_t *out, int n_images,
                         hipStream_t stream, int in_layout, const sicn_options &o, const ChipGeom &chip);
hipError_t launch_gdn_generic(const sicn_gdn &g, uint8_t *data, long long n_pos, hipStream_t stream);
// the kernels' root (k_gdn_body.hpp: gdn_root) against exact integers for n_begin <= n < n_begin + count, on the device
hipError_t gdn_selftest_roots(int inverse, uint32_t n_begin, unsigned long long count, unsigned long long *mismatches);
}  // namespace sicn
