/*
 * sicn_gdn.h — fixed-point GDN / IGDN activation for the conv / deconv layers of sicn.h.
 *
 * EXTENSION BEYOND THE REFERENCE.  The reference has no GDN: activations.hpp:127-224 offers
 * PassThrough, Threshold and ChannelWise only, and the net's non-linearity is the sign-bit ReLU of
 * conv_nonsquare_top.cpp:273-275 / 189-191.  A layer given a `sicn_gdn` runs exactly the reference layer up to and
 * including the bias add (conv_nonsquare_top.cpp:272) and then applies this activation INSTEAD of that ReLU.
 * Parity status: UNPINNED (own specification: oracle/sicn_gdn_oracle.c, restated here):
 *
 *   per pixel, C channels, v = the 8-bit lane after the bias add read as int8:
 *     x_i = max(v_i, -127)
 *     n_i = beta_i + sum_j gamma[i][j] * x_j^2                 beta in [1, 65535], gamma in [0, 127]
 *     GDN  (inverse = 0):  r_i = floor(2^16 / sqrt(n_i)) = max{ r : r^2 * n_i <= 2^32 }
 *     IGDN (inverse = 1):  r_i = floor(2^8  * sqrt(n_i)) = max{ r : r^2 <= n_i * 2^16 }
 *     y_i = clamp((x_i * r_i + 2^(shift-1)) >> shift, -128, 127)   (arithmetic shift), stored as y_i mod 256
 *
 * All integer, bit-exact by definition.  With beta / gamma read as Q8 (256 = 1.0) shift = 12 is the textbook
 * y = x / sqrt(beta + sum gamma x^2) (GDN) or y = x * sqrt(...) (IGDN).
 *
 * Pointers are DEVICE pointers unless named *_host.  Launch functions only enqueue on hip_stream.
 */
#ifndef SICN_GDN_H
#define SICN_GDN_H

#include "sicn.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sicn_gdn sicn_gdn; /* one activation's beta / gamma, resident on the device */

/* beta_host: uint32[channels] in [1, 65535]; gamma_host: uint8[channels][channels] in [0, 127], row i = output
 * channel; shift in [1, 24]; inverse 0 = GDN, 1 = IGDN.  Synchronous (uploads to the current device). */
int sicn_gdn_create(int channels, int inverse, int shift, const uint32_t *beta_host, const uint8_t *gamma_host,
                    sicn_gdn **out);
void sicn_gdn_free(sicn_gdn *g);

/* The activation alone, IN PLACE over [n_positions][channels] lanes (an NHWC tensor of any image shape). */
int sicn_gdn_apply(const sicn_gdn *g, uint8_t *lanes_nhwc, long long n_positions, void *hip_stream);

/* conv2d<> / deconv522<> with the activation in place of the ReLU (gdn == NULL: the reference layer).  Which kernels run is not
 * part of the contract — the bytes are: by default the layer kernel stores its pre-activation lanes and the activation kernel
 * rewrites them in place, except for conv2d 3 -> 128 channels, which applies the activation itself before its one store
 * (sicn_options.gdn_fuse, sicn.h). */
int sicn_conv2d_gdn(const sicn_layer_desc *desc, const sicn_weights *w, const sicn_gdn *gdn, const uint8_t *in_nhwc,
                    uint8_t *out_nhwc, int n_images, const sicn_options *opt, void *hip_stream);
int sicn_deconv522_gdn(const sicn_layer_desc *desc, const sicn_weights *w, const sicn_gdn *gdn, const uint8_t *in_nhwc,
                       uint8_t *out_nhwc, int n_images, const sicn_options *opt, void *hip_stream);

/* A chain whose layer i uses gdn[i] (NULL entries: the reference ReLU); gdn[i]'s channel count must equal
 * descs[i].OFM_CH.  The net keeps references to the activations (caller keeps them alive).  With sicn_options.gdn_fuse = 2 an
 * intermediate layer's activation may be applied by the NEXT layer's kernel on the way in (128 channels -> RGB); a tapped layer's
 * output and the chain's last output are always the activated bytes. */
int sicn_net_create_gdn(const sicn_layer_desc *descs, sicn_weights *const *weights, const sicn_gdn *const *gdn,
                        int n_layers, const sicn_options *opt, sicn_net **out);

/* Test hook: the kernels' two integer roots — floor(2^16 / sqrt(n)) (inverse = 0) and floor(2^8 sqrt(n)) (inverse = 1), which
 * the device code gets from a float estimate and an integer fix-up — against integer bisection, on the device, for every
 * n in [n_begin, n_begin + count).  Synchronous.  Returns the number of n whose root differs (0 = exact), < 0 on error. */
long long sicn_gdn_selftest_roots(int inverse, uint32_t n_begin, unsigned long long count);
/* The same for the two-test form of floor(2^8 sqrt(n)) that the MFMA kernels (C <= 192: n < 2^29) use; n_begin + count <= 2^29. */
long long sicn_gdn_selftest_roots_narrow(uint32_t n_begin, unsigned long long count);

#ifdef __cplusplus
}
#endif
#endif /* SICN_GDN_H */
