/*
 * sicn_gdn.h — fixed-point GDN / IGDN activation for the conv / deconv layers of sicn.h.
 *
 * EXTENSION BEYOND THE REFERENCE.  The reference has no GDN: activations.hpp:127-224 offers
 * PassThrough, Threshold and ChannelWise only, and the net's non-linearity is the sign-bit ReLU of
 * conv_nonsquare_top.cpp:273-275 / 189-191.  A layer given a `sicn_gdn` runs exactly the reference layer up to and
 * including the bias add (conv_nonsquare_top.cpp:272) and then applies this activation INSTEAD of that ReLU.
 * Parity status: UNPINNED (own specification, VERSION 2 since library 0.3: oracle/sicn_gdn_oracle.c, restated here; version 1 —
 * floor(2^16 / sqrt(n)) to 16 absolute bits, an arithmetic shift, x clamped at -127 — produced different bytes and is gone):
 *
 *   per pixel, C channels, v = the 8-bit lane after the bias add:
 *     x_i  = int8(v_i)
 *     n_i  = beta_i + sum_j gamma[i][j] * x_j^2                beta in [1, 65535], gamma in [0, 127]          (exact integer)
 *     nq_i = n_i rounded to 24 significant bits (nearest-even), cut to its top 11 significant bits
 *     GDN  (inverse = 0):  r_i = trunc11(2^(16-shift) (1 +  5 * 2^-16) / sqrt(nq_i))     trunc11: the exact real number cut to
 *     IGDN (inverse = 1):  r_i = trunc11(2^( 8-shift) (1 + 33 * 2^-16) * sqrt(nq_i))     11 significant bits, towards zero
 *     u_i  = x_i * r_i + 128 rounded once to IEEE binary32 (nearest-even): one fused multiply-add
 *     y_i  = clamp(nearest-even integer of u_i, 0, 255) - 128, stored as y_i mod 256
 *
 * A pure function of integers: every step is an exact integer operation, a correctly rounded IEEE-754 operation, or a root needed
 * to 11 bits whose rounding bias (5 / 33) keeps all 2048 possible nq mantissas >= 7.5 binary32 ulps from a step of trunc11, so a
 * 1-ulp hardware root decides every case the same way (sicn_gdn_selftest_roots proves it on the device for every n).  With beta /
 * gamma read as Q8 (256 = 1.0) shift = 12 is the textbook y = x / sqrt(beta + sum gamma x^2) (GDN) or y = x * sqrt(...) (IGDN).
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

/* 2: the specification above.  Encoder and decoder of a hyperprior bitstream must agree on it (the bytes a version-1 library produced
 * differ). */
int sicn_gdn_spec_version(void);

/* Test hook: r = trunc11((1 + b 2^-16) / sqrt(nq)) (inverse = 0) or trunc11((1 + b 2^-16) sqrt(nq)) (inverse = 1) exactly as the kernels
 * compute it (one root instruction, one multiply, two masks) against the defining integer inequalities, on the device, for every n in
 * [n_begin, n_begin + count) (n = 0 skipped), n_begin + count <= 2^31.  Synchronous.  Returns the number of n whose root differs
 * (0 = exact), < 0 on error. */
long long sicn_gdn_selftest_roots(int inverse, uint32_t n_begin, unsigned long long count);

#ifdef __cplusplus
}
#endif
#endif /* SICN_GDN_H */
