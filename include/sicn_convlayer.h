/*
 * sicn_convlayer.h — the generic finn-hlslib convolution layer surface on the GPU.
 *
 *   sicn_conv_layer_batch  <-  ConvLayer_Batch<ConvKernelDim, IFMChannels, IFMDim, OFMChannels, OFMDim,
 *                                              SIMD, PE, TSrcI, TDstI, TWeightI>
 *                                             (in, out, weights, activation, reps, r)       convlayer.h:89-125
 *      = ConvolutionInputGenerator<K, IFMCh, TSrcI::width, IFMDim, OFMDim, SIMD, 1>  (square image,
 *        stride 1, NO padding: OFMDim = IFMDim - K + 1; window order ky -> kx -> channel,
 *        slidingwindow.h:163-270)  ->  Matrix_Vector_Activate_Batch (mvau.hpp:87-179)
 *   sicn_convlayer_desc    <-  those template parameters plus the interpretation functors:
 *        TSrcI   = Slice<ap_uint<IN_BIT>> or Slice<ap_int<IN_BIT>>          (interpret.hpp:191-244)
 *        weights = FixedPointWeights<SIMD, ap_int<W_BIT>, PE, TILES>, TWeightI = Identity (weights.hpp:110-150)
 *        activation = PassThroughActivation<TA>                              (activations.hpp:127-134)
 *                   | ThresholdsActivation<NF, PE, NumTH, TA, TR, ActVal, comp::less<TA>> (activations.hpp:168-190)
 *        TA = ap_int<ACC_BIT> / ap_uint<ACC_BIT>: the accumulator type (mvau.hpp:112), every += wraps
 *        TDstI   = Slice<ap_(u)int<OUT_BIT>>: each output lane keeps the low OUT_BIT bits of the result
 *
 * In the reference this surface is only instantiated in a commented-out line
 * (conv_nonsquare_top.cpp:223), so there are no reference outputs to pin it with: parity status
 * UNPINNED (restated from the cited source; GPU == CPU restatement in the tests).  It is served by
 * an int8 MFMA implicit-GEMM kernel when IFM_CH is a multiple of 16 and by a shape-agnostic direct
 * kernel otherwise — a functional surface, not a tuned hot path (the hot path is sicn.h).
 *
 * Data model = the reference's stream words, byte for byte (round 5: sub-byte lanes).  Input: per pixel one word
 * `ap_uint<IFM_CH * IN_BIT>` (convlayer.h:100: hls::stream<ap_uint<IFMChannels * TSrcI::width>>), lane c in bits
 * [c * IN_BIT, (c + 1) * IN_BIT) (Slice<>, interpret.hpp:191-244), words stored little-endian and back to back:
 * [reps][IFM_DIM][IFM_DIM][IFM_CH * IN_BIT / 8] bytes, IN_BIT in {1, 2, 4, 8} (IFM_CH * IN_BIT a multiple of 8).  Output: per pixel
 * one word `ap_uint<OFM_CH * OUT_BIT>`, lane o in bits [o * OUT_BIT, (o + 1) * OUT_BIT): [reps][OFM_DIM][OFM_DIM][OFM_CH * OUT_BIT / 8]
 * bytes, OUT_BIT in {2, 4, 8, 16, 32} (OFM_CH * OUT_BIT a multiple of 8) — so the output of a ThresholdsActivation layer that emits
 * 2- or 4-bit lanes (activations.hpp:168-190) IS the input stream of the next layer, with no repacking in between.  With IN_BIT = 8
 * and OUT_BIT >= 8 this is one byte per input lane and one little-endian container per output lane, as before.  Device pointers.
 */
#ifndef SICN_CONVLAYER_H
#define SICN_CONVLAYER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SICN_ACT_PASSTHROUGH 0
#define SICN_ACT_THRESHOLDS 1

typedef struct sicn_convlayer_desc {
    int32_t K;          /* ConvKernelDim, 1..11                                                   */
    int32_t IFM_CH;     /* IFMChannels                                                            */
    int32_t IFM_DIM;    /* IFMDim (square)                                                        */
    int32_t OFM_CH;     /* OFMChannels                                                            */
    int32_t OFM_DIM;    /* OFMDim = IFM_DIM - K + 1                                               */
    int32_t SIMD, PE;   /* folds of the weight wire format                                        */
    int32_t IN_BIT;     /* 1, 2, 4 or 8: TSrcI::width; IFM_CH * IN_BIT must be a multiple of 8        */
    int32_t IN_SIGNED;  /* 0: Slice<ap_uint<IN_BIT>>, 1: Slice<ap_int<IN_BIT>> (sign-extended lanes)  */
    int32_t W_BIT;      /* 2..8, signed (ap_int<W_BIT>); SIMD*W_BIT <= 64                         */
    int32_t W_TILES;    /* (OFM_CH/PE) * (K*K*IFM_CH/SIMD)                                        */
    int32_t ACC_BIT;    /* 1..32                                                                  */
    int32_t ACC_SIGNED; /* TA = ap_int<ACC_BIT> (1) or ap_uint<ACC_BIT> (0)                       */
    int32_t OUT_BIT;    /* 2, 4, 8, 16 or 32: TDstI::width, the lane keeps the low OUT_BIT bits of the result; OFM_CH * OUT_BIT % 8 == 0 */
    int32_t activation; /* SICN_ACT_PASSTHROUGH or SICN_ACT_THRESHOLDS                            */
    int32_t NUM_TH;     /* NumTH (thresholds per output channel), 0 for pass-through, <= 1024     */
    int32_t ACT_VAL;    /* ActVal: result = ActVal + #{i : thresholds[pe][nf][i] < accu}          */
} sicn_convlayer_desc;

typedef struct sicn_convlayer_params sicn_convlayer_params;

int sicn_convlayer_validate(const sicn_convlayer_desc *desc); /* pure host check */

/* m_weights: HOST [PE][W_TILES] words (element s = sign-extended bits [W_BIT*s, W_BIT*(s+1))),
 * `word_bytes` in {1,2,4,8}.  thresholds: HOST int32 [PE][NF][NUM_TH] exactly as
 * ThresholdsActivation::m_thresholds (activations.hpp:172), or NULL for pass-through. */
int sicn_convlayer_params_create(const sicn_convlayer_desc *desc, const void *m_weights, int word_bytes,
                                 const int32_t *thresholds, sicn_convlayer_params **out);
void sicn_convlayer_params_free(sicn_convlayer_params *p);

/* ConvLayer_Batch(in, out, weights, activation, reps, r): enqueue on hip_stream, no sync. */
int sicn_conv_layer_batch(const sicn_convlayer_desc *desc, const sicn_convlayer_params *params, const uint8_t *in,
                          void *out, int reps, void *hip_stream);

/* The same with the kernel chosen by the caller (tests compare the two): AUTO = the implicit-GEMM MFMA kernel
 * when the shape allows it (IFM_CH % 16 == 0, IN_BIT = 8, OUT_BIT >= 8), DIRECT = always the direct kernel (one thread per output
 * byte or container; the only one that reads and writes sub-byte lanes). */
#define SICN_CONVLAYER_KERNEL_AUTO 0
#define SICN_CONVLAYER_KERNEL_DIRECT 1
int sicn_conv_layer_batch_kernel(const sicn_convlayer_desc *desc, const sicn_convlayer_params *params, const uint8_t *in,
                                 void *out, int reps, int kernel, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SICN_CONVLAYER_H */
