/*
 * sicn.h — C ABI of libsicn.so, the MI355X (gfx950) implementation of the integer conv/deconv
 * transform path of shengjie-chen/simple_image_compression_network.
 *
 * Every entry point replaces one piece of the reference's HLS interface (paths relative to the
 * reference tree):
 *
 *   sicn_conv2d            <- template conv2d<...>(weights, bias, in, out, numReps)
 *                             conv_nonsquare_top.cpp:198-280; conv2d_layer0, :282-286
 *   sicn_deconv522         <- template deconv522<...>(weights, bias, in, out, numReps)
 *                             conv_nonsquare_top.cpp:71-195; deconv2d_layer4, :288-291
 *   sicn_eight_layers_net  <- eight_layers_net(in, out, numReps), conv_nonsquare_top.cpp:295-357
 *   sicn_weights_from_finn_tiles
 *                          <- FixedPointWeights<SIMD, ap_int<4>, PE, TILES>::m_weights[PE][TILES]
 *                             weights.hpp:110-150 and the bias table
 *                             FixedPointWeights<1, ap_int<8>, 1, OFM_CH> (memdata_nonsquare.h:16)
 *   sicn_layer_desc        <- the CONV_n_* macro set, config_nonsquare.h:2-16
 *
 * Data model.  An `hls::stream<ap_uint<C*8>>` carrying H*W words (channel c in bits [8c,8c+8),
 * conv3_nonsquare_tb.cpp:807-808) is byte-identical to a row-major [H][W][C] uint8 array; that
 * array, in DEVICE memory, is the only tensor format of this ABI.  `n_images` images are
 * concatenated ([n][H][W][C]).  The reference's `numReps` is only meaningful at 1
 * (conv_nonsquare_top.cpp:111,133,184,246,268 iterate one image); `n_images` here is a true batch
 * of independent single-image reference calls.
 *
 * Numerics: bit-exact with the HLS C-simulation.  The MVAU accumulator is ap_uint<8>
 * (mvau.hpp:112, activations.hpp:112-115,127-134) so results are defined mod 2^8:
 *   out = relu7((sum_k in[k]*W[o][k] + bias[o]) mod 256),  relu7(v) = v >= 128 ? 0 : v.
 *
 * Conventions: all functions return 0 or a negative errno-style code (never exit(), never throw —
 * the reference's CASSERT_DATAFLOW exit(-1), bnn-library.h:55, becomes SICN_EINVAL).  Buffers are
 * owned by the caller; the library never frees caller memory.  Entry points that take a
 * `hip_stream` only enqueue work on that hipStream_t (NULL = default stream) and perform no
 * allocation or synchronisation, so they can be captured into a hipGraph.  Handles are immutable
 * after creation; concurrent launches on different streams (and from different host threads) are
 * allowed provided each uses its own workspace.  The library keeps NO mutable process-wide state:
 * kernel-selection knobs travel in a `sicn_options` value (per call / per net); the SICN_* environment
 * variables are read exactly once, when the library is loaded, into the defaults `sicn_options_init`
 * hands out.  The only state a launch touches in a handle is the optional profiling ring of a net,
 * whose slots are reserved with an atomic counter.
 *
 * Size limits of the specialised kernels (SICN_EINVAL beyond them): one image's input and output
 * < 2 GiB per layer (31-bit offsets inside an image: loads and stores go through buffer descriptors
 * whose range check is the padding / masking; 8K RGB images fit: 8192 x 4320 x 128 B / 4 = 1.1 GB at
 * layer 0's output), the RGB input tensor of a batch < 2 GiB, n_images <= 65535.
 *
 * Naming trap inherited from the reference: IFM_ROW / OFM_ROW are WIDTHS (x, fast dimension),
 * IFM_COL / OFM_COL are HEIGHTS (conv_nonsquare_top.cpp:283-285).
 */
#ifndef SICN_H
#define SICN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SICN_OK 0
#define SICN_EINVAL (-22)   /* bad descriptor / argument (reference: CASSERT_DATAFLOW)          */
#define SICN_ENOMEM (-12)   /* host or device allocation failed                                */
#define SICN_ENODEV (-19)   /* no usable gfx950 device / HIP runtime error at launch           */
#define SICN_ENOSPC (-28)   /* caller-provided workspace too small                             */

/* config_nonsquare.h:2-16, one struct per layer, plus which template the layer instantiates. */
typedef struct sicn_layer_desc {
    int32_t K;          /* CONV_n_K        kernel size, must be 5                               */
    int32_t S;          /* CONV_n_S        stride, must be 2                                    */
    int32_t P;          /* CONV_n_P        padding, must be 2                                   */
    int32_t IFM_CH;     /* CONV_n_IFM_CH   input channels                                       */
    int32_t IFM_ROW;    /* CONV_n_IFM_ROW  input WIDTH                                          */
    int32_t IFM_COL;    /* CONV_n_IFM_COL  input HEIGHT                                         */
    int32_t OFM_CH;     /* CONV_n_OFM_CH   output channels                                      */
    int32_t OFM_ROW;    /* CONV_n_OFM_ROW  output WIDTH  (conv: ceil(in/2); deconv: 2*in)       */
    int32_t OFM_COL;    /* CONV_n_OFM_COL  output HEIGHT                                        */
    int32_t SIMD;       /* CONV_n_SIMD     input lanes per weight word (wire format only)       */
    int32_t PE;         /* CONV_n_PE       output lanes per weight tile (wire format only)      */
    int32_t IN_BIT;     /* CONV_n_IN_BIT   must be 8                                            */
    int32_t OUT_BIT;    /* CONV_n_OUT_BIT  must be 8                                            */
    int32_t W_BIT;      /* CONV_n_W_BIT    must be 4                                            */
    int32_t W_TILES;    /* CONV_n_W_TILES  = (OFM_CH/PE) * (25*IFM_CH/SIMD)                     */
    int32_t transposed; /* 0: conv2d<> ; 1: deconv522<>                                         */
} sicn_layer_desc;

/* Kernel-selection knobs (tests, A/B experiments; the defaults are the product).  Plain data, copied by
 * every function that takes it.  Always start from sicn_options_init(). */
typedef struct sicn_options {
    int32_t struct_bytes;    /* sizeof(sicn_options) of the caller's build                          */
    int32_t force_generic;   /* 1: the shape-agnostic kernel (k_generic) for every layer            */
    int32_t mfma_shape;      /* 0 / 16: v_mfma_i32_16x16x64_i8 kernels; 32: the 32x32x32 kernels (ALT build only) */
    int32_t tile_x;          /* 0: by layer shape and grid size; 16 / 32: force that M-tile width   */
    int32_t strip_chunks;    /* 0: automatic; n: cut the vertical strips of the RGB layers into n   */
    int32_t no_phase_layout; /* 0: default; 1: never use the PHASE layout; 2: not towards layer 7   */
    int32_t split_n;         /* 0: automatic (grids of at most 128 tiles: half the CUs idle); 1: never; > 1: always — the */
                             /*    pipelined 8 x 16 kernels give each workgroup 64 of the layer's 128 / 192 output     */
                             /*    channels (k_mfma16p.hip, launch_p)                                                 */
    int32_t wave_tile;       /* 0: automatic; 64: always the 64 x 128-per-wave kernels (two waves per SIMD); 128: the  */
                             /*    WIDE PERSISTENT kernels (k_mfma16x.hip: one workgroup of 4 waves per CU walks through */
                             /*    16 x 32 tiles, 128 x 128 outputs per wave) wherever they exist (conv / deconv        */
                             /*    128 -> 128), whatever the grid size; automatic: from 4 tiles per CU on              */
    int32_t prefetch;        /* 0: automatic; 1: never, 2 / 3: wherever it exists — the software-pipelined kernels      */
                             /*    (k_mfma16p.hip) for the shapes and grids the wide kernels do not take               */
    int32_t persistent_grid; /* 0: one workgroup per CU; n: at most n (rounded down to a multiple of the XCD count)       */
                             /*    workgroups for the wide persistent kernels — tests use it to make every workgroup    */
                             /*    walk through many tiles of a small input                                            */
    /* The four forms below MEASURED A LOSS against the defaults on MI355X and are built into the ALT  */
    /* library only (libsicn_alt.so, `make ALT=1`; sicn_has_alt_kernels() == 1), where their parity tests run.  The product library     */
    /* (libsicn.so) answers split_k > 1, l7_loader = 2, l0_form = 2 and gdn_fuse = 2 with SICN_EINVAL.                                   */
    int32_t split_k;         /* 0 / 1: never split K (there is no automatic K split: it measured a loss at every size); > 1 (ALT only): */
                             /*    forced where the form exists (the channel-split 8 x 16 kernels inside a net chain, whose workspace   */
                             /*    holds the partial tensors) — K is split into channel-group pairs over IFM_CH / 64 workgroups, exact  */
    int32_t l7_loader;       /* layer 7 (k_l7): 0 / 1: four waves, every wave requests its share of a step's rows; 2 (ALT only): the     */
                             /*    loader-wave form k_l7s (a fifth wave issues all row requests; 8 % slower)      */
    int32_t l0_form;         /* layer 0: 0 / 1: one workgroup per run of tiles (k_l0); 2 (ALT only): the persistent kernel k_l0p (two    */
                             /*    workgroups per CU walk many runs, persistent_grid caps them; measured 11 % slower)        */
    int32_t gdn_fuse;        /* layers with a sicn_gdn: 0: layer 0 with 128 channels applies its activation itself, before its          */
                             /*    one store (k_l0g); 1: never (layer kernel, then k_gdn in place); 2 (ALT only): 0 + the 128 -> RGB     */
                             /*    layer of a chain applies the activation of the layer before it on the way in (k_l7g: measured no      */
                             /*    faster than k_gdn + k_l7)                                                              */
    int32_t reserved[2];
} sicn_options;

typedef struct sicn_weights sicn_weights; /* one layer's weights+bias, resident on the device  */
typedef struct sicn_net sicn_net;         /* a chain of layers (the 8-layer net, or any chain) */

/* Library / device ------------------------------------------------------------------------- */
int sicn_version(void);                  /* 1000*major + minor                                 */
int sicn_has_alt_kernels(void);          /* 1: this build carries the alternate kernel families (the 32x32x32 MFMA kernels of
                                          * k_mfma.hip, `make ALT=1` -> libsicn_alt.so: parity tests only); 0: the product build,
                                          * which rejects sicn_options.mfma_shape = 32 with SICN_EINVAL */
const char *sicn_strerror(int code);
/* Device rule.  The library holds gfx950 code objects only: an entry point that would touch a device whose gcnArchName does not
 * start with "gfx950" returns SICN_ENODEV (one line on stderr) — weights upload, layers, nets.  Grids, strip cuts and the
 * XCD-aware tile order are sized from hipDeviceProp_t of the CURRENT device, read once per device: n_cu = multiProcessorCount,
 * n_xcd = the largest power of two <= 8 that leaves at least 20 CUs per XCD (256 CUs -> 8, a DPX partition of 128 -> 4, QPX
 * 64 -> 2, CPX 32 -> 1; hipDeviceProp_t carries no XCD count, and the value only steers which tiles share an L2).
 * sicn_debug_plan shows what a layer would launch on a chip of n_cu CUs, without a GPU: out[] = { n_cu, n_xcd, kernel kind
 * (0 generic, 1 mfma conv, 2 mfma deconv, 3 layer 0, 4 layer 7), mfma family (0 plain, 1 pipelined, 2 wide persistent), tile_x,
 * split_n, split_k, grid x, grid y, grid z, strip chunks (wide persistent: 1 = part of the tiles dealt dynamically), layer-0 tiles per
 * run }.  sicn_debug_xcd_item is the host mirror of
 * the kernels' workgroup -> work item mapping (-1: padding workgroup). */
/* dst[n][h][w][c] = the top-left h x w corner of every image of src[n][src_h][src_w][c] (device pointers, one launch, enqueue
 * only).  A deconv522 doubles a size that a conv2d rounded up, so a tensor rebuilt by deconvs can be one row / column larger than
 * the one it mirrors (the hyperprior's scale map against the latent); this is the crop. */
int sicn_crop_nhwc(const uint8_t *src, uint8_t *dst, int n_images, int src_h, int src_w, int h, int w, int channels,
                   void *hip_stream);
int sicn_debug_plan(const sicn_layer_desc *desc, int n_images, const sicn_options *opt, int n_cu, int32_t out[12]);
long long sicn_debug_xcd_item(long long block, long long n_items, int n_xcd);
int sicn_validate_desc(const sicn_layer_desc *desc); /* pure host check, no GPU needed         */
/* Fills *opt with the library defaults (= all zero, overridden by the SICN_MFMA_SHAPE, SICN_TILE_X,
 * SICN_STRIP_CHUNKS, SICN_NO_PHASE_LAYOUT, SICN_SPLIT_N, SICN_SPLIT_K, SICN_L7_LOADER, SICN_L0_FORM, SICN_GDN_FUSE, SICN_WAVE_TILE, SICN_PREFETCH, SICN_FORCE_GENERIC environment
 * variables as they were when the library was loaded; out-of-range values are ignored with one warning on stderr). */
void sicn_options_init(sicn_options *opt);

/* Weights ---------------------------------------------------------------------------------- */
/* Ingests the reference wire format verbatim.  `m_weights` = HOST array [PE][W_TILES] of words
 * holding SIMD nibbles (element s in bits [4s,4s+4), weights.hpp:134-139), each word stored in
 * `word_bytes` (1,2,4 or 8) little-endian bytes; `bias` = HOST int8[OFM_CH].  Only desc fields
 * IFM_CH, OFM_CH, SIMD, PE, W_TILES, transposed are used (weights do not depend on image size).
 * Synchronous (uploads to the current device).  A handle is immutable after creation: any number of launches, streams,
 * host threads and captured graphs may use it at the same time (the persistent kernels keep no scheduler state in a handle:
 * tiles are dealt statically, and the dynamic part of the wide kernels' deal lives in the workspace of the call, below). */
int sicn_weights_from_finn_tiles(const sicn_layer_desc *desc, const void *m_weights, int word_bytes,
                                 const int8_t *bias, sicn_weights **out);
void sicn_weights_free(sicn_weights *w);

/* Single layers ---------------------------------------------------------------------------- */
/* in_nhwc : DEVICE [n_images][IFM_COL][IFM_ROW][IFM_CH] uint8
 * out_nhwc: DEVICE [n_images][OFM_COL][OFM_ROW][OFM_CH] uint8 (values 0..127)                  */
int sicn_conv2d(const sicn_layer_desc *desc, const sicn_weights *w, const uint8_t *in_nhwc,
                uint8_t *out_nhwc, int n_images, void *hip_stream);
int sicn_deconv522(const sicn_layer_desc *desc, const sicn_weights *w, const uint8_t *in_nhwc,
                   uint8_t *out_nhwc, int n_images, void *hip_stream);
/* The same with explicit options (NULL = defaults). */
int sicn_conv2d_opt(const sicn_layer_desc *desc, const sicn_weights *w, const uint8_t *in_nhwc,
                    uint8_t *out_nhwc, int n_images, const sicn_options *opt, void *hip_stream);
int sicn_deconv522_opt(const sicn_layer_desc *desc, const sicn_weights *w, const uint8_t *in_nhwc,
                       uint8_t *out_nhwc, int n_images, const sicn_options *opt, void *hip_stream);
/* Name of the kernel family that will serve `desc` under the default options ("l0_rgb", "mfma_conv",
 * "mfma_deconv", "l7_rgb", "generic"); static string. */
const char *sicn_kernel_for(const sicn_layer_desc *desc);

/* Layer chains ----------------------------------------------------------------------------- */
/* descs[i+1] input dims/channels must equal descs[i] output dims/channels.  The net keeps
 * references to `weights` (caller keeps them alive). */
int sicn_net_create(const sicn_layer_desc *descs, sicn_weights *const *weights, int n_layers,
                    sicn_net **out);
/* The same with explicit options, fixed for the life of the net (NULL = defaults). */
int sicn_net_create_opt(const sicn_layer_desc *descs, sicn_weights *const *weights, int n_layers,
                        const sicn_options *opt, sicn_net **out);
void sicn_net_free(sicn_net *net);
/* Bytes of DEVICE scratch `sicn_net_forward` needs for a batch of n_images: two ping-pong activation buffers and, for chains
 * whose grids are small enough for the K split (sicn_options.split_k), the slices' partial output tensors and one arrival word
 * per workgroup.  The scratch needs NO initialisation: an arrival word only counts when it carries the net's random 56-bit tag,
 * and the workgroup that finishes a tile clears it (any other content, e.g. uninitialised memory, reads as "nobody arrived").
 * Depends on the current device's CU count (K split is a small-grid measure); asked without a device it assumes 256 CUs.
 * Behind those: 528 words per layer for the wide persistent kernels' tile deal (one ticket counter per XCD + one mailbox per
 * workgroup, k_mfma16x.hip DealX), which sicn_net_forward zeroes itself (one small kernel at the head of the call).  A workspace
 * without room for them (a size computed by library 0.2) still works: those kernels then deal all their tiles statically.
 * Like the ping-pong buffers, a workspace serves ONE call in flight at a time. */
size_t sicn_net_workspace_bytes(const sicn_net *net, int n_images);
/* Runs layers [first_layer, last_layer] of the chain.  `tap_layer` >= 0 additionally delivers that
 * layer's output (e.g. 3 = the latent, conv_3_out, conv_nonsquare_top.cpp:322-325) in `tap_out` (NHWC): an inner
 * tapped layer is written straight into `tap_out` and the next layer reads it from there (no copy), so `tap_out`
 * must not overlap `in`, `out` or the workspace and must stay untouched until the call has run. */
int sicn_net_forward(const sicn_net *net, int first_layer, int last_layer, const uint8_t *in,
                     uint8_t *out, int tap_layer, uint8_t *tap_out, int n_images, void *workspace,
                     size_t workspace_bytes, void *hip_stream);
/* eight_layers_net(in, out, numReps): the whole chain; latent_or_null receives layer 3's output. */
int sicn_eight_layers_net(const sicn_net *net, const uint8_t *in, uint8_t *out,
                          uint8_t *latent_or_null, int n_images, void *workspace,
                          size_t workspace_bytes, void *hip_stream);

/* Per-layer device timing (measurement aid) ------------------------------------------------ */
/* When enabled, sicn_net_forward brackets every layer launch with hipEvents on the launch stream.
 * sicn_net_layer_ms synchronises on the last recorded events and returns, per layer, the SUM of
 * milliseconds and the number of launches since the last reset.  Forward calls on several streams
 * may record concurrently (slots are reserved atomically; when the ring of 8192 slots is full, further
 * launches are simply not timed).  sicn_net_profile / sicn_net_layer_ms themselves must not run
 * concurrently with forward calls on the same net. */
int sicn_net_profile(sicn_net *net, int enable);
int sicn_net_layer_ms(sicn_net *net, int reset, float *ms_sum /*[n_layers]*/,
                      int *launches /*[n_layers]*/);

#ifdef __cplusplus
}
#endif
#endif /* SICN_H */
