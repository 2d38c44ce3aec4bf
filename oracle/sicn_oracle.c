/*
 * ORACLE — test infrastructure, NOT product code.
 *
 * Plain-C CPU restatement of the reference's conv/deconv transform path, stage by stage, the way
 * the HLS C-simulation executes it: every `hls::stream` is a FIFO that one stage fills completely
 * before the next stage drains it, so each stage here is a function from one
 * byte array to the next.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libsicn.so) never links or calls it.
 *
 * Pinning: PINNED by reference code for the arithmetic (closed form, weight unpack, padding, zero-stuffing, bias /
 * ReLU): the reference's own golden convolution conv_nonsquare<> (conv.hpp:91-123) is compiled UNMODIFIED in the
 * build container (oracle/ref_harness.cpp -> oracle/_ref/libsicn_refconv.so, `make -C oracle ref`) and
 * tests/golden/make_ref_conv_vectors.py ran it on seeded random layers (random nibble weights per PE, pixels
 * >= 128, odd sizes, Cin/Cout in {3,128,192}) -> tests/golden/ref_conv_vectors.npz, and layer by layer over the
 * whole net with the PARAM tables at 256x256 and 768x512 -> tests/golden/ref_conv_hashes.json (which reproduces
 * all 24 hashes of SURVEY.md Appendix A).  Every form in this file — including the stage-by-stage dataflow
 * form — must reproduce those bytes (tests/test_oracle_golden.py).  NOT anchored by reference code: the dataflow
 * stages as such (sliding-window FSM, MVAU fold order, width converters): conv_nonsquare_top.cpp and the
 * finn-hlslib headers need Vivado-HLS 2020.1 ap_int.h / hls_stream.h / ap_axi_sdata.h (README:5), which this
 * image lacks, and no stand-ins are written.  They are restated from the cited lines and checked for equality
 * with the pinned closed form, exactly the check the reference's own testbench makes
 * (conv3_nonsquare_tb.cpp:1068-1104), restated as sicn_or_naive_*.
 *
 * Tensor layout: row-major [H][W][C] uint8 == stream of H*W words of C*8 bits with channel c in
 * bits [8c, 8c+8) (conv3_nonsquare_tb.cpp:807-808, 1080).  A stream of SIMD*8-bit words is the
 * same bytes in the same order (StreamingDataWidthConverter_Batch splits LSB-first,
 * streamtools.h:477-495, and concatenates first-word-lowest, streamtools.h:503-523).
 *
 * Names: x = fast dimension = reference "ROW" macros (IFM_ROW = width), y = "COL" (height).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SICN_OR_OK 0
#define SICN_OR_EINVAL (-22)
#define SICN_OR_ENOMEM (-12)
#define SICN_OR_EUNDERRUN (-61) /* a stage tried to read an empty stream (C-sim would warn/hang) */

typedef struct {
    int32_t K, S, P, IFM_CH, IFM_ROW, IFM_COL, OFM_CH, OFM_ROW, OFM_COL;
    int32_t SIMD, PE, IN_BIT, OUT_BIT, W_BIT, W_TILES, transposed;
} sicn_or_layer_desc; /* same field order as include/sicn.h sicn_layer_desc */

/* ------------------------------------------------------------------------------------------ */
/* a1: FMPadding_nonsquare (streamtools.h:361-406), called via padding2 (top:59-69) with       */
/*     Padding_x = Padding_y = 2*PADDING, PaddingStyle 2 -> PaddingUp = PaddingLeft = PADDING.  */
/*     out dims are the PADDED dims (OutputDim_x/y).                                           */
/* ------------------------------------------------------------------------------------------ */
void sicn_or_fm_padding(const uint8_t *in, uint8_t *out, int out_x, int out_y, int pad_x,
                        int pad_y, int ch)
{
    const int up = pad_y / 2 + ((pad_y % 2) > 0), left = pad_x / 2 + ((pad_x % 2) > 0);
    const int down = pad_y - up, right = pad_x - left;
    size_t rd = 0, wr = 0;
    for (int y = 0; y < out_y; y++)
        for (int x = 0; x < out_x; x++) {
            if (y < up || y >= out_y - down || x < left || x >= out_x - right)
                memset(out + wr, 0, (size_t)ch);
            else {
                memcpy(out + wr, in + rd, (size_t)ch);
                rd += (size_t)ch;
            }
            wr += (size_t)ch;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a3: ConvolutionInputGenerator_NonSquare (slidingwindow.h:1242-1353), FSM restated line by    */
/*     line with words of `simd` bytes.  Returns words written, or SICN_OR_EUNDERRUN.           */
/*     `in_words` = words available in the input stream.                                       */
/* ------------------------------------------------------------------------------------------ */
long long sicn_or_swg_nonsquare_fsm(const uint8_t *in, long long in_words, uint8_t *out, int kx_dim,
                                    int ky_dim, int ifm_ch, int ifm_x, int ifm_y, int ofm_x,
                                    int ofm_y, int simd, int stride_x, int stride_y)
{
    if (ifm_ch % simd) return SICN_OR_EINVAL; /* CASSERT_DATAFLOW, slidingwindow.h:1259 */
    const unsigned mf = (unsigned)(ifm_ch / simd);
    const unsigned number_blocks = (unsigned)(ky_dim / stride_y + 1);
    const unsigned block_words = (unsigned)stride_x * (unsigned)ifm_x * mf;
    uint8_t *buf = (uint8_t *)malloc((size_t)number_blocks * block_words * (size_t)simd);
    if (!buf) return SICN_OR_ENOMEM;
    const unsigned cycles_write_block = (unsigned)ofm_x * (unsigned)kx_dim * (unsigned)ky_dim * mf;
    const unsigned cycles_read_block = block_words;
    const unsigned max_cycles =
        cycles_write_block > cycles_read_block ? cycles_write_block : cycles_read_block;
    const unsigned long long base_iter =
        (unsigned long long)ifm_x * (unsigned)ky_dim * mf + (unsigned long long)ofm_y * max_cycles;
    unsigned counter_internal_block = 0, current_block_write = 0, current_line = 0, read_block = 0;
    unsigned inp = 0, ofm_yc = 0, ofm_xc = 0, k_y = 0, k_x = 0, count_simd = 0;
    long long rd = 0, wr = 0;
    long long rc = 0;
    for (unsigned long long i = 0; i < base_iter; i++) {
        if (inp < (unsigned)ifm_x * (unsigned)ky_dim * mf) { /* initial fill, :1281 */
            if (rd >= in_words) { rc = SICN_OR_EUNDERRUN; break; }
            memcpy(buf + ((size_t)current_block_write * block_words + current_line) * simd,
                   in + (size_t)rd * simd, (size_t)simd);
            rd++;
            current_line++;
            inp++;
            if (current_line == block_words) {
                current_line = 0;
                current_block_write++;
                if (current_block_write == number_blocks) current_block_write = 0;
                read_block++;
                counter_internal_block = 0;
            }
        } else {
            if (counter_internal_block < cycles_write_block - 1) { /* :1297 */
                unsigned current_block_read = current_block_write + 1 + k_y / (unsigned)stride_y;
                if (current_block_read >= number_blocks) current_block_read -= number_blocks;
                /* :1302 uses IFMDim_y here; it is multiplied by k_y % Stride_y (== 0 at stride 1) */
                unsigned line = ((k_y % (unsigned)stride_y) * (unsigned)ifm_y +
                                 ofm_xc * (unsigned)stride_x + k_x) * mf + count_simd;
                memcpy(out + (size_t)wr * simd,
                       buf + ((size_t)current_block_read * block_words + line) * simd, (size_t)simd);
                wr++;
                count_simd++;
                if (count_simd == mf) {
                    count_simd = 0;
                    k_x++;
                    if (k_x == (unsigned)kx_dim) {
                        k_x = 0;
                        k_y++;
                        if (k_y == (unsigned)ky_dim) {
                            k_y = 0;
                            ofm_xc++;
                            if (ofm_xc == (unsigned)ofm_x) {
                                ofm_xc = 0;
                                ofm_yc++;
                                if (ofm_yc == (unsigned)ofm_y) {
                                    ofm_yc = 0;
                                    inp = 0;
                                }
                            }
                        }
                    }
                }
            }
            if (counter_internal_block < cycles_read_block - 1 &&
                read_block < (unsigned)ifm_y / (unsigned)stride_y) { /* :1327 */
                if (rd >= in_words) { rc = SICN_OR_EUNDERRUN; break; }
                memcpy(buf + ((size_t)current_block_write * block_words + current_line) * simd,
                       in + (size_t)rd * simd, (size_t)simd);
                rd++;
                current_line++;
                if (current_line == block_words) {
                    current_line = 0;
                    read_block++;
                    current_block_write++;
                    if (current_block_write == number_blocks) current_block_write = 0;
                }
            }
            counter_internal_block++;
            if (counter_internal_block == max_cycles - 1) counter_internal_block = 0; /* :1345 */
        }
    }
    free(buf);
    return rc ? rc : wr;
}

/* The same stage as the plain im2col it is meant to be (stride 1): for oy, ox: for ky, kx: the
 * ifm_ch bytes of pixel (oy+ky, ox+kx).  Used to state the FSM's valid domain in the tests. */
void sicn_or_im2col_s1(const uint8_t *in, uint8_t *out, int kdim, int ch, int ifm_x, int ofm_x,
                       int ofm_y)
{
    size_t wr = 0;
    for (int oy = 0; oy < ofm_y; oy++)
        for (int ox = 0; ox < ofm_x; ox++)
            for (int ky = 0; ky < kdim; ky++)
                for (int kx = 0; kx < kdim; kx++) {
                    memcpy(out + wr, in + ((size_t)(oy + ky) * ifm_x + (ox + kx)) * ch, (size_t)ch);
                    wr += (size_t)ch;
                }
}

/* a4: "get real windows" (top:243-259): keep a stride-1 window iff row%S==0 && col%S==0.
 * Window = kdim*kdim*ch bytes.  Returns the number of windows kept. */
long long sicn_or_decimate(const uint8_t *in, uint8_t *out, int rows, int cols, int stride_y,
                           int stride_x, size_t window_bytes)
{
    long long kept = 0;
    for (int row = 0; row < rows; row++)
        for (int col = 0; col < cols; col++) {
            if ((row % stride_y == 0) & (col % stride_x == 0)) {
                memcpy(out + (size_t)kept * window_bytes,
                       in + ((size_t)row * cols + col) * window_bytes, window_bytes);
                kept++;
            }
        }
    return kept;
}

/* ------------------------------------------------------------------------------------------ */
/* a5: Matrix_Vector_Activate_Batch (mvau.hpp:87-179) with                                      */
/*     TSrcI = Slice<ap_uint<8>> (interpret.hpp:213-217: lane read as UNSIGNED 8 bit),          */
/*     TWeightI = Identity, weights = FixedPointWeights<SIMD, ap_int<4>, PE, TILES>              */
/*       (weights.hpp:134-139: element s = sign-extended nibble in bits [4s,4s+4) of             */
/*        m_weights[pe][tile]),                                                                 */
/*     activation = PassThroughActivation<ap_uint<8>> -> accumulator type ap_uint<8>            */
/*       (mvau.hpp:112, activations.hpp:112-115): EVERY `res +=` of mac.hpp:166-169 wraps       */
/*       mod 2^8.                                                                               */
/*     in : reps windows of matrix_w bytes;  out: reps * NF words of PE bytes, which the        */
/*     following up-converter (streamtools.h:503-523) concatenates first-word-lowest, i.e. the  */
/*     byte of output channel nf*PE+pe lands at offset nf*PE+pe.                                */
/* ------------------------------------------------------------------------------------------ */
int sicn_or_mvau(const uint8_t *in, uint8_t *out, const uint64_t *m_weights /* [PE][TILES] */,
                 int matrix_w, int matrix_h, int simd, int pe_n, int tiles, long long reps)
{
    if (matrix_h % pe_n || matrix_w % simd || simd * 4 > 64) return SICN_OR_EINVAL;
    const int nf_n = matrix_h / pe_n, sf_n = matrix_w / simd;
    if (tiles != nf_n * sf_n) return SICN_OR_EINVAL;
    uint8_t *accu = (uint8_t *)malloc((size_t)pe_n);
    if (!accu) return SICN_OR_ENOMEM;
    for (long long r = 0; r < reps; r++) {
        const uint8_t *vec = in + (size_t)r * matrix_w; /* inputBuf[sf], mvau.hpp:125-134 */
        int tile = 0;
        for (int nf = 0; nf < nf_n; nf++) {
            for (int pe = 0; pe < pe_n; pe++) accu[pe] = 0; /* activation.init, mvau.hpp:137-145 */
            for (int sf = 0; sf < sf_n; sf++, tile++) {
                const uint8_t *in_elem = vec + (size_t)sf * simd;
                for (int pe = 0; pe < pe_n; pe++) {
                    const uint64_t word = m_weights[(size_t)pe * tiles + tile];
                    uint8_t res = accu[pe];
                    for (int s = 0; s < simd; s++) { /* mac<SIMD>, mac.hpp:163-172 */
                        int w = (int)((word >> (4 * s)) & 15u);
                        if (w > 7) w -= 16;
                        res = (uint8_t)(res + w * (int)in_elem[s]);
                    }
                    accu[pe] = res;
                }
            }
            memcpy(out + ((size_t)r * nf_n + nf) * pe_n, accu, (size_t)pe_n); /* mvau.hpp:160-170 */
        }
    }
    free(accu);
    return SICN_OR_OK;
}

/* a6: bias + ReLU (top:267-278, 183-194): lane = lane + bias[j] in 8 bits; if bit 7 set -> 0 */
void sicn_or_bias_relu(uint8_t *io, const int8_t *bias, long long pixels, int ch)
{
    for (long long i = 0; i < pixels; i++)
        for (int j = 0; j < ch; j++) {
            uint8_t v = (uint8_t)(io[(size_t)i * ch + j] + (uint8_t)bias[j]);
            if (v & 0x80u) v = 0;
            io[(size_t)i * ch + j] = v;
        }
}

/* a7: deconv zero-insert (top:110-126) then side pad (top:130-150):
 *     in [H][W][C] -> inner (2H-1)x(2W-1) -> side 2H x 2W */
void sicn_or_zero_insert_side_pad(const uint8_t *in, uint8_t *out, int in_x, int in_y, int ch)
{
    const int ix = 2 * in_x - 1, iy = 2 * in_y - 1;
    uint8_t *inner = (uint8_t *)calloc((size_t)ix * iy, (size_t)ch);
    size_t wr = 0, rd = 0;
    for (int y = 0; y < in_y; y++) { /* inner padding */
        for (int x = 0; x < in_x; x++) {
            memcpy(inner + wr, in + rd, (size_t)ch);
            wr += (size_t)ch;
            rd += (size_t)ch;
            if (x < in_x - 1) wr += (size_t)ch; /* inner_pad.write(0) */
        }
        if (y < in_y - 1) wr += (size_t)((in_x - 1) * 2 + 1) * ch;
    }
    wr = 0;
    rd = 0;
    for (int y = 0; y < iy + 1; y++) /* side padding */
        for (int x = 0; x < ix + 1; x++) {
            if (y >= iy || x >= ix)
                memset(out + wr, 0, (size_t)ch);
            else {
                memcpy(out + wr, inner + rd, (size_t)ch);
                rd += (size_t)ch;
            }
            wr += (size_t)ch;
        }
    free(inner);
}

static int check_desc(const sicn_or_layer_desc *d)
{
    if (d->K != 5 || d->S != 2 || d->P != 2 || d->IN_BIT != 8 || d->OUT_BIT != 8 || d->W_BIT != 4)
        return SICN_OR_EINVAL;
    if (d->IFM_CH <= 0 || d->OFM_CH <= 0 || d->IFM_ROW <= 0 || d->IFM_COL <= 0 || d->SIMD <= 0 ||
        d->PE <= 0)
        return SICN_OR_EINVAL;
    if (d->IFM_CH % d->SIMD || d->OFM_CH % d->PE) return SICN_OR_EINVAL;
    if (d->W_TILES != (d->OFM_CH / d->PE) * (25 * d->IFM_CH / d->SIMD)) return SICN_OR_EINVAL;
    if (d->transposed) {
        if (d->OFM_ROW != 2 * d->IFM_ROW || d->OFM_COL != 2 * d->IFM_COL) return SICN_OR_EINVAL;
    } else if (d->OFM_ROW != (d->IFM_ROW + 1) / 2 || d->OFM_COL != (d->IFM_COL + 1) / 2)
        return SICN_OR_EINVAL;
    return SICN_OR_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a8: conv2d<> as the dataflow runs it (top:198-280): a1 -> a2 -> a3 -> a4 -> a5 -> a2 -> a6.  */
/*     use_fsm != 0 runs the sliding-window FSM itself; 0 runs the plain im2col it implements.  */
/* ------------------------------------------------------------------------------------------ */
int sicn_or_conv2d_dataflow(const sicn_or_layer_desc *d, const uint64_t *m_weights,
                            const int8_t *bias, const uint8_t *in, uint8_t *out, int use_fsm)
{
    int rc = check_desc(d);
    if (rc || d->transposed) return SICN_OR_EINVAL;
    const int C = d->IFM_CH, px = d->IFM_ROW + 2 * d->P, py = d->IFM_COL + 2 * d->P;
    const int sx = px - d->K + 1, sy = py - d->K + 1; /* stride-1 window grid */
    const size_t win = (size_t)d->K * d->K * C;
    uint8_t *in_pad = (uint8_t *)malloc((size_t)px * py * C);
    uint8_t *conv_inp = (uint8_t *)malloc((size_t)sx * sy * win);
    uint8_t *conv_inp2 = (uint8_t *)malloc((size_t)d->OFM_ROW * d->OFM_COL * win);
    if (!in_pad || !conv_inp || !conv_inp2) { rc = SICN_OR_ENOMEM; goto done; }
    sicn_or_fm_padding(in, in_pad, px, py, 2 * d->P, 2 * d->P, C);
    /* StreamingDataWidthConverter C*8 -> SIMD*8 (top:236): identity on bytes */
    if (use_fsm) {
        long long n = sicn_or_swg_nonsquare_fsm(in_pad, (long long)px * py * (C / d->SIMD), conv_inp,
                                                d->K, d->K, C, px, py, sx, sy, d->SIMD, 1, 1);
        if (n < 0) { rc = (int)n; goto done; }
        if (n != (long long)sx * sy * d->K * d->K * (C / d->SIMD)) { rc = SICN_OR_EUNDERRUN; goto done; }
    } else
        sicn_or_im2col_s1(in_pad, conv_inp, d->K, C, px, sx, sy);
    if (sicn_or_decimate(conv_inp, conv_inp2, sy, sx, d->S, d->S, win) !=
        (long long)d->OFM_ROW * d->OFM_COL) { rc = SICN_OR_EINVAL; goto done; }
    rc = sicn_or_mvau(conv_inp2, out, m_weights, (int)win, d->OFM_CH, d->SIMD, d->PE, d->W_TILES,
                      (long long)d->OFM_ROW * d->OFM_COL);
    if (rc) goto done;
    /* StreamingDataWidthConverter PE*8 -> OFM_CH*8 (top:265): identity on bytes */
    sicn_or_bias_relu(out, bias, (long long)d->OFM_ROW * d->OFM_COL, d->OFM_CH);
done:
    free(in_pad);
    free(conv_inp);
    free(conv_inp2);
    return rc;
}

/* a9: deconv522<> as the dataflow runs it (top:71-195): a7 -> a1 -> a2 -> a3(s1) -> a5 -> a2 -> a6 */
int sicn_or_deconv522_dataflow(const sicn_or_layer_desc *d, const uint64_t *m_weights,
                               const int8_t *bias, const uint8_t *in, uint8_t *out, int use_fsm)
{
    int rc = check_desc(d);
    if (rc || !d->transposed) return SICN_OR_EINVAL;
    const int C = d->IFM_CH, ux = 2 * d->IFM_ROW, uy = 2 * d->IFM_COL;
    const int padding = d->K - d->P - 1; /* top:97 */
    const int px = ux + 2 * padding, py = uy + 2 * padding;
    const int ox = px - 5 + 1, oy = py - 5 + 1;
    const size_t win = (size_t)25 * C;
    uint8_t *side = (uint8_t *)malloc((size_t)ux * uy * C);
    uint8_t *all_pad = (uint8_t *)malloc((size_t)px * py * C);
    uint8_t *conv_inp = (uint8_t *)malloc((size_t)ox * oy * win);
    if (!side || !all_pad || !conv_inp) { rc = SICN_OR_ENOMEM; goto done; }
    sicn_or_zero_insert_side_pad(in, side, d->IFM_ROW, d->IFM_COL, C);
    sicn_or_fm_padding(side, all_pad, px, py, 2 * padding, 2 * padding, C);
    if (use_fsm) {
        long long n = sicn_or_swg_nonsquare_fsm(all_pad, (long long)px * py * (C / d->SIMD), conv_inp,
                                                5, 5, C, px, py, ox, oy, d->SIMD, 1, 1);
        if (n < 0) { rc = (int)n; goto done; }
        if (n != (long long)ox * oy * 25 * (C / d->SIMD)) { rc = SICN_OR_EUNDERRUN; goto done; }
    } else
        sicn_or_im2col_s1(all_pad, conv_inp, 5, C, px, ox, oy);
    rc = sicn_or_mvau(conv_inp, out, m_weights, (int)win, d->OFM_CH, d->SIMD, d->PE, d->W_TILES,
                      (long long)ox * oy);
    if (rc) goto done;
    sicn_or_bias_relu(out, bias, (long long)ox * oy, d->OFM_CH);
done:
    free(side);
    free(all_pad);
    free(conv_inp);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Weight unpack exactly as the testbench does it (conv3_nonsquare_tb.cpp:546-571):             */
/* W[o][kx][ky][c], with the testbench's counters (chan innermost, then kx, then ky, then        */
/* o += PE).                                                                                     */
/* ------------------------------------------------------------------------------------------ */
void sicn_or_tb_unpack_weights(const uint64_t *m_weights, int simd, int pe_n, int tiles, int cin,
                               int cout, int8_t *W /* [cout][5][5][cin] indexed [o][kx][ky][c] */)
{
    const int tx = cin * 25 / simd, ty = cout / pe_n;
    int kx = 0, ky = 0, chan = 0;
    for (int pe = 0; pe < pe_n; pe++) {
        int o = pe;
        for (int oy = 0; oy < ty; oy++)
            for (int ox = 0; ox < tx; ox++)
                for (int s = 0; s < simd; s++) {
                    int w = (int)((m_weights[(size_t)pe * tiles + oy * tx + ox] >> (4 * s)) & 15u);
                    if (w > 7) w -= 16;
                    W[(((size_t)o * 5 + kx) * 5 + ky) * cin + chan] = (int8_t)w;
                    if (++chan == cin) {
                        chan = 0;
                        if (++kx == 5) {
                            kx = 0;
                            if (++ky == 5) {
                                ky = 0;
                                o += pe_n;
                                if (o == cout) o = 0;
                            }
                        }
                    }
                }
    }
}

/* The reference's golden model: conv_nonsquare (conv.hpp:91-123) over a padded image indexed
 * [x][y][c] with TO = ap_int<8> accumulator (wraps mod 2^8 on every +=), then + BIAS in ap_int<8>
 * and "< 0 -> 0" (conv3_nonsquare_tb.cpp:616-627).  `padded` is [H][W][C] here; the reference's
 * [x][y] indexing of the same pixels is only a transposition of array storage. */
static void naive_conv_bias_relu(const uint8_t *padded, int pad_x, const int8_t *W, const int8_t *bias,
                                 int cin, int cout, int ofm_x, int ofm_y, int stride, uint8_t *out)
{
    for (int y = 0; y < ofm_y; y++)
        for (int x = 0; x < ofm_x; x++)
            for (int h = 0; h < cout; h++) {
                int8_t tmp = 0;
                for (int ky = 0; ky < 5; ky++)
                    for (int kx = 0; kx < 5; kx++)
                        for (int w = 0; w < cin; w++) {
                            int img = padded[((size_t)(y * stride + ky) * pad_x + (x * stride + kx)) * cin + w];
                            int wt = W[(((size_t)h * 5 + kx) * 5 + ky) * cin + w];
                            tmp = (int8_t)(tmp + img * wt);
                        }
                tmp = (int8_t)(tmp + bias[h]);
                if (tmp < 0) tmp = 0;
                out[((size_t)y * ofm_x + x) * cout + h] = (uint8_t)tmp;
            }
}

/* verify_conv2d (conv3_nonsquare_tb.cpp:513-629) */
int sicn_or_naive_conv2d(const sicn_or_layer_desc *d, const uint64_t *m_weights, const int8_t *bias,
                         const uint8_t *in, uint8_t *out)
{
    int rc = check_desc(d);
    if (rc || d->transposed) return SICN_OR_EINVAL;
    const int C = d->IFM_CH, px = d->IFM_ROW + 2 * d->P, py = d->IFM_COL + 2 * d->P;
    int8_t *W = (int8_t *)malloc((size_t)d->OFM_CH * 25 * C);
    uint8_t *pad = (uint8_t *)calloc((size_t)px * py, (size_t)C);
    if (!W || !pad) { free(W); free(pad); return SICN_OR_ENOMEM; }
    sicn_or_tb_unpack_weights(m_weights, d->SIMD, d->PE, d->W_TILES, C, d->OFM_CH, W);
    for (int y = 0; y < d->IFM_COL; y++) /* tb:581-598 */
        memcpy(pad + ((size_t)(y + d->P) * px + d->P) * C, in + (size_t)y * d->IFM_ROW * C,
               (size_t)d->IFM_ROW * C);
    naive_conv_bias_relu(pad, px, W, bias, C, d->OFM_CH, d->OFM_ROW, d->OFM_COL, d->S, out);
    free(W);
    free(pad);
    return SICN_OR_OK;
}

/* verify_deconv2d (conv3_nonsquare_tb.cpp:632-748): padded map [2W+4][2H+4], non-zero only where
 * (ox+1-P)%2 != 0 && (oy+1-P)%2 != 0 inside the border, value in[(ox-P)/2][(oy-P)/2]; stride-1 conv */
int sicn_or_naive_deconv2d(const sicn_or_layer_desc *d, const uint64_t *m_weights,
                           const int8_t *bias, const uint8_t *in, uint8_t *out)
{
    int rc = check_desc(d);
    if (rc || !d->transposed) return SICN_OR_EINVAL;
    const int C = d->IFM_CH, P = d->P, px = 2 * d->IFM_ROW + 2 * P, py = 2 * d->IFM_COL + 2 * P;
    int8_t *W = (int8_t *)malloc((size_t)d->OFM_CH * 25 * C);
    uint8_t *pad = (uint8_t *)calloc((size_t)px * py, (size_t)C);
    if (!W || !pad) { free(W); free(pad); return SICN_OR_ENOMEM; }
    sicn_or_tb_unpack_weights(m_weights, d->SIMD, d->PE, d->W_TILES, C, d->OFM_CH, W);
    for (int oy = 0; oy < py; oy++)
        for (int ox = 0; ox < px; ox++) {
            int zero = ((ox < P) | (ox >= d->IFM_ROW * 2 + P)) | ((oy < P) | (oy >= d->IFM_COL * 2 + P)) |
                       ((ox + 1 - P) % 2 == 0) | ((oy + 1 - P) % 2 == 0);
            if (!zero)
                memcpy(pad + ((size_t)oy * px + ox) * C,
                       in + ((size_t)((oy - P) / 2) * d->IFM_ROW + (ox - P) / 2) * C, (size_t)C);
        }
    naive_conv_bias_relu(pad, px, W, bias, C, d->OFM_CH, d->OFM_ROW, d->OFM_COL, 1, out);
    free(W);
    free(pad);
    return SICN_OR_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Closed forms of SURVEY.md §8(a) a8/a9, wide accumulate then truncate (the ring homomorphism   */
/* Z -> Z/256).  Fast enough for 1080p-size checks; `threads` > 1 uses OpenMP when built with it. */
/* ------------------------------------------------------------------------------------------ */
static void unpack_okc(const uint64_t *m_weights, const sicn_or_layer_desc *d, int8_t *W /* [o][k] */)
{
    const int kk = 25 * d->IFM_CH, sf_n = kk / d->SIMD, nf_n = d->OFM_CH / d->PE;
    for (int pe = 0; pe < d->PE; pe++)
        for (int nf = 0; nf < nf_n; nf++)
            for (int sf = 0; sf < sf_n; sf++) {
                uint64_t word = m_weights[(size_t)pe * d->W_TILES + nf * sf_n + sf];
                for (int s = 0; s < d->SIMD; s++) {
                    int w = (int)((word >> (4 * s)) & 15u);
                    if (w > 7) w -= 16;
                    W[(size_t)(nf * d->PE + pe) * kk + sf * d->SIMD + s] = (int8_t)w;
                }
            }
}

/* sum_c s[c]*w[c]; cloned for AVX2 with run-time dispatch (the .so is built on one host and timed on
 * another, so no -march=native) */
__attribute__((target_clones("avx2", "default"))) static int32_t dot_u8_i8(const uint8_t *s, const int8_t *w, int n)
{
    int32_t a = 0;
#pragma omp simd reduction(+ : a)
    for (int c = 0; c < n; c++) a += (int32_t)s[c] * (int32_t)w[c];
    return a;
}

/* relu = 1: the reference layer.  relu = 0: the lane BEFORE the sign-bit ReLU (conv_nonsquare_top.cpp:272 without
 * :273-275) — the input of the GDN / IGDN extension (oracle/sicn_gdn_oracle.c), which replaces the ReLU. */
int sicn_or_layer_direct_act(const sicn_or_layer_desc *d, const uint64_t *m_weights, const int8_t *bias,
                             const uint8_t *in, uint8_t *out, int threads, int relu);

int sicn_or_layer_direct(const sicn_or_layer_desc *d, const uint64_t *m_weights, const int8_t *bias,
                         const uint8_t *in, uint8_t *out, int threads)
{
    return sicn_or_layer_direct_act(d, m_weights, bias, in, out, threads, 1);
}

int sicn_or_layer_direct_act(const sicn_or_layer_desc *d, const uint64_t *m_weights, const int8_t *bias,
                             const uint8_t *in, uint8_t *out, int threads, int relu)
{
    int rc = check_desc(d);
    if (rc) return rc;
    const int C = d->IFM_CH, N = d->OFM_CH, IW = d->IFM_ROW, IH = d->IFM_COL;
    const int OW = d->OFM_ROW, OH = d->OFM_COL, tr = d->transposed;
    int8_t *W = (int8_t *)malloc((size_t)N * 25 * C);
    if (!W) return SICN_OR_ENOMEM;
    unpack_okc(m_weights, d, W);
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (int y = 0; y < OH; y++)
        for (int x = 0; x < OW; x++) {
            const uint8_t *src[25];
            int wk[25], nt = 0;
            for (int ky = 0; ky < 5; ky++)
                for (int kx = 0; kx < 5; kx++) {
                    int iy, ix;
                    if (tr) { /* Up_pad[y+ky][x+kx] = in[(y+ky-2)/2][(x+kx-2)/2] iff both even */
                        int py = y + ky - 2, px = x + kx - 2;
                        if ((py & 1) || (px & 1)) continue;
                        iy = py >> 1;
                        ix = px >> 1;
                    } else {
                        iy = 2 * y + ky - 2;
                        ix = 2 * x + kx - 2;
                    }
                    if (iy < 0 || iy >= IH || ix < 0 || ix >= IW) continue;
                    src[nt] = in + ((size_t)iy * IW + ix) * C;
                    wk[nt++] = (ky * 5 + kx) * C;
                }
            for (int o = 0; o < N; o++) {
                int32_t acc = 0;
                const int8_t *wo = W + (size_t)o * 25 * C;
                for (int t = 0; t < nt; t++) acc += dot_u8_i8(src[t], wo + wk[t], C);
                uint8_t v = (uint8_t)((acc + bias[o]) & 0xFF);
                if (relu && (v & 0x80u)) v = 0;
                out[((size_t)y * OW + x) * N + o] = v;
            }
        }
    free(W);
    return SICN_OR_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* ConvLayer_Batch (convlayer.h:89-125) — the generic finn-hlslib layer, restated as the         */
/* dataflow runs it: square ConvolutionInputGenerator (slidingwindow.h:163-270; same FSM as the  */
/* non-square one with IFMDim_x = IFMDim_y, stride 1, NO padding) -> Matrix_Vector_Activate_Batch */
/* with TA = ap_int/ap_uint<ACC_BIT> wrapping at every += (mvau.hpp:112,149-156; mac.hpp:163-172) */
/* and PassThroughActivation (activations.hpp:127-134) or ThresholdsActivation                   */
/* (activations.hpp:168-190: result = ActVal + #{i : m_thresholds[pe][nf][i] < accu}).           */
/* The reference never executes this surface (conv_nonsquare_top.cpp:223 is commented out):      */
/* parity UNPINNED.  Field order = include/sicn_convlayer.h sicn_convlayer_desc.                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t K, IFM_CH, IFM_DIM, OFM_CH, OFM_DIM, SIMD, PE, IN_BIT, IN_SIGNED, W_BIT, W_TILES;
    int32_t ACC_BIT, ACC_SIGNED, OUT_BIT, activation, NUM_TH, ACT_VAL;
} sicn_or_convlayer_desc;

static int64_t wrap_bits(int64_t v, int bits, int is_signed)
{
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    uint64_t u = (uint64_t)v & mask;
    if (is_signed && bits < 64 && ((u >> (bits - 1)) & 1)) return (int64_t)(u | ~mask);
    return (int64_t)u;
}

/* in: one byte per lane (low IN_BIT bits = the lane); out: uint32 per lane = the low OUT_BIT bits of the activation result */
int sicn_or_convlayer_dataflow(const sicn_or_convlayer_desc *d, const uint64_t *m_weights /* [PE][TILES] */,
                               const int32_t *thresholds /* [PE][NF][NUM_TH] or NULL */, const uint8_t *in,
                               uint32_t *out, int use_fsm)
{
    if (d->IFM_CH % d->SIMD || d->OFM_CH % d->PE || d->OFM_DIM != d->IFM_DIM - d->K + 1 || d->IN_BIT < 1 || d->IN_BIT > 8)
        return SICN_OR_EINVAL;
    const int C = d->IFM_CH, K = d->K, kk = K * K * C, sf_n = kk / d->SIMD, nf_n = d->OFM_CH / d->PE;
    if (d->W_TILES != nf_n * sf_n || d->SIMD * d->W_BIT > 64) return SICN_OR_EINVAL;
    const long long windows = (long long)d->OFM_DIM * d->OFM_DIM;
    uint8_t *conv_inp = (uint8_t *)malloc((size_t)windows * kk);
    int64_t *accu = (int64_t *)malloc(sizeof(int64_t) * (size_t)d->PE);
    if (!conv_inp || !accu) { free(conv_inp); free(accu); return SICN_OR_ENOMEM; }
    int rc = SICN_OR_OK;
    if (use_fsm) {
        long long n = sicn_or_swg_nonsquare_fsm(in, (long long)d->IFM_DIM * d->IFM_DIM * (C / d->SIMD), conv_inp, K, K, C,
                                                d->IFM_DIM, d->IFM_DIM, d->OFM_DIM, d->OFM_DIM, d->SIMD, 1, 1);
        if (n != windows * K * K * (C / d->SIMD)) rc = n < 0 ? (int)n : SICN_OR_EUNDERRUN;
    } else
        sicn_or_im2col_s1(in, conv_inp, K, C, d->IFM_DIM, d->OFM_DIM, d->OFM_DIM);
    const uint64_t wmask = (1ull << d->W_BIT) - 1;
    for (long long r = 0; r < windows && !rc; r++) {
        const uint8_t *vec = conv_inp + (size_t)r * kk;
        int tile = 0;
        for (int nf = 0; nf < nf_n; nf++) {
            for (int pe = 0; pe < d->PE; pe++) accu[pe] = 0;
            for (int sf = 0; sf < sf_n; sf++, tile++)
                for (int pe = 0; pe < d->PE; pe++) {
                    const uint64_t word = m_weights[(size_t)pe * d->W_TILES + tile];
                    int64_t res = accu[pe];
                    for (int s = 0; s < d->SIMD; s++) {
                        int w = (int)((word >> (d->W_BIT * s)) & wmask);
                        if (w >> (d->W_BIT - 1)) w -= 1 << d->W_BIT;
                        /* one BYTE per lane here (the sliding-window stage moves lanes, not bits); the lane is TSrcI = Slice<ap_(u)int<IN_BIT>>:
                         * its low IN_BIT bits, sign-extended for ap_int (interpret.hpp:191-244).  The packing of lanes into stream words
                         * is restated separately (oracle/sicn_ref.py: pack_stream_lanes / unpack_stream_lanes). */
                        int x = (int)vec[sf * d->SIMD + s] & ((1 << d->IN_BIT) - 1);
                        if (d->IN_SIGNED && (x >> (d->IN_BIT - 1))) x -= 1 << d->IN_BIT;
                        res = wrap_bits(res + (int64_t)w * x, d->ACC_BIT, d->ACC_SIGNED); /* T res += ...; T = TA */
                    }
                    accu[pe] = res;
                }
            for (int pe = 0; pe < d->PE; pe++) {
                int64_t result = accu[pe];
                if (d->activation == 1) {
                    result = d->ACT_VAL;
                    for (int i = 0; i < d->NUM_TH; i++) {
                        const int64_t th = wrap_bits(thresholds[((size_t)pe * nf_n + nf) * d->NUM_TH + i], d->ACC_BIT, d->ACC_SIGNED);
                        result += (th < accu[pe]) ? 1 : 0;
                    }
                }
                const uint64_t omask = d->OUT_BIT >= 32 ? 0xFFFFFFFFull : ((1ull << d->OUT_BIT) - 1);
                out[(size_t)r * d->OFM_CH + nf * d->PE + pe] = (uint32_t)((uint64_t)result & omask);
            }
        }
    }
    free(conv_inp);
    free(accu);
    return rc;
}
