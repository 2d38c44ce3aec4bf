// Generic finn-hlslib ConvLayer_Batch surface (include/sicn_convlayer.h): any kernel size, channel
// count, fold, accumulator width, pass-through or multi-threshold activation.  One thread per output
// lane, direct evaluation:
//   acc  = wrap_TA( sum_{ky,kx,c} x[y+ky][x+kx][c] * W[o][(ky*K+kx)*C + c] )      mvau.hpp:87-179
//   out  = low OUT_BIT bits of  activation(acc)                                    activations.hpp:127-190
// Stride 1, no padding, square image (convlayer.h:116-118).  Parity unpinned (no reference outputs
// exist for this surface, sicn_convlayer.h).
//
// Two kernels.  k_convlayer_mfma (IFM_CH a multiple of 16): implicit GEMM on v_mfma_i32_16x16x64_i8 —
// a wave owns 64 consecutive output positions x 64 output channels (16 accumulator tiles); a K step is
// 64 channel bytes of ONE kernel tap, read straight from the NHWC image (16 bytes per lane, the cache
// hierarchy serves the K*K-fold reuse), weights from a zero-padded [O/16][tap][C/64][16][64] image.
// int8 x int8 is signed x signed: unsigned inputs are read as x - 128 (one v_xor per dword) and
// 128 * sum_k W[o][k] is the accumulators' start value.  k_convlayer: one thread per output lane, direct
// evaluation, for every other shape.  Neither is a tuned hot path (that is sicn.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/sicn.h"
#include "../../include/sicn_convlayer.h"

struct sicn_convlayer_params {
    sicn_convlayer_desc d;
    int8_t *d_w_okc;     // [OFM_CH][K*K*IFM_CH] sign-extended weights
    int32_t *d_thr;      // [OFM_CH][NUM_TH] thresholds in channel order, or nullptr
    int8_t *d_w_mfma;    // [ceil(O/16)][K*K][ceil(C/64)][16 rows][64 bytes], zero padded; nullptr if C % 16 != 0
    int32_t *d_wsum;     // [ceil(O/16)*16] sum_k W[o][k]
};

namespace {

__device__ __forceinline__ long long wrap_acc(long long v, int bits, int is_signed)
{
    const unsigned long long mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    unsigned long long u = (unsigned long long)v & mask;
    if (is_signed && bits < 64 && (u >> (bits - 1)) & 1) return (long long)(u | ~mask);
    return (long long)u;
}

// lane c of a pixel's stream word (interpret.hpp:191-244: Slice<ap_(u)int<W>> reads bits [c W, (c + 1) W)), W in {1, 2, 4, 8}: never
// straddles a byte
__device__ __forceinline__ int lane_value(const uint8_t *pixel, int c, int bits, int is_signed)
{
    const int off = c * bits;
    int v = (pixel[off >> 3] >> (off & 7)) & ((1 << bits) - 1);
    if (is_signed && (v >> (bits - 1))) v -= 1 << bits;
    return v;
}

// one thread per output UNIT: a byte holding 8 / OUT_BIT lanes (OUT_BIT < 8) or one lane's container (OUT_BIT >= 8)
__global__ __launch_bounds__(256) void k_convlayer(const uint8_t *__restrict__ in, void *__restrict__ out,
                                                   const int8_t *__restrict__ w_okc, const int32_t *__restrict__ thr,
                                                   sicn_convlayer_desc d)
{
    const int lanes_per_unit = d.OUT_BIT < 8 ? 8 / d.OUT_BIT : 1;
    const int units_per_pixel = d.OFM_CH / lanes_per_unit;
    const size_t per_img = (size_t)d.OFM_DIM * d.OFM_DIM * units_per_pixel;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_img) return;
    const int img = blockIdx.y;
    const int unit = (int)(idx % units_per_pixel);
    const size_t pix = idx / units_per_pixel;
    const int x = (int)(pix % d.OFM_DIM), y = (int)(pix / d.OFM_DIM);
    const int C = d.IFM_CH, K = d.K;
    const int in_pixel_bytes = C * d.IN_BIT / 8;
    const uint8_t *im = in + (size_t)img * d.IFM_DIM * d.IFM_DIM * in_pixel_bytes;
    uint32_t packed = 0;
    for (int l = 0; l < lanes_per_unit; l++) {
        const int o = unit * lanes_per_unit + l;
        const int8_t *wo = w_okc + (size_t)o * K * K * C;
        long long acc = 0;   // exact: |sum| <= 121 * C * 255 * 128 fits easily
        for (int ky = 0; ky < K; ky++)
            for (int kx = 0; kx < K; kx++) {
                const uint8_t *s = im + ((size_t)(y + ky) * d.IFM_DIM + (x + kx)) * in_pixel_bytes;
                const int8_t *wk = wo + (ky * K + kx) * C;
                if (d.IN_BIT == 8) {
                    if (d.IN_SIGNED)
                        for (int c = 0; c < C; c++) acc += (int)(int8_t)s[c] * (int)wk[c];
                    else
                        for (int c = 0; c < C; c++) acc += (int)s[c] * (int)wk[c];
                } else
                    for (int c = 0; c < C; c++) acc += lane_value(s, c, d.IN_BIT, d.IN_SIGNED) * (int)wk[c];
            }
        const long long a = wrap_acc(acc, d.ACC_BIT, d.ACC_SIGNED);   // TA: every += wraps, the final wrap is the same
        long long r = a;
        if (d.activation == SICN_ACT_THRESHOLDS) {
            r = d.ACT_VAL;
            const int32_t *t = thr + (size_t)o * d.NUM_TH;
            for (int i = 0; i < d.NUM_TH; i++) r += (wrap_acc((long long)t[i], d.ACC_BIT, d.ACC_SIGNED) < a) ? 1 : 0;
        }
        if (d.OUT_BIT < 8)
            packed |= ((uint32_t)r & ((1u << d.OUT_BIT) - 1)) << (l * d.OUT_BIT);   // lane o in bits [o OUT_BIT, (o + 1) OUT_BIT) of the word
        else
            packed = (uint32_t)r;
    }
    const size_t oi = (size_t)img * per_img + idx;
    if (d.OUT_BIT <= 8)
        ((uint8_t *)out)[oi] = (uint8_t)packed;
    else if (d.OUT_BIT == 16)
        ((uint16_t *)out)[oi] = (uint16_t)packed;
    else
        ((uint32_t *)out)[oi] = packed;
}

typedef int v4i_t __attribute__((ext_vector_type(4)));

// wave = 64 consecutive output positions (4 column tiles of 16) x 64 output channels (4 weight tiles)
__global__ __launch_bounds__(256) void k_convlayer_mfma(const uint8_t *__restrict__ in, void *__restrict__ out,
                                                        const int8_t *__restrict__ wm, const int32_t *__restrict__ wsum,
                                                        const int32_t *__restrict__ thr, sicn_convlayer_desc d)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int C = d.IFM_CH, K = d.K, D = d.IFM_DIM, OD = d.OFM_DIM, O = d.OFM_CH;
    const int npos = OD * OD, nchunk = (C + 63) / 64, KK = K * K;
    const int p0 = (blockIdx.x * 4 + wv) * 64;            // first position of this wave
    const int j0 = blockIdx.y * 4;                        // first 16-channel weight tile
    const int ntile = (O + 15) / 16;
    const uint8_t *img = in + (size_t)blockIdx.z * D * D * C;
    if (p0 >= npos) return;

    // this lane's pixel (per column tile): top-left corner of its window, as a byte offset
    uint32_t pix[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int p = p0 + 16 * c + col;
        p = p < npos ? p : npos - 1;                      // clamped positions are computed and not stored
        const int y = p / OD, x = p - y * OD;
        pix[c] = (uint32_t)((y * D + x) * C);
    }
    v4i_t acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v4i_t b;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int o = (j0 + j) * 16 + 4 * g + r;       // C/D row 4g + r of tile j
            b[r] = (d.IN_SIGNED || j0 + j >= ntile) ? 0 : 128 * wsum[o];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c][j] = b;
    }
    const uint32_t flip = d.IN_SIGNED ? 0u : 0x80808080u;
    // K walk: step s = (tap t, 64-channel chunk cc), t outer.  The operands of step s + 1 are requested before the 16 MFMAs of step s
    // (round 5: the loop used to load, wait and multiply — one exposed memory round trip per step; two register sets, the loop unrolled by two)
    const int nstep = KK * nchunk;
    int t_n = 0, cc_n = 0, ky_n = 0, kx_n = 0;            // the step the next request is for
    auto request = [&](v4i_t (&bf)[4], v4i_t (&af)[4]) {
        const uint32_t tap = (uint32_t)((ky_n * D + kx_n) * C);
        const int c0 = cc_n * 64 + 16 * g;                // this lane's 16 channel bytes of the K step
        const bool valid = c0 < C;                         // C % 16 == 0: a 16-byte group is all in or all out (weights 0)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint4 q = *reinterpret_cast<const uint4 *>(img + pix[c] + tap + (valid ? c0 : 0));
            bf[c] = v4i_t{(int)(q.x ^ flip), (int)(q.y ^ flip), (int)(q.z ^ flip), (int)(q.w ^ flip)};
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int jj = j0 + j < ntile ? j0 + j : ntile - 1;
            af[j] = *reinterpret_cast<const v4i_t *>(wm + ((((size_t)jj * KK + t_n) * nchunk + cc_n) * 16 + col) * 64 + 16 * g);
        }
        if (++cc_n == nchunk) {
            cc_n = 0;
            t_n++;
            if (++kx_n == K) {
                kx_n = 0;
                ky_n++;
            }
        }
    };
    auto multiply = [&](const v4i_t (&bf)[4], const v4i_t (&af)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[j], bf[c], acc[c][j], 0, 0, 0);
    };
    v4i_t bf0[4], af0[4], bf1[4], af1[4];
    request(bf0, af0);
    for (int s_ = 0; s_ < nstep; s_ += 2) {
        if (s_ + 1 < nstep) request(bf1, af1);
        multiply(bf0, af0);
        if (s_ + 1 < nstep) {
            if (s_ + 2 < nstep) request(bf0, af0);
            multiply(bf1, af1);
        }
    }
    // epilogue: lane holds, per (column tile c, weight tile j), channels 16(j0+j) + 4g .. +3 of position p0 + 16c + col
    const size_t per_img = (size_t)npos * O;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int p = p0 + 16 * c + col;
        if (p >= npos) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (j0 + j) * 16 + 4 * g + r;
                if (o >= O) continue;
                const long long a = wrap_acc((long long)acc[c][j][r], d.ACC_BIT, d.ACC_SIGNED);
                long long res = a;
                if (d.activation == SICN_ACT_THRESHOLDS) {
                    res = d.ACT_VAL;
                    const int32_t *tt = thr + (size_t)o * d.NUM_TH;
                    for (int i = 0; i < d.NUM_TH; i++) res += (wrap_acc((long long)tt[i], d.ACC_BIT, d.ACC_SIGNED) < a) ? 1 : 0;
                }
                const size_t oi = (size_t)blockIdx.z * per_img + (size_t)p * O + o;
                if (d.OUT_BIT == 8)
                    ((uint8_t *)out)[oi] = (uint8_t)res;
                else if (d.OUT_BIT == 16)
                    ((uint16_t *)out)[oi] = (uint16_t)res;
                else
                    ((uint32_t *)out)[oi] = (uint32_t)res;
            }
        }
    }
}

}  // namespace

extern "C" int sicn_convlayer_validate(const sicn_convlayer_desc *d)
{
    if (!d) return SICN_EINVAL;
    if (d->K < 1 || d->K > 11 || d->IFM_CH <= 0 || d->OFM_CH <= 0 || d->SIMD <= 0 || d->PE <= 0) return SICN_EINVAL;
    if (d->IFM_DIM < d->K || d->IFM_DIM > (1 << 15) || d->OFM_DIM != d->IFM_DIM - d->K + 1) return SICN_EINVAL;
    if (d->IFM_CH % d->SIMD || d->OFM_CH % d->PE) return SICN_EINVAL;   // slidingwindow.h:177, mvau.hpp:101-105
    if ((d->IN_BIT != 1 && d->IN_BIT != 2 && d->IN_BIT != 4 && d->IN_BIT != 8) || (d->IN_SIGNED != 0 && d->IN_SIGNED != 1)) return SICN_EINVAL;
    if ((d->IFM_CH * d->IN_BIT) % 8) return SICN_EINVAL;   // a pixel's stream word is whole bytes
    if (d->W_BIT < 2 || d->W_BIT > 8 || d->SIMD * d->W_BIT > 64) return SICN_EINVAL;
    if ((long long)d->W_TILES != (long long)(d->OFM_CH / d->PE) * ((long long)d->K * d->K * d->IFM_CH / d->SIMD)) return SICN_EINVAL;
    if (d->ACC_BIT < 1 || d->ACC_BIT > 32 || (d->ACC_SIGNED != 0 && d->ACC_SIGNED != 1)) return SICN_EINVAL;
    if (d->OUT_BIT != 2 && d->OUT_BIT != 4 && d->OUT_BIT != 8 && d->OUT_BIT != 16 && d->OUT_BIT != 32) return SICN_EINVAL;
    if ((d->OFM_CH * d->OUT_BIT) % 8) return SICN_EINVAL;
    if (d->activation == SICN_ACT_PASSTHROUGH) {
        if (d->NUM_TH != 0) return SICN_EINVAL;
    } else if (d->activation == SICN_ACT_THRESHOLDS) {
        if (d->NUM_TH < 1 || d->NUM_TH > 1024) return SICN_EINVAL;
    } else
        return SICN_EINVAL;
    return SICN_OK;
}

extern "C" void sicn_convlayer_params_free(sicn_convlayer_params *p)
{
    if (!p) return;
    if (p->d_w_okc) (void)hipFree(p->d_w_okc);
    if (p->d_thr) (void)hipFree(p->d_thr);
    if (p->d_w_mfma) (void)hipFree(p->d_w_mfma);
    if (p->d_wsum) (void)hipFree(p->d_wsum);
    delete p;
}

extern "C" int sicn_convlayer_params_create(const sicn_convlayer_desc *d, const void *m_weights, int word_bytes,
                                            const int32_t *thresholds, sicn_convlayer_params **out)
{
    if (!out) return SICN_EINVAL;
    *out = nullptr;
    int rc = sicn_convlayer_validate(d);
    if (rc) return rc;
    if (!m_weights || (word_bytes != 1 && word_bytes != 2 && word_bytes != 4 && word_bytes != 8)) return SICN_EINVAL;
    if (d->SIMD * d->W_BIT > word_bytes * 8) return SICN_EINVAL;
    if ((d->activation == SICN_ACT_THRESHOLDS) != (thresholds != nullptr)) return SICN_EINVAL;
    const int kk = d->K * d->K * d->IFM_CH, sf_n = kk / d->SIMD, nf_n = d->OFM_CH / d->PE;
    std::vector<int8_t> w;
    std::vector<int32_t> t;
    try {
        w.resize((size_t)d->OFM_CH * kk);
        if (thresholds) t.resize((size_t)d->OFM_CH * d->NUM_TH);
    } catch (const std::bad_alloc &) { return SICN_ENOMEM; }
    // FixedPointWeights: W[o = nf*PE + pe][k = sf*SIMD + s] = sign-extended element s of m_weights[pe][nf*SF + sf]
    const uint8_t *raw = (const uint8_t *)m_weights;
    const int wb = d->W_BIT;
    for (int pe = 0; pe < d->PE; pe++)
        for (int nf = 0; nf < nf_n; nf++) {
            for (int sf = 0; sf < sf_n; sf++) {
                const size_t idx = (size_t)pe * d->W_TILES + (size_t)nf * sf_n + sf;
                uint64_t word = 0;
                for (int b = 0; b < word_bytes; b++) word |= (uint64_t)raw[idx * word_bytes + b] << (8 * b);
                for (int s = 0; s < d->SIMD; s++) {
                    int v = (int)((word >> (wb * s)) & ((1u << wb) - 1));
                    if (v >> (wb - 1)) v -= 1 << wb;
                    w[(size_t)(nf * d->PE + pe) * kk + sf * d->SIMD + s] = (int8_t)v;
                }
            }
            // ThresholdsActivation::m_thresholds[PE][NF][NumTH] -> [o][i]
            if (thresholds)
                for (int i = 0; i < d->NUM_TH; i++)
                    t[(size_t)(nf * d->PE + pe) * d->NUM_TH + i] = thresholds[((size_t)pe * nf_n + nf) * d->NUM_TH + i];
        }
    sicn_convlayer_params *p = new (std::nothrow) sicn_convlayer_params();
    if (!p) return SICN_ENOMEM;
    p->d = *d;
    p->d_w_okc = nullptr;
    p->d_thr = nullptr;
    p->d_w_mfma = nullptr;
    p->d_wsum = nullptr;
    bool ok = hipMalloc((void **)&p->d_w_okc, w.size()) == hipSuccess &&
              hipMemcpy(p->d_w_okc, w.data(), w.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && d->IFM_CH % 16 == 0 && d->IN_BIT == 8 && d->OUT_BIT >= 8) {   // the MFMA image (byte lanes only): [O/16][tap][C/64][16 rows][64 bytes], zero padded
        const int KK = d->K * d->K, nchunk = (d->IFM_CH + 63) / 64, ntile = (d->OFM_CH + 15) / 16;
        std::vector<int8_t> wm;
        std::vector<int32_t> ws;
        try {
            wm.assign((size_t)ntile * KK * nchunk * 16 * 64, 0);
            ws.assign((size_t)ntile * 16, 0);
        } catch (const std::bad_alloc &) {
            sicn_convlayer_params_free(p);
            return SICN_ENOMEM;
        }
        for (int o = 0; o < d->OFM_CH; o++)
            for (int t = 0; t < KK; t++)
                for (int c = 0; c < d->IFM_CH; c++) {
                    const int8_t v = w[(size_t)o * kk + t * d->IFM_CH + c];
                    wm[((((size_t)(o / 16) * KK + t) * nchunk + c / 64) * 16 + o % 16) * 64 + c % 64] = v;
                    ws[o] += v;
                }
        ok = hipMalloc((void **)&p->d_w_mfma, wm.size()) == hipSuccess &&
             hipMemcpy(p->d_w_mfma, wm.data(), wm.size(), hipMemcpyHostToDevice) == hipSuccess &&
             hipMalloc((void **)&p->d_wsum, ws.size() * 4) == hipSuccess &&
             hipMemcpy(p->d_wsum, ws.data(), ws.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok && thresholds)
        ok = hipMalloc((void **)&p->d_thr, t.size() * 4) == hipSuccess &&
             hipMemcpy(p->d_thr, t.data(), t.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        sicn_convlayer_params_free(p);
        return SICN_ENOMEM;
    }
    *out = p;
    return SICN_OK;
}

extern "C" int sicn_conv_layer_batch(const sicn_convlayer_desc *d, const sicn_convlayer_params *p, const uint8_t *in,
                                     void *out, int reps, void *hip_stream)
{
    return sicn_conv_layer_batch_kernel(d, p, in, out, reps, SICN_CONVLAYER_KERNEL_AUTO, hip_stream);
}

extern "C" int sicn_conv_layer_batch_kernel(const sicn_convlayer_desc *d, const sicn_convlayer_params *p, const uint8_t *in,
                                            void *out, int reps, int kernel, void *hip_stream)
{
    if (kernel != SICN_CONVLAYER_KERNEL_AUTO && kernel != SICN_CONVLAYER_KERNEL_DIRECT) return SICN_EINVAL;
    int rc = sicn_convlayer_validate(d);
    if (rc) return rc;
    if (!p || !in || !out || reps < 0 || reps > 65535) return SICN_EINVAL;
    const sicn_convlayer_desc &q = p->d;   // the parameters must have been made for this layer
    if (q.K != d->K || q.IFM_CH != d->IFM_CH || q.OFM_CH != d->OFM_CH || q.W_BIT != d->W_BIT ||
        q.activation != d->activation || q.NUM_TH != d->NUM_TH)
        return SICN_EINVAL;
    if (reps == 0) return SICN_OK;
    if (q.IN_BIT != d->IN_BIT || q.IN_SIGNED != d->IN_SIGNED) return SICN_EINVAL;
    const size_t per_img = (size_t)d->OFM_DIM * d->OFM_DIM * (d->OUT_BIT < 8 ? d->OFM_CH * d->OUT_BIT / 8 : d->OFM_CH);   // output units
    const size_t blocks = (per_img + 255) / 256;
    if (blocks > 0x7fffffffu) return SICN_EINVAL;
    if (p->d_w_mfma && d->IN_BIT == 8 && d->OUT_BIT >= 8 && (size_t)d->IFM_DIM * d->IFM_DIM * d->IFM_CH < 0x7fffffffu && kernel != SICN_CONVLAYER_KERNEL_DIRECT) {
        const unsigned npos = (unsigned)(d->OFM_DIM * d->OFM_DIM);
        dim3 grid((npos + 255) / 256, (unsigned)((d->OFM_CH + 63) / 64), (unsigned)reps);
        hipLaunchKernelGGL(k_convlayer_mfma, grid, dim3(256), 0, (hipStream_t)hip_stream, in, out, p->d_w_mfma, p->d_wsum,
                           p->d_thr, *d);
        return hipGetLastError() == hipSuccess ? SICN_OK : SICN_ENODEV;
    }
    hipLaunchKernelGGL(k_convlayer, dim3((unsigned)blocks, (unsigned)reps), dim3(256), 0, (hipStream_t)hip_stream, in, out,
                       p->d_w_okc, p->d_thr, *d);
    return hipGetLastError() == hipSuccess ? SICN_OK : SICN_ENODEV;
}
