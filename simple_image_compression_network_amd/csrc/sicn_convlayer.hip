// Generic finn-hlslib ConvLayer_Batch surface (include/sicn_convlayer.h): any kernel size, channel
// count, fold, accumulator width, pass-through or multi-threshold activation.  One thread per output
// lane, direct evaluation:
//   acc  = wrap_TA( sum_{ky,kx,c} x[y+ky][x+kx][c] * W[o][(ky*K+kx)*C + c] )      mvau.hpp:87-179
//   out  = low OUT_BIT bits of  activation(acc)                                    activations.hpp:127-190
// Stride 1, no padding, square image (convlayer.h:116-118).  Parity unpinned (no reference outputs
// exist for this surface, sicn_convlayer.h); functional surface, not a tuned hot path.
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "../../include/sicn.h"
#include "../../include/sicn_convlayer.h"

struct sicn_convlayer_params {
    sicn_convlayer_desc d;
    int8_t *d_w_okc;     // [OFM_CH][K*K*IFM_CH] sign-extended weights
    int32_t *d_thr;      // [OFM_CH][NUM_TH] thresholds in channel order, or nullptr
};

namespace {

__device__ __forceinline__ long long wrap_acc(long long v, int bits, int is_signed)
{
    const unsigned long long mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    unsigned long long u = (unsigned long long)v & mask;
    if (is_signed && bits < 64 && (u >> (bits - 1)) & 1) return (long long)(u | ~mask);
    return (long long)u;
}

__global__ __launch_bounds__(256) void k_convlayer(const uint8_t *__restrict__ in, void *__restrict__ out,
                                                   const int8_t *__restrict__ w_okc, const int32_t *__restrict__ thr,
                                                   sicn_convlayer_desc d)
{
    const size_t per_img = (size_t)d.OFM_DIM * d.OFM_DIM * d.OFM_CH;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_img) return;
    const int img = blockIdx.y;
    const int o = (int)(idx % d.OFM_CH);
    const size_t pix = idx / d.OFM_CH;
    const int x = (int)(pix % d.OFM_DIM), y = (int)(pix / d.OFM_DIM);
    const int C = d.IFM_CH, K = d.K;
    const uint8_t *im = in + (size_t)img * d.IFM_DIM * d.IFM_DIM * C;
    const int8_t *wo = w_okc + (size_t)o * K * K * C;
    long long acc = 0;   // exact: |sum| <= 121 * C * 255 * 128 fits easily
    for (int ky = 0; ky < K; ky++)
        for (int kx = 0; kx < K; kx++) {
            const uint8_t *s = im + ((size_t)(y + ky) * d.IFM_DIM + (x + kx)) * C;
            const int8_t *wk = wo + (ky * K + kx) * C;
            if (d.IN_SIGNED)
                for (int c = 0; c < C; c++) acc += (int)(int8_t)s[c] * (int)wk[c];
            else
                for (int c = 0; c < C; c++) acc += (int)s[c] * (int)wk[c];
        }
    long long a = wrap_acc(acc, d.ACC_BIT, d.ACC_SIGNED);   // TA: every += wraps, the final wrap is the same
    long long r = a;
    if (d.activation == SICN_ACT_THRESHOLDS) {
        r = d.ACT_VAL;
        const int32_t *t = thr + (size_t)o * d.NUM_TH;
        for (int i = 0; i < d.NUM_TH; i++) r += (wrap_acc((long long)t[i], d.ACC_BIT, d.ACC_SIGNED) < a) ? 1 : 0;
    }
    const size_t oi = (size_t)img * per_img + idx;
    if (d.OUT_BIT == 8)
        ((uint8_t *)out)[oi] = (uint8_t)r;
    else if (d.OUT_BIT == 16)
        ((uint16_t *)out)[oi] = (uint16_t)r;
    else
        ((uint32_t *)out)[oi] = (uint32_t)r;
}

}  // namespace

extern "C" int sicn_convlayer_validate(const sicn_convlayer_desc *d)
{
    if (!d) return SICN_EINVAL;
    if (d->K < 1 || d->K > 11 || d->IFM_CH <= 0 || d->OFM_CH <= 0 || d->SIMD <= 0 || d->PE <= 0) return SICN_EINVAL;
    if (d->IFM_DIM < d->K || d->IFM_DIM > (1 << 15) || d->OFM_DIM != d->IFM_DIM - d->K + 1) return SICN_EINVAL;
    if (d->IFM_CH % d->SIMD || d->OFM_CH % d->PE) return SICN_EINVAL;   // slidingwindow.h:177, mvau.hpp:101-105
    if (d->IN_BIT != 8 || (d->IN_SIGNED != 0 && d->IN_SIGNED != 1)) return SICN_EINVAL;
    if (d->W_BIT < 2 || d->W_BIT > 8 || d->SIMD * d->W_BIT > 64) return SICN_EINVAL;
    if ((long long)d->W_TILES != (long long)(d->OFM_CH / d->PE) * ((long long)d->K * d->K * d->IFM_CH / d->SIMD)) return SICN_EINVAL;
    if (d->ACC_BIT < 1 || d->ACC_BIT > 32 || (d->ACC_SIGNED != 0 && d->ACC_SIGNED != 1)) return SICN_EINVAL;
    if (d->OUT_BIT != 8 && d->OUT_BIT != 16 && d->OUT_BIT != 32) return SICN_EINVAL;
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
    bool ok = hipMalloc((void **)&p->d_w_okc, w.size()) == hipSuccess &&
              hipMemcpy(p->d_w_okc, w.data(), w.size(), hipMemcpyHostToDevice) == hipSuccess;
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
    int rc = sicn_convlayer_validate(d);
    if (rc) return rc;
    if (!p || !in || !out || reps < 0 || reps > 65535) return SICN_EINVAL;
    const sicn_convlayer_desc &q = p->d;   // the parameters must have been made for this layer
    if (q.K != d->K || q.IFM_CH != d->IFM_CH || q.OFM_CH != d->OFM_CH || q.W_BIT != d->W_BIT ||
        q.activation != d->activation || q.NUM_TH != d->NUM_TH)
        return SICN_EINVAL;
    if (reps == 0) return SICN_OK;
    const size_t per_img = (size_t)d->OFM_DIM * d->OFM_DIM * d->OFM_CH;
    const size_t blocks = (per_img + 255) / 256;
    if (blocks > 0x7fffffffu) return SICN_EINVAL;
    hipLaunchKernelGGL(k_convlayer, dim3((unsigned)blocks, (unsigned)reps), dim3(256), 0, (hipStream_t)hip_stream, in, out,
                       p->d_w_okc, p->d_thr, *d);
    return hipGetLastError() == hipSuccess ? SICN_OK : SICN_ENODEV;
}
