"""ORACLE (test infrastructure, NOT product code) — numpy closed-form restatement of the
reference's conv/deconv transform path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`simple_image_compression_network_amd/`) never does.

Pinning status: PINNED by reference code.  The reference's own golden convolution `conv_nonsquare<>`
(conv.hpp:91-123) compiles here as it stands (oracle/ref_harness.cpp -> oracle/_ref, `make -C oracle ref`);
tests/golden/make_ref_conv_vectors.py ran it on seeded random layers and over the whole net with the PARAM
tables, and tests/test_oracle_golden.py holds this module to those bytes (tests/golden/ref_conv_vectors.npz,
ref_conv_hashes.json — the latter reproduces all 24 hashes of SURVEY.md Appendix A).  The dataflow templates
(conv_nonsquare_top.cpp) need Vivado-HLS headers and were not built; no stand-in header was written.

Each function cites the reference lines it follows.  Tensors are row-major `[H][W][C] uint8`
(byte-identical to the reference's `hls::stream<ap_uint<C*8>>`, channel c in bits [8c,8c+8),
conv3_nonsquare_tb.cpp:807-808,1080).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

K5 = 5


# ------------------------------------------------------------------------------------------
# Weight wire format: FixedPointWeights<SIMD, ap_int<4>, PE, TILES>  (weights.hpp:110-150)
# ------------------------------------------------------------------------------------------
def unpack_finn_tiles(words: np.ndarray, simd: int, pe: int, cin: int, cout: int) -> np.ndarray:
    """m_weights[PE][TILES] words -> W[o][ky][kx][c] int8.

    weights.hpp:134-139: element `s` of tile word t is the sign-extended nibble in bits
    [4s, 4s+4).  mvau.hpp:149-156 with invariant tile = nf*SF + sf (mvau.hpp:118): output
    channel o = nf*PE + pe, K index k = sf*SIMD + s.  The sliding-window generator emits
    ky -> kx -> channel-chunk order (slidingwindow.h:1304-1325), so k = (ky*5+kx)*Cin + c;
    cross-checked by the testbench's own unpack loop (conv3_nonsquare_tb.cpp:546-571).
    """
    words = np.asarray(words, dtype=np.uint64)
    kk = K5 * K5 * cin
    sf_n, nf_n = kk // simd, cout // pe
    assert words.shape == (pe, nf_n * sf_n), (words.shape, pe, nf_n, sf_n)
    shifts = (np.arange(simd, dtype=np.uint64) * np.uint64(4))
    nib = ((words[:, :, None] >> shifts[None, None, :]) & np.uint64(15)).astype(np.int16)
    nib = np.where(nib > 7, nib - 16, nib).astype(np.int8)           # [pe][tile][s]
    nib = nib.reshape(pe, nf_n, sf_n * simd)                         # [pe][nf][k]
    w = nib.transpose(1, 0, 2).reshape(cout, K5, K5, cin)            # o = nf*PE + pe
    return np.ascontiguousarray(w)


def pack_finn_tiles(w: np.ndarray, simd: int, pe: int) -> np.ndarray:
    """Inverse of `unpack_finn_tiles`: W[o][ky][kx][c] int4-valued -> uint64 words [PE][TILES]."""
    cout, _, _, cin = w.shape
    kk = K5 * K5 * cin
    sf_n, nf_n = kk // simd, cout // pe
    assert w.min() >= -8 and w.max() <= 7
    nib = (w.reshape(nf_n, pe, sf_n, simd).astype(np.int16) & 15).astype(np.uint64)
    shifts = (np.arange(simd, dtype=np.uint64) * np.uint64(4))
    words = (nib << shifts[None, None, None, :]).sum(axis=3, dtype=np.uint64)   # [nf][pe][sf]
    return np.ascontiguousarray(words.transpose(1, 0, 2).reshape(pe, nf_n * sf_n))


# ------------------------------------------------------------------------------------------
# Closed forms of SURVEY.md §8(a) rows a8 / a9
# ------------------------------------------------------------------------------------------
def _exact_gemm_ok(k: int, xmax: int) -> None:
    # float32 sums of integers stay exact while every partial sum is < 2^24
    assert k * xmax * 8 < (1 << 24), "fp32 accumulation would not be exact"


def _bias_relu_wrap8(acc: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """conv_nonsquare_top.cpp:267-278 / 183-194: lane = (lane + bias) mod 2^8; MSB set -> 0.
    The MVAU accumulator is ap_uint<8> (mvau.hpp:112 with activations.hpp:112-115,127-134), so
    every `+=` of mac.hpp:166-169 wraps mod 256; computing wide and truncating once is the same
    ring homomorphism."""
    v = (acc.astype(np.int64) + bias.astype(np.int64)[None, None, :]) & 0xFF
    v[v >= 128] = 0
    return v.astype(np.uint8)


def _conv_taps(xpad: np.ndarray, w: np.ndarray, oh: int, ow: int, stride: int) -> np.ndarray:
    """sum_{ky,kx,c} xpad[stride*y+ky][stride*x+kx][c] * W[o][ky][kx][c]  -> int64 [oh][ow][o]."""
    cout, _, _, cin = w.shape
    _exact_gemm_ok(25 * cin, int(xpad.max()) if xpad.size else 0)
    acc = np.zeros((oh * ow, cout), dtype=np.float32)
    xf = xpad.astype(np.float32)
    wf = w.astype(np.float32)
    for ky in range(K5):
        for kx in range(K5):
            a = xf[ky:ky + stride * (oh - 1) + 1:stride, kx:kx + stride * (ow - 1) + 1:stride, :]
            acc += a.reshape(oh * ow, cin) @ wf[:, ky, kx, :].T
    return np.rint(acc).astype(np.int64).reshape(oh, ow, cout)


def conv2d_ref(x: np.ndarray, w: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """`conv2d<>` (conv_nonsquare_top.cpp:198-280): 2-pixel zero border (padding2 ->
    FMPadding_nonsquare, top:59-69, streamtools.h:369-406), stride-1 im2col
    (slidingwindow.h:1254-1353) decimated to even rows/cols (top:243-259), MVAU, bias+ReLU.
    out[y][x][o] = relu7((sum xpad[2y+ky][2x+kx][c] W[o][ky][kx][c] + b[o]) mod 256)."""
    h, wd, _ = x.shape
    oh, ow = (h + 1) // 2, (wd + 1) // 2
    xpad = np.pad(x, ((2, 2), (2, 2), (0, 0)))
    return _bias_relu_wrap8(_conv_taps(xpad, w, oh, ow, 2), bias)


def zero_stuff_pad(x: np.ndarray) -> np.ndarray:
    """deconv522 input staging (conv_nonsquare_top.cpp:110-156): zero-insert to (2H-1)x(2W-1),
    one zero column right / row bottom, then 2-pixel border: Up_pad[2i+2][2j+2] = x[i][j]."""
    h, wd, c = x.shape
    up = np.zeros((2 * h + 4, 2 * wd + 4, c), dtype=x.dtype)
    up[2:2 + 2 * h:2, 2:2 + 2 * wd:2, :] = x
    return up


def deconv522_ref(x: np.ndarray, w: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """`deconv522<>` (conv_nonsquare_top.cpp:71-195): stride-1 5x5 conv over the zero-stuffed,
    padded map, kernel NOT flipped; out is 2H x 2W."""
    h, wd, _ = x.shape
    return _bias_relu_wrap8(_conv_taps(zero_stuff_pad(x), w, 2 * h, 2 * wd, 1), bias)


def layer_preact_ref(x: np.ndarray, w: np.ndarray, bias: np.ndarray, transposed: int) -> np.ndarray:
    """The layer's 8-bit lanes after the bias add, BEFORE the sign-bit ReLU (conv_nonsquare_top.cpp:272)."""
    h, wd, _ = x.shape
    if transposed:
        acc = _conv_taps(zero_stuff_pad(x), w, 2 * h, 2 * wd, 1)
    else:
        acc = _conv_taps(np.pad(x, ((2, 2), (2, 2), (0, 0))), w, (h + 1) // 2, (wd + 1) // 2, 2)
    return ((acc + bias.astype(np.int64)[None, None, :]) & 0xFF).astype(np.uint8)


GDN_BIAS = (5, 33)          # r = trunc11(root * 2^s * (1 + GDN_BIAS[inverse] * 2^-16)), oracle/sicn_gdn_oracle.c
GDN_N_BITS = 11             # significant bits of nq
GDN_R_BITS = 11             # significant bits of r


def _rne_shift(v: int, sh: int) -> int:
    """v / 2^sh rounded to the nearest integer, ties to even (v >= 0, sh >= 0)."""
    if sh == 0:
        return v
    q, rem, half = v >> sh, v & ((1 << sh) - 1), 1 << (sh - 1)
    return q + (1 if rem > half or (rem == half and q & 1) else 0)


def gdn_quantise_n(n: int) -> Tuple[int, int]:
    """nq = n rounded to 24 significant bits (ties to even), cut to its top 11: returns (m, e), nq = m 2^e, 1024 <= m < 2048."""
    length = n.bit_length()
    if length > 24:
        n = _rne_shift(n, length - 24) << (length - 24)     # may carry into bit `length`: still a multiple of 2^(length-24)
        length = n.bit_length()
    e = length - GDN_N_BITS
    return (n >> e, e) if e >= 0 else (n << -e, e)


def gdn_root(m: int, e: int, inverse: bool, shift: int) -> Tuple[int, int]:
    """r = trunc11(2^s (1 + b 2^-16) / sqrt(nq)) (GDN, s = 16 - shift) or trunc11(2^s (1 + b 2^-16) sqrt(nq)) (IGDN, s = 8 - shift)
    for nq = m 2^e, as (M, k): r = M 2^k, 1024 <= M < 2048.  Exact: floor(V 2^T) = isqrt(floor(V^2 4^T)) and a floor of a floor."""
    import math
    t = 64
    b2 = (65536 + GDN_BIAS[1 if inverse else 0]) ** 2
    s = (8 if inverse else 16) - shift
    num, den = b2, 1 << 32                                   # V^2 = num / den, built from integers only
    if inverse:
        num *= m
        if e >= 0: num <<= e
        else: den <<= -e
    else:
        den *= m
        if e >= 0: den <<= e
        else: num <<= -e
    if s >= 0: num <<= 2 * s
    else: den <<= -2 * s
    v = math.isqrt((num << (2 * t)) // den)
    length = v.bit_length()
    return v >> (length - GDN_R_BITS), length - GDN_R_BITS - t


def gdn_ref(lanes: np.ndarray, beta: np.ndarray, gamma: np.ndarray, inverse: bool, shift: int) -> np.ndarray:
    """Second, independent statement of the fixed-point GDN / IGDN v2 of oracle/sicn_gdn_oracle.c (parity unpinned — the
    reference has no GDN), in integers only: the roots by math.isqrt on big integers, the binary32 fused multiply-add and the
    nearest-even conversion restated as integer roundings (the C file uses fmaf / nearbyintf and a 128-bit bisection).
        x = int8(v); n_i = beta_i + sum_j gamma[i][j] x_j^2; nq = top 11 bits of (n rounded to 24 bits)
        r = trunc11(root(nq) 2^s (1 + b 2^-16));  u = round24(x r + 128);  y = clamp(round(u), 0, 255) - 128   (mod 256)"""
    c = lanes.shape[-1]
    x = lanes.reshape(-1, c).view(np.int8).astype(np.int64)
    n = beta.astype(np.int64)[None, :] + (x * x) @ gamma.astype(np.int64).T
    uniq, inv = np.unique(n.reshape(-1), return_inverse=True)
    roots = [gdn_root(*gdn_quantise_n(int(v)), inverse, shift) for v in uniq]
    big_m = np.array([r[0] for r in roots], np.int64)[inv].reshape(n.shape)
    k = np.array([r[1] for r in roots], np.int64)[inv].reshape(n.shape)
    return gdn_output_ref(x, big_m, k).reshape(lanes.shape)


def gdn_output_ref(x: np.ndarray, big_m: np.ndarray, k: np.ndarray) -> np.ndarray:
    """The output step in integers: the stored byte for lanes x (int64, -128 .. 127) and roots r = big_m 2^k (1024 <= big_m < 2048):
    u = x r + 128 rounded once to 24 significant bits (nearest-even), then to the nearest-even integer, clamped to 0 .. 255, minus 128."""
    x, big_m, k = np.broadcast_arrays(np.asarray(x, np.int64), np.asarray(big_m, np.int64), np.asarray(k, np.int64))
    p = x * big_m                                            # |p| < 2^19; the exact value is p 2^k + 128
    out = np.empty(p.shape, np.int64)
    pos = k >= 0                                             # an integer: below 2^24 it is exact, above it the clamp decides
    out[pos] = np.clip((p[pos] << k[pos]) + 128, 0, 255)
    s = -k[~pos]
    v = p[~pos] + (np.int64(128) << s)                       # in units of 2^-s, < 2^43
    neg = v <= 0
    v = np.where(neg, 1, v)
    length = np.frexp(v.astype(np.float64))[1].astype(np.int64)   # bit length (exact below 2^53)
    d = np.maximum(length - 24, 0)                           # first rounding: to 24 significant bits
    q, rem, half = v >> d, v & ((np.int64(1) << d) - 1), (np.int64(1) << d) >> 1
    v = (q + ((d > 0) & ((rem > half) | ((rem == half) & (q & 1 == 1))))) << d
    q, rem, half = v >> s, v & ((np.int64(1) << s) - 1), np.int64(1) << (s - 1)   # second rounding: to an integer (s >= 1)
    y = q + ((rem > half) | ((rem == half) & (q & 1 == 1)))
    out[~pos] = np.where(neg, 0, np.clip(y, 0, 255))
    return ((out - 128) & 0xFF).astype(np.uint8)


Params = Sequence[Tuple[np.ndarray, np.ndarray, int]]   # (W[o][ky][kx][c], bias[o], transposed)


def eight_layers_net_ref(x: np.ndarray, params: Params) -> List[np.ndarray]:
    """`eight_layers_net` (conv_nonsquare_top.cpp:295-357): returns every layer output."""
    outs = []
    for w, b, transposed in params:
        x = deconv522_ref(x, w, b) if transposed else conv2d_ref(x, w, b)
        outs.append(x)
    return outs


def load_param_fixture(path) -> List[Tuple[np.ndarray, np.ndarray, int]]:
    """tests/golden/param_weights.npz -> [(W, bias, transposed)] for the 8 reference layers."""
    z = np.load(path)
    chans = ((3, 128, 0), (128, 128, 0), (128, 128, 0), (128, 192, 0),
             (192, 128, 1), (128, 128, 1), (128, 128, 1), (128, 3, 1))
    out = []
    for n, (cin, cout, tr) in enumerate(chans):
        simd, wbit, pe, tiles = (int(v) for v in z[f"w{n}_meta"])
        assert wbit == 4 and tiles == (cout // pe) * (25 * cin // simd)
        out.append((unpack_finn_tiles(z[f"w{n}_words"], simd, pe, cin, cout), z[f"b{n}"].copy(), tr))
    return out



# ------------------------------------------------------------------------------------------
# Generic ConvLayer_Batch (convlayer.h:89-125) — closed form; parity unpinned (never run by the reference)
# ------------------------------------------------------------------------------------------
def pack_finn_tiles_generic(w_ok: np.ndarray, simd: int, pe: int, w_bit: int) -> np.ndarray:
    """W[o][k] (signed, fits w_bit) -> FixedPointWeights words uint64 [PE][TILES] (weights.hpp:110-150)."""
    cout, kk = w_ok.shape
    sf_n, nf_n = kk // simd, cout // pe
    el = (w_ok.reshape(nf_n, pe, sf_n, simd).astype(np.int64) & ((1 << w_bit) - 1)).astype(np.uint64)
    shifts = (np.arange(simd, dtype=np.uint64) * np.uint64(w_bit))
    words = (el << shifts[None, None, None, :]).sum(axis=3, dtype=np.uint64)
    return np.ascontiguousarray(words.transpose(1, 0, 2).reshape(pe, nf_n * sf_n))


def _wrap(v: np.ndarray, bits: int, signed: bool) -> np.ndarray:
    u = v.astype(np.int64) & ((1 << bits) - 1)
    return np.where(u >> (bits - 1) & 1, u - (1 << bits), u) if signed else u


def conv_layer_batch_ref(x: np.ndarray, w_ok: np.ndarray, k: int, in_signed: bool, acc_bit: int, acc_signed: bool,
                         out_bit: int, thresholds_oi=None, act_val: int = 0, in_bit: int = 8) -> np.ndarray:
    """x [D][D][C] uint8 lanes, w_ok [O][k*k*C] with K index (ky*k+kx)*C + c (slidingwindow.h:163-270 emits
    ky -> kx -> channel).  Stride 1, no padding.  thresholds_oi: [O][NumTH] in channel order, or None for
    PassThroughActivation.  Returns uint32 lanes holding the low out_bit bits."""
    d, _, c = x.shape
    od = d - k + 1
    xs = _wrap(x, in_bit, in_signed)     # TSrcI = Slice<ap_(u)int<in_bit>>: the lane's low in_bit bits, sign-extended for ap_int
    cols = np.stack([xs[ky:ky + od, kx:kx + od, :] for ky in range(k) for kx in range(k)], axis=2).reshape(od * od, k * k * c)
    acc = _wrap(cols @ w_ok.astype(np.int64).T, acc_bit, acc_signed)
    if thresholds_oi is not None:
        th = _wrap(np.asarray(thresholds_oi), acc_bit, acc_signed)                 # TA m_thresholds
        acc = act_val + (th[None, :, :] < acc[:, :, None]).sum(axis=2)
    return (acc & ((1 << out_bit) - 1)).astype(np.uint32).reshape(od, od, -1)


def pack_stream_lanes(lanes: np.ndarray, bits: int) -> np.ndarray:
    """[...][C] lane values -> the stream words of those pixels as bytes [...][C * bits / 8]: word = sum_c (lane_c mod 2^bits) << (c * bits)
    (`ap_uint<C * bits>` with Slice<> lane c in bits [c bits, (c + 1) bits), interpret.hpp:191-244; convlayer.h:100), little-endian.
    Restated with Python integers per pixel, deliberately not with the byte-level shifts the kernels use."""
    c = lanes.shape[-1]
    assert (c * bits) % 8 == 0
    flat = lanes.reshape(-1, c)
    out = np.empty((flat.shape[0], c * bits // 8), np.uint8)
    for i, px in enumerate(flat):
        word = 0
        for k, v in enumerate(px):
            word |= (int(v) & ((1 << bits) - 1)) << (k * bits)
        out[i] = np.frombuffer(word.to_bytes(c * bits // 8, "little"), np.uint8)
    return out.reshape(lanes.shape[:-1] + (c * bits // 8,))


def unpack_stream_lanes(words: np.ndarray, bits: int, channels: int) -> np.ndarray:
    """Inverse of pack_stream_lanes: bytes [...][C * bits / 8] -> lanes [...][C] as unsigned values (uint32)."""
    nb = channels * bits // 8
    assert words.shape[-1] == nb
    flat = words.reshape(-1, nb)
    out = np.empty((flat.shape[0], channels), np.uint32)
    for i, px in enumerate(flat):
        word = int.from_bytes(px.tobytes(), "little")
        out[i] = [(word >> (k * bits)) & ((1 << bits) - 1) for k in range(channels)]
    return out.reshape(words.shape[:-1] + (channels,))
